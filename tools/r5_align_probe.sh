#!/bin/bash
# round 5 (ADVICE): align_depth -- (1) phase ablations of the FINAL kernel (al1..al3 from
# tools/experiments/profiling_probes.patch: no flush / no LDS splat / no projection; tools/build_probe_variants.sh align first),
# pipelined and unpipelined; (2) ORBFE_ALIGN_CHUNK sweep with the FETCH / WRITE counters: do fill, atomics and close stay
# cache-resident at smaller chunks?   usage (through gpurun): tools/r5_align_probe.sh <tag>
TAG=${1:-r5align}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
B="python3 $R/bench.py --mode align --steps 10 --warmup 3 --no-cpu-baseline"
cd $R
for v in base al1 al2 al3; do
  if [ $v = base ]; then unset ORBFE_LIB; else export ORBFE_LIB=$R/jetracer-orbslam2_amd/.variants/$v/liborbfe.so; fi
  timeout -k 10 100 $B > $OUT/abl_$v.json 2>> $OUT/abl.err
  ORBFE_ALIGN_NO_PIPE=1 timeout -k 10 100 $B > $OUT/abl_${v}_nopipe.json 2>> $OUT/abl.err
done
unset ORBFE_LIB
for c in 8 16 32 64 128 256; do
  ORBFE_ALIGN_CHUNK=$c timeout -k 10 100 $B > $OUT/chunk_$c.json 2>> $OUT/abl.err
done
cd /tmp && export TMPDIR=/tmp
for c in 16 32 128; do
  export ORBFE_ALIGN_CHUNK=$c
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pf$c -o run -- $B > /dev/null 2>&1
  timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pw$c -o run -- $B > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $OUT/pf$c $OUT/pw$c > $OUT/pmc_chunk$c.json
  rm -rf $OUT/pf$c $OUT/pw$c
done
unset ORBFE_ALIGN_CHUNK
cd $R
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/abl_*.json")) + sorted(glob.glob("$OUT/chunk_*.json"), key=lambda p: int(p.split("_")[-1][:-5])):
    try:
        d = json.load(open(f)); print("%-28s ms per 1024 frames %.4f  roofline frac %.3f" % (os.path.basename(f)[:-5], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
    except Exception as e: print(f, "ERR", e)
for f in sorted(glob.glob("$OUT/pmc_chunk*.json")):
    d = json.load(open(f)); tot = 0.0
    for k, v in d.items():
        if "align" not in k: continue
        fs, ws = v.get("FETCH_SIZE", {}), v.get("WRITE_SIZE", {})
        n = max(fs.get("n", 0), ws.get("n", 0))
        fb, wb = fs.get("avg", 0) * 2048 * n / 13.0, ws.get("avg", 0) * 1024 * n / 13.0  # 13 calls profiled; FETCH in KB x 2 (gfx950), WRITE in KB
        tot += fb + wb
        print("%-14s %-44s launches per call %5.1f  fetch %.2f GB  write %.2f GB per call of 1024 frames" % (os.path.basename(f)[:-5], k.split("orbfe::")[-1][:44], n / 13.0, fb / 1e9, wb / 1e9))
    print("%-14s total %.2f GB per call (algorithmic 2.50 GB)" % (os.path.basename(f)[:-5], tot / 1e9))
PY
