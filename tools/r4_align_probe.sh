#!/bin/bash
# align_depth: phase ablations (variants al1..al3 = no flush / no LDS splat / no projection; build them first on the build host:
#   tools/build_probe_variants.sh align: al1..al3 = -DORBFE_ALIGN_ABLATE=n on tools/experiments/profiling_probes.patch) and PMC passes
TAG=${1:-r4b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$(pwd)
B="python3 $R/bench.py --mode align --steps 10 --warmup 3 --no-cpu-baseline"
for v in base al1 al2 al3; do
  if [ $v = base ]; then unset ORBFE_LIB; else export ORBFE_LIB=$R/jetracer-orbslam2_amd/.variants/$v/liborbfe.so; fi
  timeout -k 10 100 $B > $OUT/abl_$v.json 2>> $OUT/abl.err; echo "abl $v rc=$?"
done
unset ORBFE_LIB
ORBFE_ALIGN_PROTOCOL=zero timeout -k 10 100 $B > $OUT/abl_zero.json 2>> $OUT/abl.err
cd /tmp && export TMPDIR=/tmp
for lit in 0 1; do
  if [ $lit = 1 ]; then export ORBFE_ALIGN_PROTOCOL=literal; else export ORBFE_ALIGN_PROTOCOL=zero; fi
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pf$lit -o run -- $B > /dev/null 2>&1
  timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pw$lit -o run -- $B > /dev/null 2>&1
  timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$OUT/ps1$lit -o run -- $B > /dev/null 2>&1
  timeout -k 10 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/ps2$lit -o run -- $B > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $R/$OUT/pf$lit $R/$OUT/pw$lit $R/$OUT/ps1$lit $R/$OUT/ps2$lit > $R/$OUT/pmc_lit$lit.json
  rm -rf $R/$OUT/pf$lit $R/$OUT/pw$lit $R/$OUT/ps1$lit $R/$OUT/ps2$lit
done
unset ORBFE_ALIGN_PROTOCOL
cd $R
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/abl_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], "ms=%.4f"%d["roofline"]["avg_launch_ms"], "roof %.3f"%d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
for f in sorted(glob.glob("$OUT/pmc_lit*.json")):
    d=json.load(open(f))
    for k,v in d.items():
        if "align" not in k: continue
        print(f.split('/')[-1], k[:60], {c: round(x.get("avg",0)) for c,x in v.items()})
PY
