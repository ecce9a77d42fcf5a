#!/bin/bash
# Per-phase instruction / stall counters of the detect and describe kernels: PMC passes over
# ablated builds (det1..det3, desc1..desc3 = -DORBFE_*_STOP_AFTER=n on tools/experiments/profiling_probes.patch) and the product.
# usage (through gpurun, from the repo root):  tools/phase_counters.sh <tag> [variants...]
# -> gpurun_out/<tag>/<variant>.json + a table on stdout.  Build the variants BEFORE the gpurun call (hipcc
# cross-compiles here):  tools/build_probe_variants.sh   (applies the patch to a scratch copy of the sources; the product
# sources carry none of these macros)
TAG=${1:-phases}; shift
VARS=${@:-base det1 det2 det3 desc1 desc2 desc3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  if [ "$v" = base ]; then unset ORBFE_LIB; else export ORBFE_LIB=$ROOT/jetracer-orbslam2_amd/.variants/$v/liborbfe.so; fi
  for pass in 1 2 3 4; do
    if [ $pass = 1 ]; then C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_WAVE_CYCLES";
    elif [ $pass = 2 ]; then C="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE";
    # r4: what the waves wait for, LDS side by side with the vector ALU
    elif [ $pass = 3 ]; then C="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES";
    else C="SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES"; fi
    rocprofv3 --pmc $C --output-format csv -d "$OUT/p_${v}_${pass}" -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --prewarm 2 > /dev/null 2>&1
  done
  python3 "$ROOT/tools/pmc_summary.py" "$OUT/p_${v}_1" "$OUT/p_${v}_2" "$OUT/p_${v}_3" "$OUT/p_${v}_4" > "$OUT/$v.json"
  rm -rf "$OUT/p_${v}_1" "$OUT/p_${v}_2" "$OUT/p_${v}_3" "$OUT/p_${v}_4"
  echo "done $v"
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.load(open(f))
    for k, v in d.items():
        if "detect_tile" not in k and "describe_tile" not in k: continue
        g = lambda c: v.get(c, {}).get("avg", 0)
        cyc = g("GRBM_GUI_ACTIVE") / 8
        wc = max(g("SQ_WAVE_CYCLES"), 1)
        print("%-9s %-22s VALU %6.1fM SALU %5.1fM LDS %5.2fM  gpu-cycles %7.0f  valu_busy %.2f  lds_conflict %.2f  wait_inst_any/wave_cycles %.2f"
              "  | of wave-cycles: waiting on an LDS instruction %.3f, LDS instruction active %.3f, VALU active %.3f | lds_active/cu-cycle %.2f idx_active %.2f addr_conflict %.2f" % (
            os.path.basename(f)[:-5], k.split("orbfe::")[1][:22], g("SQ_INSTS_VALU") / 1e6, g("SQ_INSTS_SALU") / 1e6, g("SQ_INSTS_LDS") / 1e6, cyc,
            g("SQ_ACTIVE_INST_VALU") * 4 / 1024 / max(cyc, 1), g("SQ_LDS_BANK_CONFLICT") / 256 / max(cyc, 1),
            g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc, g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_ACTIVE_INST_VALU") / wc,
            g("SQ_ACTIVE_INST_LDS") * 4 / 256 / max(cyc, 1), g("SQ_LDS_IDX_ACTIVE") / 256 / max(cyc, 1), g("SQ_LDS_ADDR_CONFLICT") / 256 / max(cyc, 1)))
PY
