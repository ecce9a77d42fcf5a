#!/bin/bash
# round 5: matrix-pipe / vector-ALU counters of the matcher on the default bench step: the product (base; ORBFE_MATCH=stream | tile in
# the environment forces a form) and, with a build of tools/experiments/matcher_forms.patch selected by ORBFE_LIB, its ORBFE_MATCH_V2 variants
# usage (through gpurun): tools/r5_match_pmc.sh <tag> [variants...]   -> gpurun_out/<tag>/pmc_<variant>.json + a table
TAG=${1:-r5pmc}; shift
VARS=${@:-base 22 t2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  if [ "$v" = base ]; then unset ORBFE_MATCH_V2; else export ORBFE_MATCH_V2=$v; fi
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY \
      --output-format csv -d $OUT/p1_$v -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --prewarm 2 > /dev/null 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES \
      --output-format csv -d $OUT/p2_$v -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --prewarm 2 > /dev/null 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k_$v -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 > /dev/null 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/p1_$v $OUT/p2_$v > $OUT/pmc_$v.json
  cp $(find $OUT/k_$v -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$v.csv 2>/dev/null
  rm -rf $OUT/p1_$v $OUT/p2_$v $OUT/k_$v
done
python3 - <<PY
import json, csv, glob, os
for v in "$VARS".split():
    d = json.load(open("$OUT/pmc_%s.json" % v))
    t = {}
    try:
        for row in csv.DictReader(open("$OUT/kernel_stats_%s.csv" % v)):
            t[row["Name"].split("(")[0]] = float(row["AverageNs"]) / 1e6
    except Exception:
        pass
    for k, c in d.items():
        if "match_mfma" not in k and "match_tile" not in k: continue
        g = lambda n: c.get(n, {}).get("avg", 0)
        cyc = g("GRBM_GUI_ACTIVE") / 8
        simd = 1024 * max(cyc, 1)
        wc = max(g("SQ_WAVE_CYCLES"), 1)
        ms = [x for n, x in t.items() if n == k]
        print("%-5s %-40s %s ms | MFMA busy %.3f  VALU issue quad-cycles %.3f  co-execution %.3f of SIMD-cycles | per wave: MFMA %.0f VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f | of wave-cycles: waiting %.3f, on LDS %.3f"
              % (v, k.split("orbfe::")[1][:40], ("%.4f" % ms[0]) if ms else "?", g("SQ_VALU_MFMA_BUSY_CYCLES") / simd, g("SQ_ACTIVE_INST_VALU") * 4 / simd,
                 g("SQ_VALU_MFMA_COEXEC_CYCLES") / simd, g("SQ_INSTS_MFMA") / max(g("SQ_WAVES"), 1), g("SQ_INSTS_VALU") / max(g("SQ_WAVES"), 1),
                 g("SQ_INSTS_SALU") / max(g("SQ_WAVES"), 1), g("SQ_INSTS_LDS") / max(g("SQ_WAVES"), 1), g("SQ_INSTS_VMEM_RD") / max(g("SQ_WAVES"), 1),
                 g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc))
PY
