#!/bin/bash
# round 4: (1) MFMA / VALU co-execution micro-probe, (2) detect with its ring reads taken out (upper bound of any LDS-side
# rewrite of phase C), (3) per-phase counters with the LDS wait counters, (4) SQ_VALU_MFMA_COEXEC_CYCLES of the matcher
TAG=${1:-r4det}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$(pwd)
timeout -k 10 120 tools/coexec_probe > $OUT/coexec_probe.txt 2>&1; echo "coexec rc=$?"; cat $OUT/coexec_probe.txt
for v in base ringnolds base ringnolds; do
  if [ $v = base ]; then unset ORBFE_LIB; else export ORBFE_LIB=$R/jetracer-orbslam2_amd/.variants/$v/liborbfe.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms/step %.4f' % d['ms_per_step'], {k: round(x,4) for k,x in d['stage_ms'].items()})" | tee -a $OUT/ring_nolds_ab.txt
done
unset ORBFE_LIB
bash tools/phase_counters.sh ${TAG}_phases base det1 det2 det3 ringnolds > $OUT/phase_counters.txt 2>&1; cat $OUT/phase_counters.txt | grep -v "^done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/$OUT/pcoex -o run -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --prewarm 2 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $R/$OUT/pcoex > $R/$OUT/pmc_coexec.json; rm -rf $R/$OUT/pcoex
cd $R
python3 - <<PY
import json
d=json.load(open("$OUT/pmc_coexec.json"))
for k,v in d.items():
    if "match_mfma" in k or "describe_tile" in k:
        g=lambda c: v.get(c,{}).get("avg",0); cyc=g("GRBM_GUI_ACTIVE")/8
        print(k[:44], "cycles %.0f" % cyc, "mfma_busy/SIMD-cycle %.3f" % (g("SQ_VALU_MFMA_BUSY_CYCLES")/1024/max(cyc,1)), "valu_issue %.3f" % (g("SQ_ACTIVE_INST_VALU")*4/1024/max(cyc,1)),
              "coexec/SIMD-cycle %.3f" % (g("SQ_VALU_MFMA_COEXEC_CYCLES")/1024/max(cyc,1)), "raw coexec %d busy %d" % (g("SQ_VALU_MFMA_COEXEC_CYCLES"), g("SQ_VALU_MFMA_BUSY_CYCLES")))
PY
