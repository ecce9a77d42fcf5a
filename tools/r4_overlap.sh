#!/bin/bash
# does the MFMA-bound matcher run UNDER the VALU-bound extraction when its workgroups can co-reside
# with detect's (6 x 25.9 KB fill a CU's 160 KB of LDS; the 4-wave matcher needs 36.9 KB and starves, the 2-wave form
# 18.4 KB)?  match_overlap_probe.py: the matcher of step i on a second stream under the extraction of step i + 1.
TAG=${1:-r4ov}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$(pwd)
ORBFE_MATCH_WAVES=2 timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "match" > $OUT/pytest_waves2.log 2>&1; echo "pytest waves2 rc=$?"; tail -2 $OUT/pytest_waves2.log
echo "== 4-wave matcher (36.9 KB LDS)" > $OUT/probe.txt
timeout -k 10 200 python tools/match_overlap_probe.py >> $OUT/probe.txt 2>&1
echo "== 2-wave matcher (18.4 KB LDS)" >> $OUT/probe.txt
ORBFE_MATCH_WAVES=2 timeout -k 10 200 python tools/match_overlap_probe.py >> $OUT/probe.txt 2>&1
cat $OUT/probe.txt
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ_[A-Z_]*MFMA[A-Z_]*\|SQ_BUSY_CU_CYCLES\|SQ_INSTS_MFMA" | sort -u > $OUT/mfma_counters.txt; cat $OUT/mfma_counters.txt
cd /tmp && export TMPDIR=/tmp
for nw in 4 2; do
  export ORBFE_MATCH_WAVES=$nw
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/$OUT/pm$nw -o run -- python3 $R/tools/match_overlap_probe.py > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $R/$OUT/pm$nw > $R/$OUT/pmc_waves$nw.json; rm -rf $R/$OUT/pm$nw
done
unset ORBFE_MATCH_WAVES
cd $R
python3 - <<PY
import json
for nw in (4,2):
    d=json.load(open("$OUT/pmc_waves%d.json"%nw))
    for k,v in d.items():
        if "match_mfma" in k or "detect_tile" in k:
            print(nw, k[:50], {c: round(x["avg"]) for c,x in v.items()})
PY
