#!/bin/bash
# one SQ counter pass of the default bench: tools/pmc_quick.sh <tag>  -> gpurun_out/<tag>/pmc_sq1.json (+ printed summary)
TAG=${1:-pmc}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq1" -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 ${BENCH_ARGS} > /dev/null 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_sq1" > "$OUT/pmc_sq1.json"
rm -rf "$OUT/pmc_sq1"
python3 - <<PY
import json
d=json.load(open("$OUT/pmc_sq1.json"))
for k,v in d.items():
    if 'orbfe' not in k: continue
    g=lambda c: v.get(c,{}).get('avg',0)
    cyc=g('GRBM_GUI_ACTIVE')/8
    print('%-28s VALU %.1fM SALU %.1fM LDS %.2fM  cycles %.0f  valu_busy %.2f  lds_conflict %.2f of CU-cycles'%(k.split('orbfe::')[1][:28], g('SQ_INSTS_VALU')/1e6, g('SQ_INSTS_SALU')/1e6, g('SQ_INSTS_LDS')/1e6, cyc, g('SQ_ACTIVE_INST_VALU')*4/1024/max(cyc,1), g('SQ_LDS_BANK_CONFLICT')/256/max(cyc,1)))
PY
