#!/bin/bash
# Collect the evidence behind bench.py's numbers on the GPU box (run through gpurun from the repo
# root):  tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>/{kernel_stats.csv,pmc_*.json,bench.json}
# Counter passes are separate runs with no tracing (see the gpurun rules).
set -e
TAG=${1:-final}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-extras --mode ${MODE:-c2}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- $BENCH --steps 20 --warmup 3 > "$OUT/trace.log" 2>&1
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o run -- $BENCH --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o run -- $BENCH --steps 3 --warmup 1 > /dev/null 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_fetch" "$OUT/pmc_write" > "$OUT/pmc_traffic.json"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq1" -o run -- $BENCH --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$OUT/pmc_sq2" -o run -- $BENCH --steps 3 --warmup 1 > /dev/null 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_sq1" "$OUT/pmc_sq2" > "$OUT/pmc_sq.json"
cd "$ROOT"
python3 bench.py --mode ${MODE:-c2} --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_nocpu.json" 2> "$OUT/bench.err"
cp "$OUT/bench_nocpu.json" "$OUT/bench.json"
python3 tools/make_traffic_json.py "$OUT/pmc_traffic.json" "$OUT/pmc_sq.json" ${MODE:-c2} "$ROOT/profiles/traffic.json" $(python3 -c "import json;print(json.load(open('$OUT/bench.json'))['config']['frames_per_gpu_per_step'])")
cp "$ROOT/profiles/traffic.json" "$OUT/traffic.json"  # (copy it back into profiles/ after the gpurun call and commit it)
python3 bench.py --mode ${MODE:-c2} --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
rm -rf "$OUT"/trace "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_sq1 "$OUT"/pmc_sq2
ls -la "$OUT"
