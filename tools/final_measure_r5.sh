#!/bin/bash
# Round 5: everything the documents quote, on the frozen sources.  Three gpurun calls (each < 20 min):
#   tools/final_measure_r5.sh a <tag>   c2: kernel trace + PMC passes + bench (collect_profiles.sh); bench modes ref, c4, c5, match, ingest
#   tools/final_measure_r5.sh b <tag>   c3 / ref / c4 / c5 PMC passes (traffic.json entries), align (trace + PMC + bench), matcher counters
#   tools/final_measure_r5.sh c <tag>   per-phase counters of detect / describe, the probes, the soak
# then tools/assemble_profiles_r5.sh copies the merged gpurun_out/ results into profiles/ (tracked).
PART=$1; TAG=${2:-r05_final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$(pwd)
if [ "$PART" = a ]; then
  bash tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1; echo "collect rc=$?"
  for mode in ref c4 c5; do
    timeout -k 10 300 python bench.py --mode $mode --steps 20 --warmup 5 > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err; echo "bench $mode rc=$?"
  done
  timeout -k 10 300 python bench.py --mode match --steps 20 --warmup 5 > $OUT/bench_match.json 2> $OUT/bench_match.err; echo "bench match rc=$?"
  ORBFE_MATCH=stream timeout -k 10 300 python bench.py --mode match --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_match_stream.json 2>> $OUT/bench_match.err; echo "bench match stream rc=$?"
  ORBFE_MATCH=tile timeout -k 10 300 python bench.py --mode match --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_match_tile.json 2>> $OUT/bench_match.err; echo "bench match tile rc=$?"
  timeout -k 10 400 python bench.py --ingest > $OUT/bench_ingest.json 2> $OUT/bench_ingest.err; echo "bench ingest rc=$?"
  timeout -k 10 300 python bench.py --rotate 3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_rotate3.json 2> $OUT/bench_rotate3.err; echo "bench rotate rc=$?"
  timeout -k 10 300 python bench.py --batch 256 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/bench_batch256.json 2> $OUT/bench_batch256.err; echo "bench batch256 rc=$?"
  ORBFE_BENCH_FORCE_COMM=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_forcecomm.json 2> $OUT/bench_forcecomm.err; echo "bench forcecomm rc=$?"
  ORBFE_MATCH=stream timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_c2_stream_matcher.json 2> $OUT/bench_c2_stream.err; echo "bench c2 stream matcher rc=$?"
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_c2_tile_matcher.json 2>> $OUT/bench_c2_stream.err; echo "bench c2 tile matcher rc=$?"
elif [ "$PART" = b ]; then
  for m in c3 ref c4 c5; do MODE=$m bash tools/collect_profiles.sh ${TAG}_$m > $OUT/collect_$m.log 2>&1; echo "collect $m rc=$?"; done
  B="python3 $R/bench.py --mode align --steps 5 --warmup 3 --no-cpu-baseline"
  (cd /tmp && export TMPDIR=/tmp
   rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/al_trace -o run -- python3 $R/bench.py --mode align --steps 10 --warmup 3 --no-cpu-baseline > $R/$OUT/al_trace.log 2>&1
   cp "$(find $R/$OUT/al_trace -name '*kernel_stats.csv' | head -1)" $R/$OUT/align_kernel_stats.csv
   rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/al_pf -o run -- $B > /dev/null 2>&1
   rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/al_pw -o run -- $B > /dev/null 2>&1
   rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$OUT/al_ps -o run -- $B > /dev/null 2>&1
   python3 $R/tools/pmc_summary.py $R/$OUT/al_pf $R/$OUT/al_pw > $R/$OUT/align_pmc_traffic.json
   python3 $R/tools/pmc_summary.py $R/$OUT/al_ps > $R/$OUT/align_pmc_sq.json
   rm -rf $R/$OUT/al_trace $R/$OUT/al_pf $R/$OUT/al_pw $R/$OUT/al_ps)
  python3 tools/make_align_traffic.py $OUT/align_pmc_traffic.json 8 1024 profiles/traffic.json > $OUT/align_traffic_entry.json; echo "align traffic rc=$?"
  cp profiles/traffic.json $OUT/traffic.json
  timeout -k 10 300 python bench.py --mode align --steps 20 --warmup 5 > $OUT/bench_align.json 2> $OUT/bench_align.err; echo "bench align rc=$?"
  # the matcher's matrix-pipe / vector-ALU counters: the product (tile form) and the round-4 stream form
  bash tools/r5_match_pmc.sh ${TAG}_matchpmc base > $OUT/match_pmc_tile.txt 2>&1; echo "match pmc tile rc=$?"
  ORBFE_MATCH=stream bash tools/r5_match_pmc.sh ${TAG}_matchpmc_stream base > $OUT/match_pmc_stream.txt 2>&1; echo "match pmc stream rc=$?"
else
  bash tools/phase_counters.sh ${TAG}_phases base det1 det2 det3 desc1 desc2 desc3 > $OUT/phase_counters.txt 2>&1; echo "phases rc=$?"
  timeout -k 10 60 tools/coexec_probe > $OUT/coexec_probe.txt 2>&1; echo "coexec rc=$?"
  timeout -k 10 60 tools/mfma_fold_probe > $OUT/mfma_fold_probe.txt 2>&1; echo "fold probe rc=$?"
  timeout -k 10 200 python tools/ingest_probe.py 64 256 1024 > $OUT/ingest_probe.txt 2>&1; echo "ingest probe rc=$?"
  timeout -k 10 200 python tools/stage_latency.py > $OUT/stage_latency.txt 2>&1; echo "stage latency rc=$?"
  timeout -k 10 200 python tools/latency_probe.py > $OUT/latency_probe.txt 2>&1; echo "latency probe rc=$?"
  timeout -k 10 900 python tools/soak.py 2000 > $OUT/soak.txt 2>&1; echo "soak rc=$?"; tail -8 $OUT/soak.txt
fi
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], "value=%.4g"%d["value"], d["unit"], "ms=%.4f"%d["ms_per_step"], "roof=%.3f"%d["roofline"]["frac"], {k:round(v,4) for k,v in d.get("stage_ms",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
