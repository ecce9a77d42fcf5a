cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr_c3b -o run -- python3 $GRAFT_REPO_ROOT/bench.py --mode c3 --no-cpu-baseline --no-extras --steps 20 --warmup 5 > /dev/null 2>&1
grep match_window $(find $GRAFT_REPO_ROOT/gpurun_out/tr_c3b -name "*kernel_stats.csv" | head -1) | sed 's/.*)",//'
