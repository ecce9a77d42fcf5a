cd /tmp && export TMPDIR=/tmp
for m in ref c3 c4; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr_$m -o run -- python3 $GRAFT_REPO_ROOT/bench.py --mode $m --no-cpu-baseline --no-extras --steps 20 --warmup 5 > /dev/null 2>&1
done
