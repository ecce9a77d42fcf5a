python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for v in prev base; do if [ $v = base ]; then unset ORBFE_LIB; else export ORBFE_LIB=$PWD/jetracer-orbslam2_amd/.variants/$v/liborbfe.so; fi
for m in c5 c2 c4; do python bench.py --mode $m --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v $m', round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['stage_ms'].items()})"; done; done
unset ORBFE_LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5trace4 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --mode c5 --no-cpu-baseline --no-extras --steps 50 --warmup 5 > /dev/null 2>&1
grep select_kernel $(find $GRAFT_REPO_ROOT/gpurun_out/c5trace4 -name "*kernel_stats.csv" | head -1) | sed 's/.*)",//' 
