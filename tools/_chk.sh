python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for v in prev base; do if [ $v = base ]; then unset ORBFE_LIB; else export ORBFE_LIB=$PWD/jetracer-orbslam2_amd/.variants/$v/liborbfe.so; fi
for m in c3 ref; do python bench.py --mode $m --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v $m', round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['stage_ms'].items()})"; done; done
