R=$(pwd); mkdir -p gpurun_out/r5s
for v in base t2 t8 t16 prev3 base t2 t8 t16 prev3; do
  if [ $v = base ]; then unset ORBFE_LIB; else export ORBFE_LIB=$R/jetracer-orbslam2_amd/.variants/$v/liborbfe.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms/step %.4f' % d['ms_per_step'], {k: round(x,4) for k,x in d['stage_ms'].items()})" | tee -a gpurun_out/r5s/ab.txt
done
