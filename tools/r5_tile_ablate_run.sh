R=$(pwd); mkdir -p gpurun_out/r5i
for v in base ${@:-nobar nofetch nostore noread mfmaonly} base; do
  if [ $v = base ]; then unset ORBFE_LIB; else export ORBFE_LIB=$R/jetracer-orbslam2_amd/.variants/$v/liborbfe.so; fi
  python tools/r5_match_diag.py "" 2>&1 | grep "256:" | sed "s/^/$v /"
done
