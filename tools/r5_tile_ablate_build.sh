#!/bin/bash
# round 5: timing-only ablations of match_tile_kernel (WRONG results): nobar = no s_barrier in the loop, noexp = no in-loop expansion
# arithmetic, nostore = no ring writes (the expansion and the source fetches die with them), noread = no ring reads in the loop.  Builds .variants/{nobar,noexp,nostore} from a
# scratch copy of the sources; run tools/r5_tile_ablate_run.sh through gpurun afterwards.
set -e
cd /root/repo
ROOT=/root/repo
mk() { # name, sed expr
  NAME=$1; EXPR=$2
  W=$(mktemp -d /tmp/orbfe_v.XXXX); mkdir -p $W/jetracer-orbslam2_amd; cp -r include $W/; cp -r jetracer-orbslam2_amd/csrc $W/jetracer-orbslam2_amd/; rm -rf $W/jetracer-orbslam2_amd/csrc/.obj
  python3 - "$W/jetracer-orbslam2_amd/csrc/match_mfma.hip" "$NAME" <<'PY'
import sys
p,name=sys.argv[1],sys.argv[2]
s=open(p).read()
if name=="nobar":
    assert 'asm volatile("s_barrier" ::: "memory");' in s
    s=s.replace('        asm volatile("s_barrier" ::: "memory");\n        store(slot_c, E);','        store(slot_c, E);')
elif name=="noexp":
    s=s.replace("        mma2(std::true_type{}, P, cur, E);\n        E.key = expand_key(cur);","        mma2(std::false_type{}, P, cur, E);\n        E.key = (float)cur.kp;")
elif name=="noread":  # the ring is written but never read in the loop: the operands of the prologue's read serve every step
    import re
    a=s.index("    auto step = [&](auto slot_c, int t) {"); b=s.index("    int i = 0;\n    for (; i + 2 <= S; i += 2)")
    body=s[a:b]
    body=re.sub(r"        read2\(std::integral_constant<int, [^\n]*\n","",body)
    body=body.replace("mma2(std::false_type{}, Q, cur, E);","mma2(std::false_type{}, P, cur, E);").replace("        landed(Q);\n","")
    s=s[:a]+body+s[b:]
elif name=="nostore":
    s=s.replace("        store(slot_c, E); // step t + 2 into the slot everybody has just finished with\n","")
open(p,'w').write(s)
PY
  OUT=$ROOT/jetracer-orbslam2_amd/.variants/$NAME; mkdir -p $OUT
  SRC=$W/jetracer-orbslam2_amd/csrc
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
  /opt/rocm/bin/hipcc $F -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form -c -o $OUT/mfma.o $SRC/match_mfma.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liborbfe.so jetracer-orbslam2_amd/csrc/.obj/stage_kernels.o jetracer-orbslam2_amd/csrc/.obj/batch_kernels.o $OUT/mfma.o jetracer-orbslam2_amd/csrc/.obj/align_depth.o jetracer-orbslam2_amd/csrc/.obj/ingest.o jetracer-orbslam2_amd/csrc/.obj/wire_bson.o jetracer-orbslam2_amd/csrc/.obj/pose_host.o jetracer-orbslam2_amd/csrc/.obj/steer_table.o
  rm -rf $W $OUT/mfma.o; echo built $NAME
}
for v in ${@:-nobar noexp nostore noread}; do mk $v x; done
