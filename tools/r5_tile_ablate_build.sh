#!/bin/bash
# round 5: timing-only ablations of match_tile_kernel (WRONG results): nobar = no s_barrier in the loop, nofetch = no source loads in the
# loop, nostore = no ring writes (the expansion and the source fetches die with them), noread = no ring reads in the loop, mfmaonly = none
# of the three (MFMAs and folds on stale operands).  Builds .variants/<name> from a
# scratch copy of the sources; run tools/r5_tile_ablate_run.sh through gpurun afterwards.
set -e
cd /root/repo
ROOT=/root/repo
mk() { # name, sed expr
  NAME=$1; EXPR=$2
  W=$(mktemp -d /tmp/orbfe_v.XXXX); mkdir -p $W/jetracer-orbslam2_amd; cp -r include $W/; cp -r jetracer-orbslam2_amd/csrc $W/jetracer-orbslam2_amd/; rm -rf $W/jetracer-orbslam2_amd/csrc/.obj
  python3 - "$W/jetracer-orbslam2_amd/csrc/match_mfma.hip" "$NAME" <<'PY'
import sys, re
p,name=sys.argv[1],sys.argv[2]
s=open(p).read()
def drop_barrier(s):
    assert s.count('        asm volatile("s_barrier" ::: "memory");\n') == 1
    return s.replace('        asm volatile("s_barrier" ::: "memory");\n', '')
def drop_fetch(s):
    assert "            if (!kFirst && m == 6) sr = fetch();" in s
    return s.replace("            if (!kFirst && m == 6) sr = fetch();", "")
def drop_stores(s):
    s, n = re.subn(r'            if \(!kFirst && m == [34]\) asm volatile\("ds_write_b128[^\n]*\n', "", s)
    assert n == 2
    a = s.index("            if (!kFirst && m == 5) {"); b = s.index("            if (m <= 2 || (!kFirst && m <= 5))")
    return s[:a] + s[b:]
def drop_reads(s):
    a = s.index("            if (m == 0)\n                asm volatile(\"ds_read_b128"); b = s.index("            if (!kFirst && m == 3)")
    return s[:a] + s[b:]
def apply(name):
    global s
    if name == "nobar": s = drop_barrier(s)
    elif name == "nofetch": s = drop_fetch(s)
    elif name == "nostore": s = drop_stores(s)   # (the expansion and the fetches die with the stores)
    elif name == "noread": s = drop_reads(s)     # the operands of the prologue's read serve every step
    elif name == "mfmaonly": s = drop_barrier(drop_stores(drop_reads(s)))
    elif name == "stamps":  # clock64() at entry / loop start / loop end / exit of wave 0, written over out_dist (tools/r5_tile_stamps.py reads them)
        s = s.replace("    // descriptor word `w` of record `kp` of frame f", "    const long long t_in = clock64(), w_in = wall_clock64();\n    // descriptor word `w` of record `kp` of frame f", 1)
        s = s.replace("    int i = 0;\n    for (; i + 2 <= S; i += 2) {\n        step(std::integral_constant<int, 0>{}, nx0);", "    const long long t_loop = clock64();\n    int i = 0;\n    for (; i + 2 <= S; i += 2) {\n        step(std::integral_constant<int, 0>{}, nx0);", 1)
        s = s.replace("    landed(P); // (the read of the slot after the last one", "    const long long t_done = clock64();\n    landed(P); // (the read of the slot after the last one", 1)
        a = s.index("bool match_mfma_uses_tile(int n_pairs")
        k = s.rindex("}\n", 0, s.rindex("// Which form a call takes", 0, a))
        s = s[:k] + "    if (threadIdx.x == 0 && out_dist) {\n        long long *o = reinterpret_cast<long long *>(out_dist + (size_t)pk * cap + blk * 512);\n        o[0] = t_in; o[1] = t_loop; o[2] = t_done; o[3] = clock64(); o[4] = w_in; o[5] = wall_clock64();\n    }\n" + s[k:]
        assert s.count("t_loop") == 2 and s.count("t_done") == 2
    elif name.startswith("dephodd"):  # as dephase, but the ODD workgroups of the first 512
        n = int(name[7:] or 5)
        s = s.replace("    // descriptor word `w` of record `kp` of frame f", "    { const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x; if (lin < 512u && (lin & 1u)) for (int z = 0; z < %d; z++) __builtin_amdgcn_s_sleep(127); }\n    // descriptor word `w` of record `kp` of frame f" % n, 1)
        assert "s_sleep" in s
    elif name.startswith("dephx"):  # as dephase, but by the hardware's wave slot: the second wave of each SIMD (HW_ID wave id bit 0)
        n = int(name[5:] or 5)
        s = s.replace("    // descriptor word `w` of record `kp` of frame f", "    { const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x; const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11)); if (lin < 512u && (hw & 1u)) for (int z = 0; z < %d; z++) __builtin_amdgcn_s_sleep(127); }\n    // descriptor word `w` of record `kp` of frame f" % n, 1)
        assert "s_sleep" in s
    elif name.startswith("dephase"):  # the second 256 workgroups of the launch start half a tile late: partners on a SIMD out of phase
        n = int(name[7:] or 5)
        s = s.replace("    // descriptor word `w` of record `kp` of frame f", "    { const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x; if (lin >= 256u && lin < 512u) for (int z = 0; z < %d; z++) __builtin_amdgcn_s_sleep(127); }\n    // descriptor word `w` of record `kp` of frame f" % n, 1)
        assert "s_sleep" in s
    else: raise SystemExit("unknown ablation " + name)
for part in name.split('+'): apply(part)
open(p,'w').write(s)
PY
  OUT=$ROOT/jetracer-orbslam2_amd/.variants/$NAME; mkdir -p $OUT
  SRC=$W/jetracer-orbslam2_amd/csrc
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
  /opt/rocm/bin/hipcc $F -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form -c -o $OUT/mfma.o $SRC/match_mfma.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liborbfe.so jetracer-orbslam2_amd/csrc/.obj/stage_kernels.o jetracer-orbslam2_amd/csrc/.obj/batch_kernels.o $OUT/mfma.o jetracer-orbslam2_amd/csrc/.obj/align_depth.o jetracer-orbslam2_amd/csrc/.obj/ingest.o jetracer-orbslam2_amd/csrc/.obj/wire_bson.o jetracer-orbslam2_amd/csrc/.obj/pose_host.o jetracer-orbslam2_amd/csrc/.obj/steer_table.o
  rm -rf $W $OUT/mfma.o; echo built $NAME
}
for v in ${@:-nobar nofetch nostore noread mfmaonly}; do mk $v x; done
