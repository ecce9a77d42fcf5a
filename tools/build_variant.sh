#!/bin/bash
# Build a variant of liborbfe.so with extra flags / defines into jetracer-orbslam2_amd/.variants/<name>/liborbfe.so.
# usage: tools/build_variant.sh <name> [-p <patch> ...] [extra hipcc flags...]
#        (ORBFE_LIB=<that path> selects it in the Python harness: A/B timing, profiling passes)
# The variant is compiled from a SCRATCH COPY of jetracer-orbslam2_amd/csrc + include/: the product sources are never
# touched, so orbfe.source_hash() -- which stamps the PMC numbers in profiles/traffic.json -- keeps describing the product.
# -p applies a patch from tools/experiments/ (a name or a path) to the copy first, e.g. the wrong-result profiling probes:
#     tools/build_variant.sh det2 -p profiling_probes -DORBFE_DETECT_STOP_AFTER=2
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/jetracer-orbslam2_amd/.variants/$NAME
WORK=$(mktemp -d /tmp/orbfe_variant.XXXXXX)
trap 'rm -rf "$WORK"' EXIT
mkdir -p "$OUT" "$WORK/jetracer-orbslam2_amd"
cp -r "$ROOT/include" "$WORK/include"
cp -r "$ROOT/jetracer-orbslam2_amd/csrc" "$WORK/jetracer-orbslam2_amd/csrc"
rm -rf "$WORK/jetracer-orbslam2_amd/csrc/.obj"
while [ "$1" = "-p" ]; do
  P=$2; shift 2
  [ -f "$P" ] || P=$ROOT/tools/experiments/$P
  [ -f "$P" ] || P=$P.patch
  (cd "$WORK" && patch -s -p1 < "$P")
done
SRC=$WORK/jetracer-orbslam2_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c -o $OUT/stage.o $SRC/stage_kernels.hip &
/opt/rocm/bin/hipcc $F "$@" -c -o $OUT/batch.o $SRC/batch_kernels.hip &
/opt/rocm/bin/hipcc $F "$@" -c -o $OUT/align.o $SRC/align_depth.hip &
/opt/rocm/bin/hipcc $F "$@" -c -o $OUT/ingest.o $SRC/ingest.hip &
/opt/rocm/bin/hipcc $F -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form "$@" -c -o $OUT/mfma.o $SRC/match_mfma.hip &
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -c -o $OUT/wire.o $SRC/wire_bson.cpp &
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -c -o $OUT/pose.o $SRC/pose_host.cpp &
(g++ -O2 -std=c++17 -ffp-contract=off -o $OUT/gen_steer_table $SRC/gen_steer_table.cpp && $OUT/gen_steer_table $OUT/steer_events.inc 2>/dev/null && /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -I$OUT "$@" -c -o $OUT/steer.o $SRC/steer_table.cpp) &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liborbfe.so $OUT/stage.o $OUT/batch.o $OUT/align.o $OUT/ingest.o $OUT/mfma.o $OUT/wire.o $OUT/pose.o $OUT/steer.o
rm -f $OUT/*.o $OUT/gen_steer_table $OUT/steer_events.inc
echo $OUT/liborbfe.so
