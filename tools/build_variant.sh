#!/bin/bash
# Build a kernel-variant copy of libmedscan.so: tools/build_variant.sh NAME "file1.hip file2.hip" "-DFLAGS ..."
# Only the listed translation units are recompiled (with the extra flags); the rest is linked from the in-tree objects.
# Output: build/variants/libmedscan_NAME.so (git-ignored, travels with gpurun); select with MEDSCAN_LIBRARY=<path>.
set -e
NAME=$1; FILES=$2; EXTRA=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/medical_image_classification_amd/csrc
OUT=$ROOT/build/variants/$NAME
mkdir -p $OUT
OBJS=""
for f in api scan_fwd scan_bwd scan_bwd_ssd scan_ss2d scan_ss2d_bwd gemm gemm_f32 linear_bwd cross dwconv dwconv_nhwc ln_gate block_tail ln dtproj bn_relu ssd_carry ssd_chunk rms_gate cast conv3x3 adam; do
  if [[ " $FILES " == *" $f.hip "* ]]; then
    FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc -Wno-unused-function -fno-slp-vectorize"
    [[ $f == scan_bwd || $f == scan_ss2d_bwd ]] && FL="$FL -mllvm -amdgpu-sched-strategy=${SCHED:-iterative-ilp}"
    /opt/rocm/bin/hipcc $FL $EXTRA -I$ROOT/include -I$SRC -c $SRC/$f.hip -o $OUT/$f.o 2>&1 | grep -E "error" || true
    OBJS="$OBJS $OUT/$f.o"
  else
    OBJS="$OBJS $SRC/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/variants/libmedscan_$NAME.so $OBJS
echo built $ROOT/build/variants/libmedscan_$NAME.so
