#!/bin/bash
# Builds the product library of another revision of the sources into tools/bin/ (for tools/ab_libs.py): bash tools/build_rev.sh <rev>
REV=${1:?revision}
REPO=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C $REPO archive $REV sqair_amd/csrc include | tar -x -C $TMP
OUT=$REPO/tools/bin/libsqair_hip_$(git -C $REPO rev-parse --short $REV).so
mkdir -p $REPO/tools/bin
cd $TMP/sqair_amd/csrc
for f in sqair_api sqair_linear sqair_glue sqair_bwd sqair_train sqair_linear_dx; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form=1 \
    -DSQAIR_BUILD_ID="\"rev$(git -C $REPO rev-parse --short=13 $REV)\"" -DSQAIR_BUILD_VARIANT='"product"' -c $f.hip -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT *.o && echo $OUT
rm -rf $TMP
