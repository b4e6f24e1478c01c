#!/bin/bash
# Builds the product library of another revision of the sources into tools/bin/ (for tools/ab_libs.py): bash tools/build_rev.sh <rev>
# or of the working tree with extra compiler flags: bash tools/build_rev.sh WT <name> [-DSOMETHING=1 ...] -> tools/bin/lib_<name>.so
REV=${1:?revision}
REPO=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
EXTRA=""
if [ "$REV" = "WT" ]; then
  NAME=${2:?name}; shift 2; EXTRA="$*"
  mkdir -p $TMP/sqair_amd && cp -r $REPO/sqair_amd/csrc $TMP/sqair_amd/ && cp -r $REPO/include $TMP/ && rm -rf $TMP/sqair_amd/csrc/_obj
  OUT=$REPO/tools/bin/lib_$NAME.so
  REV=HEAD
else
  git -C $REPO archive $REV sqair_amd/csrc include | tar -x -C $TMP
  OUT=$REPO/tools/bin/libsqair_hip_$(git -C $REPO rev-parse --short $REV).so
fi
mkdir -p $REPO/tools/bin
cd $TMP/sqair_amd/csrc
for f in $(ls *.hip | sed "s/\.hip$//"); do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form=1 $EXTRA \
    -DSQAIR_BUILD_ID="\"rev$(git -C $REPO rev-parse --short=13 $REV)\"" -DSQAIR_BUILD_VARIANT='"product"' -c $f.hip -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT *.o && echo $OUT
rm -rf $TMP
