#!/bin/sh
# Builds an experiment flavour of the product library: `sh profiles/tools/build_variant.sh NAME "-DFLAG ..." file.hip [file.hip ...]`
# recompiles the named sources with the extra flags and links them with the standard objects of the others into
# point-unet_amd/csrc/build/variants/libps_NAME.so (travels to the GPU box; never loaded by the package: the experiment scripts
# point _lib.LIB_PATH at it).
set -e
NAME=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../../point-unet_amd/csrc"
make -j8 >/dev/null
mkdir -p build/variants/$NAME
OBJS=""
for o in build/*.o; do
  b=$(basename $o .o)
  case "$b" in debug_hooks|debug_host|kdtree_host) continue;; esac
  skip=0
  for f in "$@"; do [ "$(basename $f .hip)" = "$b" ] && skip=1; done
  [ $skip = 1 ] || OBJS="$OBJS $o"
done
for f in "$@"; do
  b=$(basename $f .hip)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-function -Wno-pass-failed $FLAGS -c $f -o build/variants/$NAME/$b.o &
done
wait
for f in "$@"; do OBJS="$OBJS build/variants/$NAME/$(basename $f .hip).o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o build/variants/libps_$NAME.so
echo built build/variants/libps_$NAME.so
