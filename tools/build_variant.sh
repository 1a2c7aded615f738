#!/bin/bash
# A/B builds of libpcops: tools/build_variant.sh NAME "-DFLAG=1 ..."  ->  scanobjectnn_amd/libpcops_NAME.so
# (run a benchmark against it with PCOPS_LIB=$PWD/scanobjectnn_amd/libpcops_NAME.so; *.so is git-ignored and travels
# with gpurun).  The default library is untouched.
set -e
NAME=$1; shift
EXTRA="$*"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/scanobjectnn_amd/csrc
OBJ=/tmp/pcops_variant_$NAME
mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function $EXTRA"
pids=()
echo "$EXTRA" > $OBJ/flags.new
if ! cmp -s $OBJ/flags.new $OBJ/flags 2>/dev/null; then rm -f $OBJ/*.o; cp $OBJ/flags.new $OBJ/flags; fi
for f in abi sampling grouping interpolate knn mlp gather edgeconv head; do
  if [ ! -f $OBJ/$f.o ] || [ $SRC/$f.hip -nt $OBJ/$f.o ] || [ $SRC/common.h -nt $OBJ/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $OBJ/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scanobjectnn_amd/libpcops_$NAME.so $OBJ/*.o
echo built $ROOT/scanobjectnn_amd/libpcops_$NAME.so
