#!/bin/bash
# Build a variant of the library for A/B runs: tools/build_variant.sh NAME [extra hipcc flags...]
# -> build/NAME/libzeth_gpu.so (select it with ZETH_GPU_LIBRARY=build/NAME/libzeth_gpu.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/$name
for f in zeth_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result "$@" -c $f -o build/$name/$b.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC build/$name/*.o -o build/$name/libzeth_gpu.so
echo built build/$name/libzeth_gpu.so
