#!/bin/bash
# Builds experiment variants of libwapt.so (different -D switches) into pathtracer_amd/variants/.
#   tools/variants.sh name1 "-DFOO=1" name2 "-DBAR" ...
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
make -s build/scene_io.o build/bvh_build.o build/bvh_device.o
mkdir -p ../variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize $flags -c wapt_gpu.hip -o build/wapt_gpu_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics build/scene_io.o build/bvh_build.o build/bvh_device.o build/wapt_gpu_$name.o -o ../variants/libwapt_$name.so -lpthread
  echo built $name
done
