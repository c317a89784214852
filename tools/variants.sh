#!/bin/bash
# Builds experiment variants of libwapt.so (different -D switches) into pathtracer_amd/variants/.
#   tools/variants.sh name1 "-DFOO=1" name2 "-DBAR" ...
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
make -s build/scene_io.o build/bvh_build.o build/bvh_device.o
mkdir -p ../variants
# ALL=1: the flags also reach the host-side translation units (switches that change shared record layouts)
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  if [ "${ALL:-0}" = "1" ]; then
    mkdir -p build/all_$name
    /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -Wall -Wno-unused-function $flags -c scene_io.cpp -o build/all_$name/scene_io.o
    /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -Wall -Wno-unused-function $flags -c bvh_build.cpp -o build/all_$name/bvh_build.o
    /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize $flags -c bvh_device.hip -o build/all_$name/bvh_device.o
    /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize $flags -c wapt_gpu.hip -o build/all_$name/wapt_gpu.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics build/all_$name/scene_io.o build/all_$name/bvh_build.o build/all_$name/bvh_device.o build/all_$name/wapt_gpu.o -o ../variants/libwapt_$name.so -lpthread
    echo built $name
    continue
  fi
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize $flags -c wapt_gpu.hip -o build/wapt_gpu_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics build/scene_io.o build/bvh_build.o build/bvh_device.o build/wapt_gpu_$name.o -o ../variants/libwapt_$name.so -lpthread
  echo built $name
done
