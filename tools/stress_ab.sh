#!/bin/bash
# The stress scene with the default library and every variant in pathtracer_amd/variants/ (one process each, same box).
#   tools/stress_ab.sh [stress.py args]
shopt -s nullglob
run() { name=$1; shift; env "$@" timeout -k 10 400 python tools/stress.py $ARGS 2> gpurun_out/stress_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'Mpaths/s %.1f Mrays/s %.0f nodes %d gb %.2f sah %.2f create %.1fs mean %.6f' % (d['mpaths_per_s'], d['mrays_per_s'], d['bvh_nodes'], d['bvh_gb'], d['sah'], d['scene_create_s'], d['mean_radiance']))"; }
ARGS="${@:---spp 8 --materials opaque}"
run base A=1
for f in pathtracer_amd/variants/libwapt_*.so; do n=$(basename $f .so); run ${n#libwapt_} WAPT_LIB=$PWD/$f; done
