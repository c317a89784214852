#!/bin/bash
# The three BVH builders on the stress scene, one process each on the same box.
#   tools/builder_ab.sh [stress.py args]
ARGS="${@:---spp 8 --materials opaque}"
run() { name=$1; shift; env "$@" timeout -k 10 500 python tools/stress.py $ARGS 2> gpurun_out/builder_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'Mpaths/s %.1f Mrays/s %.0f nodes %d gb %.2f sah %.2f build %.2fs create %.1fs mean %.6f' % (d['mpaths_per_s'], d['mrays_per_s'], d['bvh_nodes'], d['bvh_gb'], d['sah'], d['build_s'], d['scene_create_s'], d['mean_radiance']))" || exit 1; }
run ploc A=1 && run lbvh WAPT_DEVICE_BVH=lbvh && run host WAPT_BVH_BUILDER=host
