#!/bin/bash
# Everything the round's profile directory holds, in one GPU call:
#   tools/profile_all.sh <tag>     -> gpurun_out/profile_<tag>/ (headline), scene_<tag>_fog, scene_<tag>_milk, stress lines
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
tools/profile_round.sh $TAG || exit 1
tools/profile_scene.sh ${TAG}_fog --scene fog --spp 2 --no-pmc || exit 1
tools/profile_scene.sh ${TAG}_milk --scene dragonMilk --spp 8 --no-pmc || exit 1
tools/bench_scenes.sh > $R/gpurun_out/profile_$TAG/scenes.txt 2>&1
for m in opaque mixed; do
  timeout -k 10 400 python tools/stress.py --spp 8 --materials $m 2>/dev/null | tail -n 1 > $R/gpurun_out/profile_$TAG/stress_$m.json
done
python tools/kernel_resources.py > $R/gpurun_out/profile_$TAG/kernel_resources.txt 2>/dev/null
echo all profiles done
