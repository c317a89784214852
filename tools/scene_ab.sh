#!/bin/bash
# One scene, the default library against every variant in pathtracer_amd/variants/, alternating, same box:
#   tools/scene_ab.sh fog 2 [rounds]
shopt -s nullglob
scene=$1; spp=$2; rounds=${3:-2}
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --scene $scene --spp $spp --steps 2 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$scene [$name]', round(d['value'],2), round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['roofline']['kernel_ms_per_step'].items() if v})"; }
run warmup A=1 > /dev/null
for r in $(seq $rounds); do
  run base A=1
  for f in pathtracer_amd/variants/libwapt_*.so; do n=$(basename $f .so); run ${n#libwapt_} WAPT_LIB=$PWD/$f; done
done
