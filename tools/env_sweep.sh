#!/bin/bash
# Short benchmarks of several scenes under a list of environment settings (one process each, same box):
#   tools/env_sweep.sh "NAME:VAR=VAL,VAR2=VAL2" ...       scenes: SCENES="dragonChrome:64 dragonMilk:8"
out=gpurun_out/sweep; mkdir -p $out
SCENES=${SCENES:-"dragonChrome:64 dragonGlass:64 dragonMilk:8"}
run() { # name scene spp env...
  name=$1; scene=$2; spp=$3; shift 3
  env "$@" timeout -k 10 300 python bench.py --scene $scene --spp $spp --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-pmc > $out/$name.$scene.json 2> $out/$name.$scene.err || { echo "$name $scene FAILED"; tail -3 $out/$name.$scene.err; return 1; }
  python - "$name" $scene $out/$name.$scene.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1]); r=d.get("roofline",{}); u=r.get("lane_utilisation",{})
pr=u.get("probe",{})
print("%-10s %-13s Mpaths/s %8.1f  ms %7.1f  nodes %d sah? trace n/ray %.2f t/ray %.2f | probe n %.2f t %.2f | %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["config"]["bvh_nodes"], r.get("node_visits_per_ray",0), r.get("tri_tests_per_ray",0), pr.get("node_visits_per_ray",0), pr.get("tri_tests_per_ray",0), {k:round(v,1) for k,v in r.get("kernel_ms_per_step",{}).items() if v}))
PY
}
for sc in $SCENES; do run warmup ${sc%%:*} ${sc##*:} A=1 > /dev/null; break; done
for sc in $SCENES; do
  run base ${sc%%:*} ${sc##*:} A=1
  for extra in "$@"; do
    run ${extra%%:*} ${sc%%:*} ${sc##*:} $(echo ${extra#*:} | tr "," " ")
  done
done
