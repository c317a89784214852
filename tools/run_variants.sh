#!/bin/bash
# Runs the short benchmark for the default library and every variant in pathtracer_amd/variants/.
shopt -s nullglob
out=gpurun_out/variants; mkdir -p $out
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --spp ${SPP:-256} --steps ${STEPS:-3} --no-cpu-baseline --no-pmc > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -3 $out/$name.err; return 1; }
  python - "$name" $out/$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(sys.argv[1], "Mpaths/s %.1f"%d["value"], "trace_ms %.4f"%r.get("avg_launch_ms",0), {k:round(v,1) for k,v in r.get("kernel_ms_per_step",{}).items()}, r.get("lane_utilisation"))
PY
}
run warmup A=1 > /dev/null || exit 1   # (the first process on a fresh box measures 1-3 % low: not compared)
run base A=1 || exit 1
for f in pathtracer_amd/variants/libwapt_*.so; do
  n=$(basename $f .so); n=${n#libwapt_}
  run $n WAPT_LIB=$PWD/$f || exit 1
done
for extra in "$@"; do   # NAME:VAR=VALUE runs of the default library
  run ${extra%%:*} $(echo ${extra#*:} | tr "," " ") || exit 1
done
