#!/bin/bash
# Like run_variants.sh for another scene: SCENE=fog SPP=2 tools/run_variants_scene.sh [NAME:VAR=VALUE,...]
shopt -s nullglob
out=gpurun_out/variants_$SCENE; mkdir -p $out
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --scene $SCENE --spp ${SPP:-2} --steps 1 --warmup 1 --no-cpu-baseline > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -3 $out/$name.err; return 1; }
  python - "$name" $out/$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(sys.argv[1], "Mpaths/s %.2f"%d["value"], "Mrays/s %.0f"%d["mrays_per_s"], "ms/step %.0f"%d["ms_per_step"], {k:round(v,1) for k,v in r.get("kernel_ms_per_step",{}).items() if v})
PY
}
run warmup A=1 > /dev/null || exit 1   # (the first process on a fresh box measures low: not compared)
run base A=1 || exit 1
for f in pathtracer_amd/variants/libwapt_*.so; do
  n=$(basename $f .so); n=${n#libwapt_}
  run $n WAPT_LIB=$PWD/$f || exit 1
done
for extra in "$@"; do
  run ${extra%%:*} $(echo ${extra#*:} | tr "," " ") || exit 1
done
