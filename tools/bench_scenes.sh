#!/bin/bash
# Short benchmark of the other BASELINE scenes (reduced spp), one JSON file per scene under gpurun_out/scenes/.
out=gpurun_out/scenes; mkdir -p $out
run() { # scene spp
  timeout -k 10 300 python bench.py --scene $1 --spp $2 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $out/$1.json 2> $out/$1.err || { echo "$1 FAILED"; tail -5 $out/$1.err; return 1; }
  python - $1 $out/$1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(sys.argv[1], "Mpaths/s %.2f"%d["value"], "Mrays/s %.0f"%d["mrays_per_s"], "rays/path %.1f"%d["rays_per_path"], "ms/step %.0f"%d["ms_per_step"], {k:round(v,1) for k,v in r.get("kernel_ms_per_step",{}).items()})
PY
}
run dragonGlass ${GLASS_SPP:-64} && run dragonMilk ${MILK_SPP:-8} && run fog ${FOG_SPP:-2} && run cornell ${CORNELL_SPP:-64}
