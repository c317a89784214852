"""Tree figures of one scene for every builder: nodes, bytes, depth, SAH cost, build time.
  python tools/bvh_info.py [chrome|glass|milk|cornell|fog]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (before the library: one HIP runtime per process)
from pathtracer_amd import scenes
from pathtracer_amd.renderer import wa_content

kind = sys.argv[1] if len(sys.argv) > 1 else "chrome"
sc = {"cornell": scenes.cornell, "fog": scenes.fog}.get(kind, lambda: scenes.dragon(kind))()
for name, builder, env in (("ploc", "device", None), ("lbvh", "device", "lbvh"), ("host", "host", None)):
    if env:
        os.environ["WAPT_DEVICE_BVH"] = env
    else:
        os.environ.pop("WAPT_DEVICE_BVH", None)
    c = wa_content(sc, 0, builder)
    i = c.bvh_info()
    print(f"{kind} {name}: nodes {i.n_nodes} tris {i.n_tris} MB {(i.node_bytes + i.tri_bytes) / 1e6:.1f} depth {i.max_depth} "
          f"sah {i.sah_cost:.2f} build {i.build_seconds:.3f}s", flush=True)
    c.close()
