import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pathtracer_amd import renderer, scenes
W, H = 1920, 1080
spp = int(sys.argv[1])
c = renderer.wa_content(scenes.dragon("chrome", res=(W, H)))
def acc(ranges, ps, seed=11):
    r = renderer.udp_renderer(c, ps, 1, 1, 1, seed=seed)
    a = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
    rays = 0
    for b, n in ranges:
        st = r.render_accumulate(a.data_ptr(), b, n)
        rays += st.rays
    torch.cuda.synchronize()
    return (a / ps).cpu().numpy().reshape(H, W, 3), rays, st.n_trace_launches
w1, r1, l1 = acc([(0, spp)], spp)
w2, r2, l2 = acc([(0, spp)], spp)
p, r3, l3 = acc([(i, 1) for i in range(spp)], spp)
def bad(a, b): return int((np.abs(a - b) > (1e-4 * np.abs(b) + 1e-6)).sum())
print(os.environ.get("WAPT_RAD_CAP"), os.environ.get("WAPT_CHUNK"), "spp", spp, "launches", l1, "rays", r1, r2, r3, "bad whole2", bad(w2, w1), "bad singles", bad(p, w1), flush=True)
