import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pathtracer_amd import renderer, scenes
W, H = 1920, 1080
c = renderer.wa_content(scenes.dragon("chrome", res=(W, H)))
def acc(ranges, ps, seed=11):
    r = renderer.udp_renderer(c, ps, 1, 1, 1, seed=seed)
    a = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
    rays = 0
    for b, n in ranges:
        st = r.render_accumulate(a.data_ptr(), b, n)
        rays += st.rays
        print("  call", b, n, "rays", st.rays, "ovf", st.stack_overflows, "launches", st.n_trace_launches, flush=True)
    torch.cuda.synchronize()
    return (a / ps).cpu().numpy().reshape(H, W, 3), rays
def cmp(a, b, name):
    d = np.abs(a - b)
    bad = d > (1e-4 * np.abs(b) + 1e-6)
    print(name, "bad", bad.sum(), "max", d.max(), "mean a", a.mean(), "mean b", b.mean(), flush=True)
    if bad.any():
        ys, xs, cs = np.nonzero(bad)
        print("  rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "first", list(zip(ys[:5], xs[:5], cs[:5])))
        print("  a", a[ys[0], xs[0]], "b", b[ys[0], xs[0]])
w1, r1 = acc([(0, 8)], 8)
w2, r2 = acc([(0, 8)], 8)
cmp(w2, w1, "whole twice")
p, r3 = acc([(0, 3), (3, 5)], 8)
cmp(p, w1, "parts vs whole")
p2, r4 = acc([(0, 1), (1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (7, 1)], 8)
cmp(p2, w1, "singles vs whole")
cmp(p2, p, "singles vs parts")
print(r1, r2, r3, r4)
