"""Finds a pixel sample whose GPU value differs from the CPU checker's and prints both traces.
Needs pathtracer_amd/libwapt_dbg.so (wapt_gpu.hip built with -DWAPT_DEBUG_PIXEL)."""
import sys, os, ctypes as C, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api
from pathtracer_amd import scenes, abi
TD = 1
def scene():
    return scenes.cornell(res=(60, 60), spheres=False, td=TD)
if len(sys.argv) > 1:   # child: render one sample with the debug library, tracing pixel argv[1]
    import pathtracer_amd.capi as capi
    smp = int(sys.argv[2])
    from pathtracer_amd import renderer
    import torch
    c = renderer.wa_content(scene())
    r = renderer.udp_renderer(c, 4, 1, 1, 1, seed=7)
    acc = torch.zeros(60 * 60 * 3, device="cuda"); torch.cuda.synchronize()
    r.render_accumulate(acc.data_ptr(), smp, 1); torch.cuda.synchronize()
    sys.exit(0)
from pathtracer_amd import renderer
import torch
sc = scene()
c = renderer.wa_content(sc)
o = oracle_api.OracleScene(sc)
found = 0
for smp in range(4):
    r = renderer.udp_renderer(c, 4, 1, 1, 1, seed=7)
    acc = torch.zeros(60 * 60 * 3, device="cuda"); torch.cuda.synchronize()
    r.render_accumulate(acc.data_ptr(), smp, 1); torch.cuda.synchronize()
    fb = acc.cpu().numpy().reshape(60, 60, 3)
    ref = o.render(4, seed=7, threads=16, sample_begin=smp, sample_count=1) * 4
    bad = ~(np.abs(fb - ref) <= 2e-3 * np.abs(ref) + 1e-5).all(-1)
    js, is_ = np.nonzero(bad)
    for j, i in list(zip(js, is_))[:2]:
        print(f"=== sample {smp} pixel ({i},{j}) gpu {fb[j,i]} ref {ref[j,i]}", flush=True)
        p = abi.RenderParams(); p.psamples, p.hsamples, p.lsamples, p.max_depth, p.me, p.mr, p.seed = 4, 1, 1, TD, 0.0, 0.01, 7
        L = oracle_api.lib()
        L.oracle_debug_path.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        buf = np.zeros((64, 8), np.float32)
        n = L.oracle_debug_path(o._h, C.byref(p), int(i), int(j), smp, buf.ctypes.data, 64)
        for k in range(n):
            print("   oracle", " ".join(f"{x:.6f}" for x in buf[k]), flush=True)
        env = dict(os.environ, WAPT_DEBUG_PIXEL=str(j * 60 + i), WAPT_LIB="libwapt_dbg.so")
        out = subprocess.run([sys.executable, __file__, str(j * 60 + i), str(smp)], env=env, capture_output=True, text=True, timeout=120)
        print(out.stdout[-3000:], out.stderr[-500:], flush=True)
        found += 1
    if found >= 4:
        break
