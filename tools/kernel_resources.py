#!/usr/bin/env python3
"""Registers, LDS, scratch and the occupancy they allow for every kernel of the library, from the code-object metadata
(hipcc -S of pathtracer_amd/csrc/wapt_gpu.hip with the Makefile's flags).  gfx950: 512 VGPRs per lane slot of a SIMD,
allocated in steps of 8, at most 8 waves per SIMD, 160 KB of LDS per CU."""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "pathtracer_amd", "csrc", "wapt_gpu.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics",
                    "-fno-slp-vectorize", "--cuda-device-only", "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
meta = text[text.index("amdhsa.kernels:"):]
print(f"{'kernel':44s} {'vgpr':>5s} {'sgpr':>5s} {'spill':>6s} {'lds B':>7s} {'scratch B':>9s} {'threads':>8s} {'waves/SIMD by vgpr':>19s}")
for blk in re.split(r"\n  - ", meta)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "0"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip().split("(")[0]
    if not re.search(r"\.name:", blk):
        continue
    v = int(g("vgpr_count"))
    waves = min(8, 512 // max(8, (v + 7) // 8 * 8))
    print(f"{name:44s} {v:5d} {int(g('sgpr_count')):5d} {int(g('vgpr_spill_count')):6d} {int(g('group_segment_fixed_size')):7d} "
          f"{int(g('private_segment_fixed_size')):9d} {int(g('max_flat_workgroup_size')):8d} {waves:19d}")
