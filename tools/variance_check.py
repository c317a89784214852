#!/usr/bin/env python3
"""Does the variance of a render fall as 1/spp?  Eight independent 1024-spp renders and one of 4096 spp, all on the
GPU: the relRMSE of the 4096-spp render against the mean of the eight, over what 1/spp scaling predicts from the
spread of the eight (debugging aid for the converged-image tests)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import renderer, scenes  # noqa: E402

sc = scenes.cornell(res=(64, 64), spheres=True, sphere_segments=(24, 48))
c = renderer.wa_content(sc)
imgs = [renderer.udp_renderer(c, 1024, 1, 1, 1, seed=20 + k).render().astype(np.float64) for k in range(8)]
big = renderer.udp_renderer(c, 4096, 1, 1, 1, seed=11).render().astype(np.float64)
big2 = renderer.udp_renderer(c, 4096, 1, 1, 1, seed=12).render().astype(np.float64)
c.close()
ref = np.mean(imgs, 0)
mask = ref.mean(-1) < 5.0
m = ref[mask].mean()
s2 = np.mean([((a - ref)[mask] ** 2).mean() for a in imgs]) * 8.0 / 7.0   # per-pixel variance of one 1024-spp render, averaged
print("sigma(1024 spp)/mean", np.sqrt(s2) / m)
for name, b in (("seed 11", big), ("seed 12", big2)):
    err = np.sqrt(((b - ref)[mask] ** 2).mean()) / m
    expect = np.sqrt(s2 * (1024.0 / 4096.0 + 1.0 / 8.0)) / m
    print(name, "4096 spp vs mean of eight:", err, "expected", expect, "ratio", err / expect)
d = np.sqrt(((big - big2)[mask] ** 2).mean()) / m
print("4096 vs 4096:", d, "expected", np.sqrt(s2 * 2 * 1024 / 4096) / m, "ratio", d / (np.sqrt(s2 * 2 * 1024 / 4096) / m))
