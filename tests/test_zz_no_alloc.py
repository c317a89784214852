"""wapt_intersect_device_static - the closest-hit batch for callers that cannot allocate device memory - against
wapt_intersect_device on the same rays, through the C ABI.  Integer and bit-level comparison: both entry points run
the same traversal kernel, so prim, obj, the bits of t and of Ng are equal.  Two calls in a row share the GPU's
static scratch array; the second one is smaller, so anything the first left behind would show."""
import ctypes as C

import numpy as np
import pytest

import helpers
from pathtracer_amd import scenes

pytestmark = pytest.mark.gpu


def _both(capi, torch, c, rays):
    n = len(rays)
    d_rays = torch.from_numpy(rays).to("cuda:0")
    d_a = torch.zeros(n * 6, dtype=torch.float32, device="cuda:0")
    d_b = torch.full((n * 6,), 7.0, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    capi.check(capi.wapt_intersect_device(c._h, d_rays.data_ptr(), n, d_a.data_ptr(), None, None))
    capi.check(capi.wapt_intersect_device_static(c._h, d_rays.data_ptr(), n, d_b.data_ptr()))
    torch.cuda.synchronize()
    return d_a.cpu().numpy().view(np.uint32), d_b.cpu().numpy().view(np.uint32)


def test_static_scratch_entry_point_equals_the_allocating_one():
    import torch
    from pathtracer_amd import capi, renderer
    sc = scenes.dragon("chrome", res=(64, 64), n_along=300, n_around=80)
    c = renderer.wa_content(sc)
    try:
        big = helpers.random_rays(sc, 400_000, seed=21)
        a, b = _both(capi, torch, c, big)
        assert (a.reshape(-1, 6)[:, 1] != 0xffffffff).mean() > 0.3  # a fair share of the rays hit something
        np.testing.assert_array_equal(a, b)
        small = helpers.random_rays(sc, 1_000, seed=22)
        a, b = _both(capi, torch, c, small)
        np.testing.assert_array_equal(a, b)
        one = helpers.random_rays(sc, 3, seed=23)
        a, b = _both(capi, torch, c, one)
        np.testing.assert_array_equal(a, b)
        # nothing to do, and too much: neither touches the buffers
        assert capi.wapt_intersect_device_static(c._h, None, 0, None) == 0
        d = torch.zeros(8, dtype=torch.float32, device="cuda:0")
        rc = capi.wapt_intersect_device_static(c._h, d.data_ptr(), (1 << 30) + 1, d.data_ptr())
        assert rc == 5 and b"2^30" in capi.wapt_last_error()  # WAPT_E_LIMIT
    finally:
        c.close()


def test_static_scratch_is_shared_by_the_scenes_of_one_gpu():
    """A second scene on the same GPU uses the same array right after the first: results still equal."""
    import torch
    from pathtracer_amd import capi, renderer
    s1 = scenes.cornell(res=(16, 16), spheres=True, sphere_segments=(16, 32))
    s2 = scenes.dragon("chrome", res=(16, 16), n_along=120, n_around=40)
    c1, c2 = renderer.wa_content(s1), renderer.wa_content(s2)
    try:
        for c, sc, n, seed in ((c1, s1, 50_000, 31), (c2, s2, 20_000, 32), (c1, s1, 500, 33)):
            a, b = _both(capi, torch, c, helpers.random_rays(sc, n, seed=seed))
            np.testing.assert_array_equal(a, b)
    finally:
        c1.close()
        c2.close()
