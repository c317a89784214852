"""Device-side BVH builder (Morton-code radix tree collapsed to 8-wide nodes on the GPU).

The acceleration structure must not change what a ray hits: closest hits through the device-built tree are
compared with the CPU checker exactly like those of the host-built tree (primID/geomID equal on >= 99.99 % of
rays, ties excused, |dt|/t <= 1e-5), and with the host-built tree ray by ray; same-seed images agree to the
tolerance of the image tests."""
import numpy as np
import pytest

import helpers
import oracle_api
from pathtracer_amd import renderer, scenes

pytestmark = pytest.mark.gpu


def _hits(scene, rays, builder):
    c = renderer.wa_content(scene, builder=builder)
    try:
        info = c.bvh_info()
        return c.traceFAST(rays), info
    finally:
        c.close()


@pytest.mark.parametrize("make", [
    lambda: scenes.cornell(res=(32, 32), spheres=False),
    lambda: scenes.cornell(res=(32, 32), spheres=True, sphere_segments=(24, 48)),
    lambda: scenes.dragon("chrome", res=(32, 32), n_along=300, n_around=80),
])
def test_device_tree_gives_the_same_hits(make):
    sc = make()
    rays = helpers.random_rays(sc, 200_000, seed=31)
    hd, info_d = _hits(sc, rays, "device")
    hh, info_h = _hits(sc, rays, "host")
    assert info_d.builder == 2 and info_h.builder == 1
    assert info_d.n_tris == info_h.n_tris
    same = (hd["prim"] == hh["prim"]) & (hd["obj"] == hh["obj"])
    dt = np.abs(hd["t"] - hh["t"]) <= 1e-5 * np.abs(hh["t"])
    print("nodes", info_d.n_nodes, info_h.n_nodes, "depth", info_d.max_depth, info_h.max_depth, "sah", info_d.sah_cost,
          info_h.sah_cost, "same", same.mean())
    # tree quality: the clustering build's surface-area cost stays near the binned-SAH tree's (dragon stand-in: 9.2 against
    # 9.2; a build whose last rounds paired clusters by position instead of by merged area measured 40 % more node visits)
    if info_h.n_tris > 10_000:
        assert info_d.sah_cost <= 1.08 * info_h.sah_cost, (info_d.sah_cost, info_h.sah_cost)
    assert (same | dt).mean() >= 0.9999     # a different triangle only where two are hit at the same distance
    assert dt.mean() >= 0.9999
    o = oracle_api.OracleScene(sc)
    ref = o.intersect(rays[:50_000], threads=16)
    frac, worst, n_bad, n_tie = helpers.compare_hits(hd[:50_000], ref)
    print("vs checker: agree", frac, "worst dt", worst, "bad", n_bad, "ties", n_tie)
    assert frac >= 0.9999 and worst <= 1e-5


def test_single_triangle_scene_builds_on_the_device():
    from pathtracer_amd.scene import Scene, material
    tri = np.array([[[0, 0, 5], [1, 0, 5], [0, 1, 5]]], np.float32)
    sc = scenes.cornell(res=(8, 8), spheres=False)
    rays = np.array([[0.2, 0.2, 0, 0.001, 0, 0, 1, 1e30], [2, 2, 0, 0.001, 0, 0, 1, 1e30]], np.float32)
    sc2 = Scene(sc.camera, [], [material("diffuse", (0.5, 0.5, 0.5))], [tri], td=1)
    hd, info = _hits(sc2, rays, "device")
    hh, _ = _hits(sc2, rays, "host")
    assert (hd["prim"] == hh["prim"]).all() and np.allclose(hd["t"], hh["t"])


def test_same_seed_image_through_the_device_tree():
    sc = scenes.dragon("glass", res=(48, 32), n_along=160, n_around=48, td=5)
    c = renderer.wa_content(sc, builder="device")
    try:
        fb = renderer.udp_renderer(c, 16, seed=4).render()
    finally:
        c.close()
    ref = oracle_api.OracleScene(sc).render(16, 1, 1, seed=4, threads=16)
    close = np.abs(fb - ref) <= 2e-3 * np.abs(ref) + 1e-4
    assert close.all(-1).mean() >= 0.995


def test_thousands_of_coincident_triangles_do_not_stall_the_clustering():
    """Every cluster of a run of identical boxes prefers the same partner, so a plain nearest-neighbour round merges
    one pair; the builder must notice and pair by position instead.  Same hits as through the host-built tree."""
    from pathtracer_amd.scene import Scene, material
    base = scenes.cornell(res=(8, 8), spheres=False)
    one = np.array([[[100, 100, 300], [200, 100, 300], [100, 200, 300]]], np.float32)
    many = np.repeat(one, 30_000, axis=0)
    sc = Scene(base.camera, base.lights, [material("diffuse", (0.5, 0.5, 0.5))] * 2,
               [base.tri_verts[base.tri_obj >= len(base.lights)].reshape(-1, 3, 3), many], td=1)
    rays = helpers.random_rays(sc, 20_000, seed=5)
    hd, info = _hits(sc, rays, "device")
    hh, _ = _hits(sc, rays, "host")
    assert info.n_tris == len(sc.tri_obj)
    # the duplicates tie with each other: compare distances and objects, not primitive numbers
    assert (hd["obj"] == hh["obj"]).mean() >= 0.9999
    np.testing.assert_allclose(hd["t"], hh["t"], rtol=1e-5)
