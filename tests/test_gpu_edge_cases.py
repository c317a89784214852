"""Edge cases of the radiance loop through the C ABI, each against the CPU checker with the same seed (tolerances
as stated in test_gpu_render.py): no lights, all three light kinds at once, degenerate and duplicate triangles,
the deepest trace depth the library accepts, a camera outside the world sphere, image sizes that are not a
multiple of the 8x8 ray-generation tile."""
import numpy as np
import pytest

from pathtracer_amd import meshes, scenes
from pathtracer_amd.scene import Scene, area_light, material, sphere_light
from test_gpu_render import gpu_render, same_seed_check

pytestmark = pytest.mark.gpu


def box_scene(lights, res=(40, 40), extra=(), extra_mats=(), **kw):
    mats, ms = scenes._box()
    return Scene(scenes._cam(res), lights, mats + list(extra_mats), ms + list(extra), **kw)


def test_a_scene_without_lights_renders_black():
    sc = box_scene([], res=(24, 24))
    fb, st = gpu_render(sc, 4, seed=1)
    assert (fb == 0).all() and st.stack_overflows == 0 and st.rays > 0
    same_seed_check(sc, 4)


def test_area_sphere_and_point_light_together_with_two_light_samples():
    lights = [area_light(**scenes.LIGHT), sphere_light((120, 300, 420), 30.0, (900000, 600000, 300000)),
              sphere_light((400, 150, 150), 1.0, (300000, 300000, 600000))]  # "l": a sphere of radius 1
    va, fa = meshes.uv_sphere(scenes.SPHERE_A["centre"], scenes.SPHERE_A["radius"], 12, 24)
    sc = box_scene(lights, extra=[meshes.expand(va, fa)], extra_mats=[material("specular", (1.0, 1.0, 1.0))])
    same_seed_check(sc, 8, lsamples=2)


def test_degenerate_and_duplicate_triangles_change_nothing():
    plain = box_scene([area_light(**scenes.LIGHT)], res=(32, 32))
    junk = np.array([[[100, 100, 100]] * 3,                              # a point
                     [[50, 50, 50], [60, 60, 60], [70, 70, 70]],         # a line
                     [[0, 0, 0], [0, 0, 0], [550, 0, 0]]], np.float32)   # two equal corners
    floor_again = plain.tri_verts[plain.tri_obj == len(plain.lights)][:2].copy()  # two triangles of the box, twice
    sc = box_scene([area_light(**scenes.LIGHT)], res=(32, 32), extra=[junk, floor_again],
                   extra_mats=[material("diffuse", (0.9, 0.1, 0.9)), material("diffuse", (0.7, 0.7, 0.7))])
    fb, ref = same_seed_check(sc, 8)
    base, _ = gpu_render(plain, 8, seed=7)
    np.testing.assert_allclose(fb, base, rtol=2e-3, atol=1e-4)


def test_deepest_trace_depth_between_two_mirrors():
    va, fa = meshes.uv_sphere(scenes.SPHERE_A["centre"], scenes.SPHERE_A["radius"], 12, 24)
    vb, fb_ = meshes.uv_sphere(scenes.SPHERE_B["centre"], scenes.SPHERE_B["radius"], 12, 24)
    sc = box_scene([area_light(**scenes.LIGHT)], res=(24, 24), extra=[meshes.expand(va, fa), meshes.expand(vb, fb_)],
                   extra_mats=[material("specular", (1.0, 1.0, 1.0)), material("dielectric", ior=1.5)], td=62)
    same_seed_check(sc, 4, frac=0.99)


def test_camera_outside_the_world_sphere_of_a_foggy_scene():
    sc = scenes.fog(res=(20, 14), lights="sphere")
    sc.world_radius = 420.0  # the camera sits 1030 from the centre: primary segments start outside the medium's sphere
    same_seed_check(sc, 2, frac=0.99)


@pytest.mark.parametrize("res", [(1, 1), (13, 7), (9, 16)])
def test_image_sizes_off_the_tile_grid(res):
    same_seed_check(scenes.cornell(res=res, spheres=False), 8)


def sphere(centre, radius, seg=(10, 20)):
    v, f = meshes.uv_sphere(centre, radius, *seg)
    return meshes.expand(v, f)


@pytest.mark.parametrize("ior", [1.0, 0.8, 2.4])
def test_dielectrics_with_unusual_indices(ior):
    """Only a dielectric with ior > 1 stops a shadow walk (cpp:757-759); through the others the walk goes on, surface by
    surface.  At ior <= 1 the Fresnel term of some events is exactly zero, and so is the weight of everything below the
    reflected branch: the reference still walks to the lights from there, the kernel does not emit a shadow walk whose
    weight is zero - same image and same path rays, fewer shadow rays (0.15 % here), so the query counts are compared
    for ior 2.4 only."""
    sc = box_scene([area_light(**scenes.LIGHT)], res=(28, 28), extra=[sphere((275, 200, 275), 130)],
                   extra_mats=[material("dielectric", ior=ior)], td=5)
    same_seed_check(sc, 8, frac=0.99, count_rays=ior > 1.0)


def test_the_parsers_flagless_sss_material_and_a_scattering_object_that_does_not_scatter():
    # "g sss" never raises a surface flag (myRenderer.cpp:573-598): paths end on it; a dielectricsss object with
    # ss = 0 only absorbs
    sss = material("sss", ior=1.3, sa=(0.001, 0.002, 0.003), ss=(0.01, 0.01, 0.01))
    dead_milk = material("dielectricsss", ior=1.3, kt=(1, 1, 1), sa=(0.002, 0.004, 0.008), ss=(0, 0, 0))
    sc = box_scene([area_light(**scenes.LIGHT)], res=(28, 28), extra=[sphere((160, 120, 300), 110), sphere((400, 120, 250), 110)],
                   extra_mats=[sss, dead_milk], td=4)
    same_seed_check(sc, 8, frac=0.99)


def test_glass_overlapping_milk_in_fog_with_several_samples_per_event():
    milk = material("dielectricsss", **scenes.MILK)
    sc = box_scene([area_light(**scenes.LIGHT), sphere_light((120, 300, 420), 20.0, (800000, 500000, 200000))],
                   res=(16, 12), extra=[sphere((230, 150, 275), 120), sphere((340, 150, 275), 120)],
                   extra_mats=[material("dielectric", ior=1.5), milk],
                   global_medium=material("medium", **scenes.FOG), td=3, mr=0.01)
    same_seed_check(sc, 2, hsamples=2, lsamples=2, frac=0.98)


def test_dense_fog_high_march_rate_and_steep_exponential_march():
    dense = material("medium", ss=(0.004, 0.006, 0.008), sa=(0.002, 0.001, 0.0005))
    sc = box_scene([area_light(**scenes.LIGHT)], res=(14, 10), global_medium=dense, td=2, mr=0.05, me=0.004)
    same_seed_check(sc, 2, frac=0.98)


def test_many_lights_several_hemisphere_and_light_samples():
    lights = [area_light(**scenes.LIGHT)]
    for k in range(6):
        a = 2.0 * np.pi * k / 6
        lights.append(sphere_light((275 + 200 * np.cos(a), 80 + 60 * k, 275 + 200 * np.sin(a)), 6.0 + 5 * k,
                                   (150000 * (1 + k % 3), 150000 * (1 + (k + 1) % 3), 150000 * (1 + (k + 2) % 3))))
    sc = box_scene(lights, res=(20, 20), extra=[sphere((275, 120, 275), 110)], extra_mats=[material("diffuse", (0.6, 0.6, 0.6))], td=2)
    same_seed_check(sc, 4, hsamples=3, lsamples=2)


@pytest.mark.parametrize("scale", [0.01, 4.0])
def test_the_scene_at_other_scales(scale):
    """Everything but TRACE_BIAS scales: the box 5.5 or 2 200 units wide.  The reference's bias of 0.001 is absolute and
    its hit points are single precision, so a larger scene pushes the bias below the rounding of the coordinates and
    which rays re-hit the surface they start on becomes a matter of the last bit on either side: measured 99.3 % of
    the pixels within tolerance at 4x, 98.96 % at 8x, 85 % at 30x (image means still within 0.2 %)."""
    base = scenes.cornell(res=(24, 24), spheres=True, sphere_segments=(10, 20))
    L = scenes.LIGHT
    light = area_light(P=tuple(scale * x for x in L["P"]), U=tuple(scale * x for x in L["U"]), V=tuple(scale * x for x in L["V"]),
                       power=tuple(scale * scale * x for x in L["power"]))
    from pathtracer_amd.scene import camera
    cam = camera(P=(275 * scale, 275 * scale, -755 * scale), at=(275 * scale, 275 * scale, 0), up=(0, 1, 0), fov=40.0, res=(24, 24), width=1.0)
    n_l = len(base.lights)
    ms = [base.tri_verts[base.tri_obj == n_l + g] * np.float32(scale) for g in range(len(base.materials) - n_l)]
    sc = Scene(cam, [light], base.materials[n_l:], ms, world_center=tuple(275.0 * scale for _ in range(3)), world_radius=1100.0 * scale, td=3)
    same_seed_check(sc, 8, frac=0.99)


def test_two_scenes_rendered_from_two_threads_at_once():
    """Scenes own their stream, queues and counters: two of them on one GPU, driven from two threads at the same time,
    give what each gives alone."""
    import threading
    from pathtracer_amd import renderer
    scs = [scenes.cornell(res=(64, 64), spheres=True, sphere_segments=(12, 24)), scenes.fog(res=(32, 24), lights="sphere")]
    spp = [16, 2]
    alone = [gpu_render(sc, n, seed=9)[0] for sc, n in zip(scs, spp)]
    contents = [renderer.wa_content(sc) for sc in scs]
    out, err = [None, None], []

    def work(k):
        try:
            for _ in range(3):
                out[k] = renderer.udp_renderer(contents[k], spp[k], 1, 1, 1, seed=9).render()
        except Exception as e:  # surfaces in the main thread below
            err.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    [c.close() for c in contents]
    assert not err, err
    for a, b in zip(alone, out):
        np.testing.assert_allclose(b, a, rtol=2e-4, atol=1e-6)


def _sheet(y, x0=150.0, x1=400.0, z0=150.0, z1=400.0):
    """A horizontal sheet (normal -y: faces the floor, like the light) spanning more than the light's quad."""
    v = np.array([[x0, y, z0], [x1, y, z0], [x1, y, z1], [x0, y, z1]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return meshes.expand(v, f)


@pytest.mark.parametrize("gap", [0.01, 0.04, 2.0])
def test_an_opaque_sheet_just_in_front_of_the_light(gap, monkeypatch):
    """Shadow walks that end on a light's quad stop searching a little short of it when nothing lies in front of the
    quad.  Here something does: an opaque sheet `gap` below the middle of the light, closer than the margin of a
    search (0.002 + 1e-4 |AB|) for the first two gaps.  The scene must fall back to the full search: same image as the
    step-by-step path and as the checker.  (A gap below TRACE_BIAS is left out: whether a walk then "arrives" through
    the sheet hangs on the last bits of single-precision distances in the reference itself.)"""
    sc = box_scene([area_light(**scenes.LIGHT)], res=(32, 32), extra=[_sheet(550.0 - gap, 240.0, 310.0, 240.0, 310.0)],
                   extra_mats=[material("diffuse", (0.5, 0.5, 0.5))])
    fast, st_fast = gpu_render(sc, 8, seed=2)
    same_seed_check(sc, 8, seed=2)
    monkeypatch.setenv("WAPT_GENERAL_SHADOW", "1")
    general, st_gen = gpu_render(sc, 8, seed=2)
    np.testing.assert_allclose(fast, general, rtol=2e-4, atol=1e-6)
    assert st_fast.rays == st_gen.rays


@pytest.mark.parametrize("gap", [0.0015, 0.01, 0.08])
def test_a_transparent_sheet_within_the_search_margin_of_the_light(gap, monkeypatch):
    """A non-opaque (scattering, flag-less sss) sheet close below the light: the walk passes it and its continued
    segment is shorter than the margin of the search (0.002 + 1e-4 |AB|).  The walk must still arrive: the fast
    path against the step-by-step path and the checker."""
    sss = material("sss", kt=(1.0, 1.0, 1.0), sa=(0.01, 0.01, 0.01), ss=(0.0, 0.0, 0.0))
    sc = box_scene([area_light(**scenes.LIGHT)], res=(32, 32), extra=[_sheet(550.0 - gap)], extra_mats=[sss])
    fast, st_fast = gpu_render(sc, 8, seed=4)
    assert fast.mean() > 0.05  # lit through the sheet
    same_seed_check(sc, 8, seed=4, frac=0.99)
    monkeypatch.setenv("WAPT_GENERAL_SHADOW", "1")
    general, st_gen = gpu_render(sc, 8, seed=4)
    np.testing.assert_allclose(fast, general, rtol=2e-4, atol=1e-6)
    assert st_fast.rays == st_gen.rays


def test_walks_that_stop_short_of_a_clear_light_equal_full_searches(monkeypatch):
    """WAPT_SHADOW_CUT=0 makes every walk search up to its end point: same image, same number of queries."""
    sc = scenes.cornell(res=(96, 96), spheres=True, sphere_segments=(16, 32))
    cut, st_cut = gpu_render(sc, 16, seed=6)
    monkeypatch.setenv("WAPT_SHADOW_CUT", "0")
    full, st_full = gpu_render(sc, 16, seed=6)
    np.testing.assert_allclose(cut, full, rtol=2e-4, atol=1e-6)
    assert st_cut.rays == st_full.rays
