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
