"""Same-seed parity at sizes where the rare branches are sampled (10^4-10^5 paths per material family, up to 10^7
closest-hit queries per scene), and converged images (4096 spp) against the CPU checker's own noise floor.

Tolerances as in test_gpu_render.py: same-seed renders agree on >= 99.5 % of the pixels (scenes with media: 99 %)
within 2e-3 relative + 1e-4 absolute and in the image means within 5e-4; a converged GPU image lies within 1.5 x the
statistical expectation derived from two independent checker renders."""
import numpy as np
import pytest

import helpers
import oracle_api
from pathtracer_amd import meshes, scenes
from pathtracer_amd.scene import Scene, area_light, material
from test_gpu_render import gpu_render, same_seed_check

pytestmark = pytest.mark.gpu

MESH = dict(n_along=300, n_around=80)  # 47 840 triangles: large enough for deep trees, small enough for the checker


def test_chrome_td6_230k_paths():
    same_seed_check(scenes.dragon("chrome", res=(160, 90), td=6, **MESH), 16)


def test_glass_td6_230k_paths():
    # (total internal reflection chains, the exit-side Fresnel term, glass blocking the shadow walks)
    same_seed_check(scenes.dragon("glass", res=(160, 90), td=6, **MESH), 16, frac=0.99)


def test_milk_41k_paths():
    same_seed_check(scenes.dragon("milk", res=(96, 54), td=3, **MESH), 8, frac=0.99)


def test_fog_sphere_lights_9k_paths_13m_rays():
    same_seed_check(scenes.fog(res=(64, 36), lights="sphere"), 4, frac=0.99)


def test_fog_area_light_9k_paths():
    same_seed_check(scenes.fog(res=(64, 36), lights="area"), 4, frac=0.99)


def _two_lights_facing(res=(48, 48), fog=True):
    """The box with a second area light on the floor facing the one in the ceiling, in a thin global medium: probes
    of one light's estimator hit the other light's quad (the directLightHit / MIS branch, cpp:519-552), shadow walks
    end on quads seen from below and from above, and both quads are Lambert reflectors lit by each other."""
    mats, ms = scenes._box()
    vb, fb = meshes.uv_sphere(scenes.SPHERE_B["centre"], scenes.SPHERE_B["radius"], 16, 32)
    ms.append(meshes.expand(vb, fb))
    mats.append(material("diffuse", (0.7, 0.7, 0.7)))
    lights = [area_light(**scenes.LIGHT),
              area_light(P=(350, 0.5, 200), U=(-150, 0, 0), V=(0, 0, 150), power=(1500000, 2500000, 4000000))]  # normal +y
    gm = material("medium", **scenes.FOG) if fog else None
    kw = dict(global_medium=gm) if fog else {}
    return Scene(scenes._cam(res), lights, mats, ms, td=3, me=0.0, mr=0.01, **kw)


def test_two_area_lights_facing_each_other_in_fog():
    same_seed_check(_two_lights_facing(), 2, frac=0.99)


def test_two_area_lights_facing_each_other_clear_air_two_samples_per_event():
    same_seed_check(_two_lights_facing(res=(96, 96), fog=False), 16, hsamples=2, lsamples=2)


def _converged(scene, name, spp=4096, ref_spp=1024, light_lum=5.0, sequential=False):
    """sequential: the checker's two renders draw from a std::mt19937_64 per thread instead of the keyed hash that
    it shares with the library - an estimate that has nothing in common with the GPU's but the estimator."""
    fb, st = gpu_render(scene, spp, seed=11)
    o = oracle_api.OracleScene(scene)
    a = o.render(ref_spp, seed=12, threads=16, sequential=sequential)
    b = o.render(ref_spp, seed=13, threads=16, sequential=sequential)
    ref = 0.5 * (a + b)
    mask = ref.mean(-1) < light_lum  # leave the directly visible light out, as the noise floor figure does
    noise = helpers.rel_rmse(a, b, mask)
    err = helpers.rel_rmse(fb, ref, mask)
    # var(gpu - ref) = s^2/spp + s^2/(2 ref_spp) against var(a - b) = 2 s^2/ref_spp
    expect = np.sqrt((1.0 / spp + 0.5 / ref_spp) / (2.0 / ref_spp))
    print(f"{name}: noise({ref_spp} vs {ref_spp}) {noise:.4f}  gpu{spp} vs checker{2 * ref_spp} {err:.4f}  expected {expect * noise:.4f}")
    assert err <= expect * noise * 1.5
    # image means: 0.5 %, or three standard errors of the mean of that many pixels where the image is noisier than that
    # (the fog scene with the area light: 9 % between two 1024-spp renders, light sampled at 1 / distance^2)
    mean_rtol = max(5e-3, 3.0 * expect * noise / np.sqrt(mask.sum()))
    np.testing.assert_allclose(fb[mask].mean(0), ref[mask].mean(0), rtol=mean_rtol)
    assert st.stack_overflows == 0


def test_converged_cornell_with_spheres():
    _converged(scenes.cornell(res=(64, 64), spheres=True, sphere_segments=(24, 48)), "cornell+spheres")


def test_converged_chrome():
    _converged(scenes.dragon("chrome", res=(96, 54), n_along=200, n_around=60), "chrome")


def test_converged_glass_td6():
    _converged(scenes.dragon("glass", res=(96, 54), td=6, n_along=200, n_around=60), "glass")


def test_converged_fog_sphere_lights_against_an_independent_generator():
    _converged(scenes.fog(res=(16, 12)), "fog, sphere lights", sequential=True)


def test_converged_fog_area_light_against_an_independent_generator():
    _converged(scenes.fog(res=(16, 12), lights="area"), "fog, area light", sequential=True)


def test_converged_milk_against_an_independent_generator():
    _converged(scenes.dragon("milk", res=(24, 14), n_along=120, n_around=40), "milk", sequential=True)
