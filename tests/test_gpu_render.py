"""Image parity of the wavefront radiance loop (through the C ABI) against the CPU checker.

Both sides draw the same keyed random numbers, so renders with the same seed agree sample by sample up
to single-precision arithmetic; a handful of paths may take a different branch where a float comparison
flips, which moves single pixels by at most one sample.  Tolerances (float32 path, stated here):
  * same-seed renders: at least 99.5 % of pixels within 2e-3 relative (+1e-4 absolute), and the image
    means within 5e-4 relative;
  * independent seeds at high sample counts: relRMSE within 1.5x the statistical noise between two
    independent renders of the checker itself.
"""
import numpy as np
import pytest

import helpers
import oracle_api
from pathtracer_amd import scenes

pytestmark = pytest.mark.gpu


def gpu_render(scene, psamples, hsamples=1, lsamples=1, seed=0, **kw):
    from pathtracer_amd import renderer
    c = renderer.wa_content(scene)
    try:
        r = renderer.udp_renderer(c, psamples, hsamples, 1, lsamples, seed=seed, **kw)
        fb = r.render()
        return fb, r.stats
    finally:
        c.close()


def same_seed_check(scene, psamples, hsamples=1, lsamples=1, seed=7, frac=0.995, count_rays=True, **kw):
    fb, st = gpu_render(scene, psamples, hsamples, lsamples, seed, **kw)
    o = oracle_api.OracleScene(scene)
    ref = o.render(psamples, hsamples, lsamples, seed=seed, threads=16,
                   td=kw.get("maxTraceDepth"), me=kw.get("rmdde"), mr=kw.get("rmdsr"))
    assert np.isfinite(fb).all()
    close = np.abs(fb - ref) <= 2e-3 * np.abs(ref) + 1e-4
    pix_ok = close.all(-1).mean()
    mg, mr_ = fb.reshape(-1, 3).mean(0), ref.reshape(-1, 3).mean(0)
    print(f"pixels within tol {pix_ok:.5f}; mean gpu {mg} ref {mr_}; rays gpu {st.rays} ref {o.rays}; "
          f"relRMSE {helpers.rel_rmse(fb, ref):.2e}")
    assert pix_ok >= frac
    np.testing.assert_allclose(mg, mr_, rtol=5e-4)
    assert st.stack_overflows == 0
    # closest-hit query counts agree to the handful of paths that branch differently
    # (probe queries that the kernel answers without a search are the checker's traceFAST calls all the same)
    if count_rays:
        assert abs(int(st.rays) + int(st.queries_decided) - int(o.rays)) <= 1e-3 * o.rays + 16
    return fb, ref


def test_cornell_empty_same_seed():
    same_seed_check(scenes.cornell(res=(60, 60), spheres=False), 32)


def test_cornell_spheres_same_seed_h2_l2():
    same_seed_check(scenes.cornell(res=(48, 48), spheres=True, sphere_segments=(16, 32)), 16, hsamples=2, lsamples=2)


def test_dragon_chrome_same_seed():
    sc = scenes.dragon("chrome", res=(64, 36), n_along=200, n_around=60)
    same_seed_check(sc, 16)


def test_dragon_glass_same_seed():
    sc = scenes.dragon("glass", res=(64, 36), n_along=200, n_around=60, td=6)
    same_seed_check(sc, 16, frac=0.99)


def test_dragon_milk_same_seed():
    sc = scenes.dragon("milk", res=(32, 18), n_along=120, n_around=40, td=3)
    same_seed_check(sc, 4, frac=0.99)


def test_fog_sphere_lights_same_seed():
    same_seed_check(scenes.fog(res=(24, 16), lights="sphere"), 2, frac=0.99)


def test_fog_area_light_exponential_march_same_seed():
    same_seed_check(scenes.fog(res=(24, 16), lights="area"), 2, frac=0.99, rmdde=0.002)


def test_orthographic_camera_and_depth_zero():
    sc = scenes.cornell(res=(32, 32), spheres=False)
    sc.camera.type = 1
    sc.camera.sensor_width = 500.0
    same_seed_check(sc, 8, maxTraceDepth=0)


def test_converged_image_matches_checker_statistically():
    sc = scenes.cornell(res=(60, 60), spheres=False)
    fb, _ = gpu_render(sc, 4096, seed=11)
    o = oracle_api.OracleScene(sc)
    a = o.render(1024, seed=12, threads=16)
    b = o.render(1024, seed=13, threads=16)
    lum = a.mean(-1)
    mask = lum < 5.0  # leave the directly visible light out, as the noise floor figure does
    noise = helpers.rel_rmse(a, b, mask)            # two independent 1024-spp checker renders
    err = helpers.rel_rmse(fb, 0.5 * (a + b), mask)  # 4096-spp GPU render against their mean (2048 spp)
    print(f"noise(1024 vs 1024) {noise:.4f}  gpu4096 vs checker2048 {err:.4f}")
    # var(gpu - mean) = s^2/4096 + s^2/2048 against var(a-b) = 2 s^2/1024: expected ratio sqrt(3/8) = 0.61
    assert err <= 0.61 * noise * 1.5
    np.testing.assert_allclose(fb[mask].mean(0), (0.5 * (a + b))[mask].mean(0), rtol=5e-3)


def test_sample_sharding_adds_up():
    """spp sharding: two half renders accumulated into one buffer equal the full render (same samples)."""
    import torch
    from pathtracer_amd import renderer
    sc = scenes.cornell(res=(40, 40), spheres=False)
    full, _ = gpu_render(sc, 8, seed=5)
    c = renderer.wa_content(sc)
    r = renderer.udp_renderer(c, 8, 1, 1, 1, seed=5)
    acc = torch.zeros(40 * 40 * 3, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    r.render_accumulate(acc.data_ptr(), 0, 4)
    r.render_accumulate(acc.data_ptr(), 4, 4)
    torch.cuda.synchronize()
    got = (acc / 8).cpu().numpy().reshape(40, 40, 3)
    c.close()
    np.testing.assert_allclose(got, full, rtol=1e-5, atol=1e-6)


def test_chunked_queues_give_the_same_image(monkeypatch):
    """Tiny queue capacities force the depth-first chunking of the scheduler; the image must not change."""
    sc = scenes.fog(res=(16, 12), lights="area")
    ref, _ = gpu_render(sc, 2, seed=3)
    monkeypatch.setenv("WAPT_RAD_CAP", "8192")
    monkeypatch.setenv("WAPT_SHADOW_CAP", "16384")
    monkeypatch.setenv("WAPT_PROBE_CAP", "16384")
    got, st = gpu_render(sc, 2, seed=3)
    assert st.stack_overflows == 0
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-6)


def test_errors_are_reported_not_thrown():
    from pathtracer_amd import capi, renderer
    sc = scenes.cornell(res=(8, 8), spheres=False)
    c = renderer.wa_content(sc)
    r = renderer.udp_renderer(c, 0)
    with pytest.raises(capi.WaptError):
        r.render()
    r = renderer.udp_renderer(c, 1, maxTraceDepth=100)
    with pytest.raises(capi.WaptError):
        r.render()
    c.close()


def test_cli_renders_the_same_image_as_the_python_mirror(tmp_path):
    """myRenderer -i scene.waff --psamples N -bmp out.bmp (the reference's command line) against
    udp_renderer.render + tonemap + write_bmp on the same scene file and seed: identical BMP files."""
    import os
    import shutil
    import subprocess
    from pathtracer_amd import meshes, renderer
    from pathtracer_amd.scene import Scene
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "pathtracer_amd", "myRenderer")
    for f in ("cornellBox.obj", "cornellBoxLeftWall.obj", "cornellBoxRightWall.obj"):
        shutil.copy(os.path.join(scenes.ASSET_DIR, f), tmp_path / f)
    va, fa = meshes.uv_sphere(scenes.SPHERE_A["centre"], scenes.SPHERE_A["radius"], 16, 32)
    meshes.write_obj(tmp_path / "sphereA.obj", va, fa)
    vb, fb = meshes.uv_sphere(scenes.SPHERE_B["centre"], scenes.SPHERE_B["radius"], 16, 32)
    meshes.write_fast(tmp_path / "sphereB.fast", vb, fb)
    (tmp_path / "cornellBox.waff").write_text(
        "v 0\nfrom  275 275 -755\nat    275 275 0\nup 0 1 0\nangle 40\nhither 0.1\nresolution 72 56\nwidth 1\n\n"
        "td 3\nme 0.0\nmr 0.01\n\nla 200 550 200  150 0 0  0 0 150  4000000 4000000 4000000\n\n"
        "g diffuse  0.7 0.7 0.7  cornellBox.obj\ng diffuse  0.7 0.07 0.07  cornellBoxLeftWall.obj\n"
        "g diffuse  0.07 0.7 0.07  cornellBoxRightWall.obj\ng specular  1.0 1.0 1.0  sphereA.obj\n"
        "g diffuse 0.7 0.7 0.7 sphereB.fast\n")
    out = subprocess.run([cli, "-i", "cornellBox.waff", "--psamples", "8", "--hsamples", "1", "--lsamples", "1",
                          "-bmp", "cli.bmp", "--seed", "4"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "Render time" in out.stdout
    assert (tmp_path / "debug_content.obj").exists()
    sc = Scene.from_waff(str(tmp_path / "cornellBox.waff"))
    c = renderer.wa_content(sc)
    fbuf = renderer.udp_renderer(c, 8, 1, 1, 1, seed=4).render()
    c.close()
    renderer.write_bmp(str(tmp_path / "py.bmp"), renderer.tonemap(fbuf))
    a, b = open(tmp_path / "cli.bmp", "rb").read(), open(tmp_path / "py.bmp", "rb").read()
    assert len(a) == 138 + 72 * 56 * 4
    pa = np.frombuffer(a, np.uint8, offset=138).astype(int)
    pb = np.frombuffer(b, np.uint8, offset=138).astype(int)
    # float atomics add in a different order from run to run: allow the last bit of a few pixels to differ
    assert a[:138] == b[:138] and np.abs(pa - pb).max() <= 1 and (pa != pb).mean() < 0.01


@pytest.mark.parametrize("name", ["cornell_spheres", "dragon_chrome", "dragon_glass", "fog_area"])
def test_shadow_walk_shortcuts_equal_the_general_path(name, monkeypatch):
    """Shadow walks that the kernel finishes from registers (opaque surface before the end point, clear miss, hit on
    the sampled light's own quad) against the same walks sent through the step-by-step path of visibility()
    (WAPT_GENERAL_SHADOW=1, read at scene creation): same image up to the order of the float atomics, same number
    of closest-hit queries.  The glass scene's walks are one search each as well (front faces of glass end a walk, back
    faces are passed, cpp:757-759)."""
    sc, spp, kw = {
        "cornell_spheres": (scenes.cornell(res=(96, 96), spheres=True, sphere_segments=(16, 32)), 16, dict(hsamples=2, lsamples=2)),
        "dragon_chrome": (scenes.dragon("chrome", res=(128, 72), n_along=200, n_around=60), 16, {}),
        "dragon_glass": (scenes.dragon("glass", res=(64, 36), n_along=200, n_around=60, td=6), 16, {}),
        "fog_area": (scenes.fog(res=(24, 16), lights="area"), 2, {}),
    }[name]
    fast, st_fast = gpu_render(sc, spp, seed=3, **kw)
    monkeypatch.setenv("WAPT_GENERAL_SHADOW", "1")
    general, st_gen = gpu_render(sc, spp, seed=3, **kw)
    np.testing.assert_allclose(fast, general, rtol=2e-4, atol=1e-6)
    if name == "dragon_glass":
        # a walk that starts INSIDE the glass (floor points under the dragon) passes a back face first: the step-by-step
        # path searches once per segment, the one-search kernel passes such faces inside its single search
        assert 0 <= st_gen.rays - st_fast.rays <= 1e-4 * st_gen.rays
    else:
        assert st_fast.rays == st_gen.rays


def _open_milk_sheet_scene(res=(32, 32)):
    """The box with an OPEN sheet of the milk material right in front of the camera: every path enters the medium
    through the sheet and then travels to a wall, so the marched segments are several times longer than the sheet's
    box diagonal - the bound that sizes the launches of scenes with media (closed objects keep it, an open mesh
    need not)."""
    from pathtracer_amd import meshes
    from pathtracer_amd.scene import Scene, area_light, camera, material
    mats, ms = scenes._box()
    v = np.array([[225, 225, 100], [225, 325, 100], [325, 325, 100], [325, 225, 100]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)  # geometric normal -z: towards the camera
    ms.append(meshes.expand(v, f))
    mats.append(material("dielectricsss", **scenes.MILK))
    cam = camera(P=(275, 275, 0), at=(275, 275, 100), up=(0, 1, 0), fov=40.0, res=res, width=1.0)
    return Scene(cam, [area_light(**scenes.LIGHT)], mats, ms, td=3, me=0.0, mr=0.05)


def test_march_samples_beyond_the_sizing_bound_are_marched_in_slices(monkeypatch):
    """Segments inside an open mesh exceed the bound the queues were sized by; the exact count of a step's march
    samples is known before anything is emitted, and the samples are then marched in slices that fit the shadow and
    probe queues.  Small queues force several slices per step; the image must not change, and must match the checker."""
    sc = _open_milk_sheet_scene()
    ref_gpu, st0 = gpu_render(sc, 2, seed=5)
    monkeypatch.setenv("WAPT_PROBE_CAP", str(1 << 16))
    monkeypatch.setenv("WAPT_SHADOW_CAP", str(1 << 16))
    sliced, st1 = gpu_render(sc, 2, seed=5)
    np.testing.assert_allclose(sliced, ref_gpu, rtol=2e-4, atol=1e-6)
    assert st1.rays == st0.rays and st1.stack_overflows == 0
    monkeypatch.delenv("WAPT_PROBE_CAP")
    monkeypatch.delenv("WAPT_SHADOW_CAP")
    same_seed_check(sc, 2, seed=5, frac=0.99)


@pytest.mark.parametrize("name", ["fog_sphere", "fog_area", "dragon_milk"])
def test_probe_rays_aimed_at_their_light_equal_plain_searches(name, monkeypatch):
    """directLightHit / directLightingSingleScatter probes only ask whether the closest hit is one given light: the
    kernel tests the light's quad first, searches no further than the quad, and skips rays that miss it (or lights
    without geometry).  WAPT_GENERAL_PROBE=1 gives every probe the plain closest-hit search: same image up to the
    order of the float atomics, and the same number of queries (searched + decided)."""
    sc, spp = {
        "fog_sphere": (scenes.fog(res=(24, 16), lights="sphere"), 2),
        "fog_area": (scenes.fog(res=(24, 16), lights="area"), 2),
        "dragon_milk": (scenes.dragon("milk", res=(32, 18), n_along=120, n_around=40, td=3), 4),
    }[name]
    fast, st_fast = gpu_render(sc, spp, seed=3)
    monkeypatch.setenv("WAPT_GENERAL_PROBE", "1")
    general, st_gen = gpu_render(sc, spp, seed=3)
    np.testing.assert_allclose(fast, general, rtol=2e-4, atol=1e-6)
    assert st_gen.queries_decided == 0 and st_fast.queries_decided > 0
    assert st_fast.rays + st_fast.queries_decided == st_gen.rays
