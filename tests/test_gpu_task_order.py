"""The order of the tasks in the queues (pixel by pixel, the default, or sample by sample) and the way camera tasks come
about (made inside the depth-0 kernels, or written by the ray-generation pass) change no result: every sample draws
its numbers from (sample index, pixel), and a pixel's sum is a sum.  Image sizes with and without whole 8x8 tiles,
queues that hold the frame in one piece, in a few chunks, and less than one pixel's samples."""
import numpy as np
import pytest

from pathtracer_amd import scenes

pytestmark = pytest.mark.gpu


def _render(scene, spp, monkeypatch, env):
    from pathtracer_amd import renderer
    for k in ("WAPT_PIXEL_MAJOR", "WAPT_FUSE_CAMERA", "WAPT_RAD_CAP"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = renderer.wa_content(scene)
    try:
        r = renderer.udp_renderer(c, spp, 1, 1, 1, seed=5)
        fb = r.render()
        return fb, r.stats
    finally:
        c.close()


@pytest.mark.parametrize("res,spp,cap", [((64, 40), 24, None), ((64, 40), 24, "4096"), ((64, 40), 24, "16"),
                                          ((37, 21), 9, None), ((37, 21), 9, "1024"), ((16, 8), 300, "256")])
def test_orders_and_camera_paths_give_one_image(res, spp, cap, monkeypatch):
    sc = scenes.cornell(res=res, spheres=True, sphere_segments=(12, 24))
    base = {} if cap is None else {"WAPT_RAD_CAP": cap}
    ref, st0 = _render(sc, spp, monkeypatch, dict(base, WAPT_PIXEL_MAJOR="0", WAPT_FUSE_CAMERA="0"))
    assert np.isfinite(ref).all() and ref.mean() > 0
    for env in ({"WAPT_PIXEL_MAJOR": "1", "WAPT_FUSE_CAMERA": "0"}, {"WAPT_PIXEL_MAJOR": "0"}, {"WAPT_PIXEL_MAJOR": "1"}, {}):
        fb, st = _render(sc, spp, monkeypatch, dict(base, **env))
        # the same samples in another order of summation: float rounding only
        np.testing.assert_allclose(fb, ref, rtol=2e-4, atol=1e-5)
        assert int(st.rays) == int(st0.rays) and int(st.paths) == int(st0.paths)
        assert st.stack_overflows == 0


def test_media_scene_in_both_orders(monkeypatch):
    sc = scenes.fog(res=(24, 16), td=2)
    a, sa = _render(sc, 6, monkeypatch, {"WAPT_PIXEL_MAJOR": "0"})
    b, sb = _render(sc, 6, monkeypatch, {"WAPT_PIXEL_MAJOR": "1"})
    np.testing.assert_allclose(a, b, rtol=5e-4, atol=1e-5)
    assert int(sa.rays) + int(sa.queries_decided) == int(sb.rays) + int(sb.queries_decided)
