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


@pytest.mark.parametrize("name,spp,budget_gb", [("chrome", 16, "0.05"), ("glass", 8, "0.05"), ("milk", 4, "0.1"), ("fog", 2, "0.1")])
def test_queue_memory_budget_changes_no_result(name, spp, budget_gb, monkeypatch):
    """WAPT_QUEUE_BUDGET_GB bounds the memory of the work queues (the frame is then rendered in more, smaller steps): the
    same samples, so the same image to the order of the float additions and the same number of closest-hit queries; the
    allocation stays within the budget."""
    sc = {"chrome": lambda: scenes.dragon("chrome", res=(320, 180), n_along=200, n_around=60),
          "glass": lambda: scenes.dragon("glass", res=(320, 180), n_along=200, n_around=60, td=5),
          "milk": lambda: scenes.dragon("milk", res=(160, 90), n_along=200, n_around=60),
          "fog": lambda: scenes.fog(res=(160, 90))}[name]()
    monkeypatch.delenv("WAPT_QUEUE_BUDGET_GB", raising=False)
    ref, st0 = _render(sc, spp, monkeypatch, {})
    fb, st = _render(sc, spp, monkeypatch, {"WAPT_QUEUE_BUDGET_GB": budget_gb})
    assert np.isfinite(ref).all() and ref.mean() > 0
    assert 0 < int(st.queue_bytes) <= float(budget_gb) * 1e9 and int(st.queue_bytes) < int(st0.queue_bytes)
    assert int(st.rays) + int(st.queries_decided) == int(st0.rays) + int(st0.queries_decided)
    assert st.stack_overflows == 0
    ok = np.abs(fb - ref) <= 5e-4 * np.abs(ref) + 1e-5
    assert ok.mean() >= 0.9999, ok.mean()
    np.testing.assert_allclose(fb.mean((0, 1)), ref.mean((0, 1)), rtol=1e-4)
