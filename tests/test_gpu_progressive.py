"""Progressive interface (reset / setIterationCount / setRenderResolution / setRegion / abort / generateTiles of
udp_renderer, src/unidirectionalPathTracer.cpp:68-76, 141-161, 907-972) through the C ABI.

The running average over n iterations of p samples must equal one render of n*p samples with the same seed (the
iterations draw consecutive blocks of the same keyed sample indices); only float rounding of the blend differs:
tolerance 1e-4 relative + 1e-6 absolute.  Lower render resolutions are checked against the CPU checker rendering a
scene of that resolution."""
import threading
import time

import numpy as np
import pytest

import oracle_api
from pathtracer_amd import capi, renderer, scenes

pytestmark = pytest.mark.gpu


def small_scene(res=(48, 40)):
    return scenes.cornell(res=res, spheres=True, sphere_segments=(12, 24))


def test_iterations_average_to_one_render():
    sc = small_scene()
    c = renderer.wa_content(sc)
    try:
        whole = renderer.udp_renderer(c, 24, seed=5).render()
        r = renderer.udp_renderer(c, 8, seed=5)
        r.reset()
        buf = np.zeros((40, 48, 3), np.float32)
        for it in range(3):
            r.setIterationCount(it)
            r.render(buf)
        np.testing.assert_allclose(buf, whole, rtol=1e-4, atol=1e-6)
        # reset() starts over: iteration 0 alone is a plain 8-sample render
        r.reset()
        r.render(buf)
        first = renderer.udp_renderer(c, 8, seed=5).render()
        np.testing.assert_allclose(buf, first, rtol=1e-5, atol=1e-7)
        r.close()
    finally:
        c.close()


def test_region_renders_only_the_region():
    sc = small_scene()
    c = renderer.wa_content(sc)
    try:
        whole = renderer.udp_renderer(c, 8, seed=9).render()
        r = renderer.udp_renderer(c, 8, seed=9)
        r.reset()
        r.setRegion(10, 31, 4, 22)
        buf = r.render()
        np.testing.assert_allclose(buf[4:22, 10:31], whole[4:22, 10:31], rtol=1e-5, atol=1e-7)
        mask = np.ones((40, 48), bool)
        mask[4:22, 10:31] = False
        assert (buf[mask] == 0).all()
        with pytest.raises(capi.WaptError):
            r.setRegion(0, 49, 0, 40)
        # tiles of generateTiles cover the image: rendering them one after the other gives the whole image
        r.reset()
        for (l, rr, b, t) in r.generateTiles(1):
            r.setRegion(l, rr, b, t)
            buf = r.render()
        np.testing.assert_allclose(buf, whole, rtol=1e-5, atol=1e-7)
        r.close()
    finally:
        c.close()


def test_lower_resolution_replicates_pixels_and_matches_checker():
    sc = small_scene(res=(48, 40))
    c = renderer.wa_content(sc)
    try:
        r = renderer.udp_renderer(c, 8, seed=3)
        r.setRenderResolution(1)  # 24 x 20, every pixel shown as a 2 x 2 block
        buf = r.render()
        blocks = buf.reshape(20, 2, 24, 2, 3)
        assert (blocks == blocks[:, :1, :, :1]).all()
        low = buf[::2, ::2]
        ref = oracle_api.OracleScene(small_scene(res=(24, 20))).render(8, 1, 1, seed=3, threads=16)
        close = np.abs(low - ref) <= 2e-3 * np.abs(ref) + 1e-4
        assert close.all(-1).mean() >= 0.995
        r.setRenderResolution(0)
        r.reset()
        full = r.render()
        np.testing.assert_allclose(full, renderer.udp_renderer(c, 8, seed=3).render(), rtol=1e-5, atol=1e-7)
        r.close()
    finally:
        c.close()


def test_abort_ends_an_iteration_and_leaves_the_average():
    sc = scenes.cornell(res=(512, 512), spheres=True, sphere_segments=(16, 32))
    c = renderer.wa_content(sc)
    try:
        r = renderer.udp_renderer(c, 16384, seed=2)  # a long iteration: 4.3 G paths
        r.reset()
        out = {}

        def run():
            try:
                r.render()
                out["rc"] = 0
            except capi.WaptError as e:
                out["rc"] = e.code
        th = threading.Thread(target=run)
        th.start()
        time.sleep(0.5)
        r.abort()
        th.join(200)
        assert not th.is_alive()
        assert out["rc"] == 6, "the iteration should have been cut short"
        # nothing was blended in, and the handle keeps working: a short iteration from black equals a plain render
        r.close()
        r = renderer.udp_renderer(c, 4, seed=2)
        r.reset()
        np.testing.assert_allclose(r.render(), renderer.udp_renderer(c, 4, seed=2).render(), rtol=1e-5, atol=1e-7)
        r.close()
    finally:
        c.close()


def test_a_progressive_handle_outlives_its_scene_safely():
    """Closing the content first cuts its renderers' progressive handles loose: every later call on such a handle is
    an error return (no access to the freed scene), and closing the renderer afterwards is fine."""
    c = renderer.wa_content(small_scene())
    r = renderer.udp_renderer(c, 2, seed=1)
    r.reset()
    r.render()
    c.close()
    with pytest.raises(capi.WaptError) as e:
        r.render()
    assert "destroyed" in str(e.value)
    with pytest.raises(capi.WaptError):
        r.reset()
    with pytest.raises(capi.WaptError):
        r.setRegion(0, 4, 0, 4)
    r.close()
    r.close()


def test_profile_is_refused_not_ignored_on_the_progressive_interface():
    c = renderer.wa_content(small_scene())
    try:
        r = renderer.udp_renderer(c, 2, seed=1)
        r.reset()
        with pytest.raises(ValueError):
            r.render(profile=True)
        r.close()
    finally:
        c.close()
