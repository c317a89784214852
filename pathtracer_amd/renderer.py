"""Host-side mirror of the reference's renderer interface on top of libwapt.so.

``wa_content`` stands for the scene container the reference builds in loadContentFromFile_waff
(src/myRenderer.cpp:649-683, src/content.cpp:13-40) and ``udp_renderer`` for
udp_renderer::udp_renderer / ::render (src/unidirectionalPathTracer.cpp:26-66, 78-103): same argument
order and meaning, the frame buffer is float RGB, row 0 at the bottom.
"""
import ctypes as C

import numpy as np

from . import abi, capi
from .scene import Scene


class wa_content:
    """Scene resident on one GPU (BVH built by the library at construction)."""

    def __init__(self, scene: Scene, device: int = 0, builder: str = "auto"):
        """builder: "auto" (the device builder), "host" (binned SAH on the CPU) or "device"."""
        self.scene = scene
        self.device = device
        self._h = C.c_void_p()
        d = scene.desc()
        b = {"auto": abi.BUILDER_AUTO, "host": abi.BUILDER_HOST_SAH, "device": abi.BUILDER_DEVICE_LBVH}[builder]
        capi.check(capi.wapt_scene_create_with(C.byref(d), device, b, C.byref(self._h)))

    def getImageResolution(self):
        return self.scene.resolution

    def bvh_info(self):
        info = abi.BvhInfo()
        capi.check(capi.wapt_scene_bvh_info(self._h, C.byref(info)))
        return info

    def traceFAST(self, rays):
        """Closest hits of a batch of rays (wa_content::traceFAST, src/content.cpp:91-123).

        rays: float32 [n,8] = org.xyz, tnear, dir.xyz, tfar.  Returns a structured array (t, prim, obj, Ng)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        capi.check(capi.wapt_intersect(self._h, rays.ctypes.data, len(rays), hits.ctypes.data))
        return hits

    def close(self):
        """Frees the scene.  Progressive handles of renderers on this content stay valid to close, nothing else:
        the library cuts them loose from the scene before it goes."""
        if self._h:
            capi.wapt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


HIT_DTYPE = np.dtype([("t", np.float32), ("prim", np.int32), ("obj", np.int32), ("Ng", np.float32, 3)])


class udp_renderer:
    def __init__(self, content: wa_content, psamples=1, hsamples=1, sphereSamples=1, lsamples=1, globalShader=None,
                 maxTraceDepth=None, rmdde=None, rmdsr=None, seed=0):
        # sphereSamples is accepted and ignored, as in the reference (its only use is shadowed by a local, cpp:591)
        # globalShader travels with the scene description; it is accepted here for signature parity
        self.content = content
        sc = content.scene
        self.params = abi.RenderParams()
        self.params.psamples, self.params.hsamples, self.params.lsamples = psamples, hsamples, lsamples
        self.params.max_depth = sc.td if maxTraceDepth is None else maxTraceDepth
        self.params.me = sc.me if rmdde is None else rmdde
        self.params.mr = sc.mr if rmdsr is None else rmdsr
        self.params.seed = seed
        self.params.sample_begin, self.params.sample_count = 0, psamples
        self.stats = abi.Stats()
        self._prog = C.c_void_p()  # created by the first call of the progressive interface

    def render(self, buffer=None, profile=False):
        """Fills ``buffer`` (float32 [H,W,3], created when None) with mean radiance and returns it.

        Once one of reset / setIterationCount / setRenderResolution / setRegion has been called, render() is one
        iteration of the progressive interface, as in the reference (cpp:141-161): the buffer receives the
        running average over the iterations so far."""
        W, H = self.content.scene.resolution
        if buffer is None:
            buffer = np.empty((H, W, 3), np.float32)
        if buffer.dtype != np.float32 or buffer.size != W * H * 3 or not buffer.flags.c_contiguous:
            raise ValueError("frame buffer must be contiguous float32 with W*H*3 elements")
        self.params.profile = int(profile)
        if self._prog:
            if profile:
                # (the progressive handle copied the parameters when it was created)
                raise ValueError("profile is not available through the progressive interface")
            capi.check(capi.wapt_progressive_render(self._prog, buffer.ctypes.data, C.byref(self.stats)))
            return buffer
        self.params.sample_begin, self.params.sample_count = 0, self.params.psamples
        capi.check(capi.wapt_render(self.content._h, C.byref(self.params), buffer.ctypes.data, C.byref(self.stats)))
        return buffer

    def render_accumulate(self, d_sum_ptr, sample_begin, sample_count, profile=False):
        """Adds the radiance SUM of samples [sample_begin, sample_begin+sample_count) into a W*H*3 float buffer in
        device memory (spp sharding across GPUs: the caller reduces the buffers and divides by psamples)."""
        self.params.profile = int(profile)
        self.params.sample_begin, self.params.sample_count = sample_begin, sample_count
        capi.check(capi.wapt_render_accumulate(self.content._h, C.byref(self.params), d_sum_ptr, C.byref(self.stats)))
        return self.stats


    # ---- progressive interface (src/unidirectionalPathTracer.cpp:68-76, 907-972) ----------------------------

    def _progressive(self):
        if not self._prog:
            capi.check(capi.wapt_progressive_create(self.content._h, C.byref(self.params), C.byref(self._prog)))
        return self._prog

    def reset(self):
        capi.check(capi.wapt_progressive_reset(self._progressive()))

    def setIterationCount(self, itCnt):
        capi.check(capi.wapt_progressive_set_iteration(self._progressive(), itCnt))

    def setRenderResolution(self, downresFactorOf2):
        capi.check(capi.wapt_progressive_set_resolution(self._progressive(), downresFactorOf2))

    def setRegion(self, l, r, b, t):
        capi.check(capi.wapt_progressive_set_region(self._progressive(), l, r, b, t))

    def abort(self):
        capi.check(capi.wapt_progressive_abort(self._progressive()))

    def generateTiles(self, subdivLevel, renW=None, renH=None):
        """Tiles of udp_renderer::generateTiles as a list of (l, r, b, t)."""
        W, H = self.content.scene.resolution
        tiles = (abi.Tile * (4 ** subdivLevel))()
        n = C.c_int32()
        capi.check(capi.wapt_generate_tiles(renW or W, renH or H, subdivLevel, tiles, C.byref(n)))
        return [(t.l, t.r, t.b, t.t) for t in tiles[:n.value]]

    def close(self):
        if self._prog:
            capi.wapt_progressive_destroy(self._prog)
            self._prog = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tonemap(fb):
    """Tone map in place: clamp(pow(x, 0.45), 0, 1) (reference src/myRenderer.cpp:174-185)."""
    H, W = fb.shape[:2]
    capi.check(capi.wapt_tonemap(fb.ctypes.data, W, H))
    return fb


def write_bmp(path, fb):
    """writeImage (reference src/myRenderer.cpp:686-726); fb must already be tone mapped."""
    import os
    H, W = fb.shape[:2]
    fb = np.ascontiguousarray(fb, np.float32)
    capi.check(capi.wapt_write_bmp(os.fsencode(path), W, H, fb.ctypes.data))
