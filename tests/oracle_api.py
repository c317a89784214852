"""ctypes access to oracle/liboracle.so — the CPU checker.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from pathtracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ROOT, "oracle", "oracle.cpp")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(build())
        l.oracle_scene_create.restype = C.c_void_p
        l.oracle_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
        l.oracle_scene_destroy.argtypes = [C.c_void_p]
        l.oracle_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int]
        l.oracle_render.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        l.oracle_last_shadow_rays.restype = C.c_uint64
        l.oracle_rand.restype = C.c_float
        l.oracle_rand.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        l.oracle_key_child.restype = C.c_uint32
        l.oracle_key_child.argtypes = [C.c_uint32, C.c_uint32]
        l.oracle_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.POINTER(C.c_float)]
        l.oracle_camera_ray.argtypes = [C.POINTER(abi.Camera), C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        l.oracle_light_sample.restype = C.c_int
        l.oracle_light_sample.argtypes = [C.POINTER(abi.Light), C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                          C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        l.oracle_light_pdf.restype = C.c_float
        l.oracle_light_pdf.argtypes = [C.POINTER(abi.Light), C.c_void_p, C.c_void_p]
        l.oracle_sample_dirs.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.POINTER(C.c_float)]
        _lib = l
    return _lib


HIT_DTYPE = np.dtype([("t", np.float32), ("prim", np.int32), ("obj", np.int32), ("Ng", np.float32, 3)])


class OracleScene:
    def __init__(self, scene):
        self.scene = scene
        self._d = scene.desc()
        self._h = lib().oracle_scene_create(C.byref(self._d))
        assert self._h

    def intersect(self, rays, brute=False, threads=8):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(len(rays), HIT_DTYPE)
        rc = lib().oracle_intersect(self._h, rays.ctypes.data, len(rays), hits.ctypes.data, 0 if brute else 1, threads)
        assert rc == 0
        return hits

    def render(self, psamples, hsamples=1, lsamples=1, td=None, me=None, mr=None, seed=0, threads=8,
               sample_begin=0, sample_count=None, sequential=False):
        """sequential: every random number from a std::mt19937_64 per rendering thread, drawn in call order, instead
        of the keyed hash that the GPU library shares (the image then depends on the thread count)."""
        sc = self.scene
        p = abi.RenderParams()
        p.psamples, p.hsamples, p.lsamples = psamples, hsamples, lsamples
        p.max_depth = sc.td if td is None else td
        p.me = sc.me if me is None else me
        p.mr = sc.mr if mr is None else mr
        p.seed = seed
        p.sample_begin = sample_begin
        p.sample_count = psamples if sample_count is None else sample_count
        p.profile = 4 if sequential else 0
        W, H = sc.resolution
        fb = np.zeros((H, W, 3), np.float32)
        rays = C.c_uint64()
        rc = lib().oracle_render(self._h, C.byref(p), fb.ctypes.data, threads, C.byref(rays))
        assert rc == 0
        self.rays = rays.value
        self.shadow_rays = lib().oracle_last_shadow_rays()
        return fb

    def close(self):
        if self._h:
            lib().oracle_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
