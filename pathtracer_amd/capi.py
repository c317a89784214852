"""Loader and prototypes of libwapt.so, the C ABI declared in include/wapt.h.

There is no fallback: if the shared library has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C pathtracer_amd/csrc``)
importing this module raises.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("WAPT_LIB", "libwapt.so"))  # WAPT_LIB: alternative build, e.g. a debug one

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP library first (make -C pathtracer_amd/csrc); "
        "pathtracer_amd has no CPU path")

# PyTorch wheels carry their own copy of the HIP runtime.  Two copies in one process cannot both own the
# GPU ("No HIP GPUs are available" from whichever initialises second), so when torch is installed it is
# imported first: its runtime has the same soname as the one libwapt.so asks for and is then shared.
try:
    import torch  # noqa: F401
except ImportError:
    pass

lib = C.CDLL(LIB_PATH)


class WaptError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"wapt error {code}: {msg}")
        self.code = code


def _proto(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


P = C.POINTER
vp = C.c_void_p

wapt_last_error = _proto("wapt_last_error", C.c_char_p)
wapt_version = _proto("wapt_version", C.c_char_p)
wapt_waff_load = _proto("wapt_waff_load", C.c_int, C.c_char_p, P(vp))
wapt_waff_desc = _proto("wapt_waff_desc", C.c_int, vp, P(abi.SceneDesc))
wapt_waff_settings = _proto("wapt_waff_settings", C.c_int, vp, P(C.c_int32), P(C.c_float), P(C.c_float))
wapt_waff_free = _proto("wapt_waff_free", None, vp)
wapt_convert_obj_to_fast = _proto("wapt_convert_obj_to_fast", C.c_int, C.c_char_p, C.c_char_p)
wapt_tonemap = _proto("wapt_tonemap", C.c_int, vp, C.c_int32, C.c_int32)
wapt_write_bmp = _proto("wapt_write_bmp", C.c_int, C.c_char_p, C.c_int32, C.c_int32, vp)
wapt_scene_create = _proto("wapt_scene_create", C.c_int, P(abi.SceneDesc), C.c_int32, P(vp))
wapt_scene_create_with = _proto("wapt_scene_create_with", C.c_int, P(abi.SceneDesc), C.c_int32, C.c_int32, P(vp))
wapt_scene_destroy = _proto("wapt_scene_destroy", None, vp)
wapt_scene_bvh_info = _proto("wapt_scene_bvh_info", C.c_int, vp, P(abi.BvhInfo))
wapt_render = _proto("wapt_render", C.c_int, vp, P(abi.RenderParams), vp, P(abi.Stats))
wapt_render_accumulate = _proto("wapt_render_accumulate", C.c_int, vp, P(abi.RenderParams), vp, P(abi.Stats))
wapt_intersect = _proto("wapt_intersect", C.c_int, vp, vp, C.c_uint32, vp)
wapt_intersect_device = _proto("wapt_intersect_device", C.c_int, vp, vp, C.c_uint32, vp, P(abi.TraceCounters),
                               P(C.c_float))
wapt_intersect_device_static = _proto("wapt_intersect_device_static", C.c_int, vp, vp, C.c_uint32, vp)
wapt_progressive_create = _proto("wapt_progressive_create", C.c_int, vp, P(abi.RenderParams), P(vp))
wapt_progressive_destroy = _proto("wapt_progressive_destroy", None, vp)
wapt_progressive_reset = _proto("wapt_progressive_reset", C.c_int, vp)
wapt_progressive_set_iteration = _proto("wapt_progressive_set_iteration", C.c_int, vp, C.c_int32)
wapt_progressive_set_resolution = _proto("wapt_progressive_set_resolution", C.c_int, vp, C.c_int32)
wapt_progressive_set_region = _proto("wapt_progressive_set_region", C.c_int, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32)
wapt_progressive_render = _proto("wapt_progressive_render", C.c_int, vp, vp, P(abi.Stats))
wapt_progressive_abort = _proto("wapt_progressive_abort", C.c_int, vp)
wapt_generate_tiles = _proto("wapt_generate_tiles", C.c_int, C.c_int32, C.c_int32, C.c_int32, P(abi.Tile), P(C.c_int32))

EXPORTS = [
    "wapt_last_error", "wapt_version", "wapt_waff_load", "wapt_waff_desc", "wapt_waff_settings", "wapt_waff_free",
    "wapt_convert_obj_to_fast", "wapt_tonemap", "wapt_write_bmp", "wapt_scene_create", "wapt_scene_create_with",
    "wapt_scene_destroy",
    "wapt_scene_bvh_info", "wapt_render", "wapt_render_accumulate", "wapt_intersect", "wapt_intersect_device", "wapt_intersect_device_static",
    "wapt_progressive_create", "wapt_progressive_destroy", "wapt_progressive_reset", "wapt_progressive_set_iteration",
    "wapt_progressive_set_resolution", "wapt_progressive_set_region", "wapt_progressive_render",
    "wapt_progressive_abort", "wapt_generate_tiles",
]


def check(rc):
    if rc != 0:
        raise WaptError(rc, wapt_last_error().decode(errors="replace"))
