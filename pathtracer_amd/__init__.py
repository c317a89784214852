"""MI355X-native radiance loop for willalexander/pathtracer scenes (host mirror over libwapt.so)."""
from . import abi  # noqa: F401
from .scene import Scene, area_light, camera, material, sphere_light  # noqa: F401


def __getattr__(name):
    # the classes that need the HIP library are imported on first use so that tools which only
    # build scene descriptions (mesh generators, the CPU checker's tests) do not need it
    if name in ("wa_content", "udp_renderer", "tonemap", "write_bmp"):
        from . import renderer
        return getattr(renderer, name)
    if name == "capi":
        import importlib
        return importlib.import_module(".capi", __name__)
    raise AttributeError(name)
