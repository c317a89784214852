"""The C ABI: libwapt.so loads without a GPU, exports every function include/wapt.h declares, and the ctypes
mirrors have the sizes the C compiler gives the structs.  No compute entry point is called here."""
import ctypes as C
import os
import re
import subprocess

import pytest

from pathtracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wapt.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wapt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_function():
    from pathtracer_amd import capi
    names = declared_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(capi.lib, n), f"{n} is declared in wapt.h but not exported"
    assert sorted(capi.EXPORTS) == names
    assert b"gfx950" in capi.wapt_version()


def test_struct_sizes_match_the_c_compiler(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "wapt.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(wapt_camera),sizeof(wapt_material),sizeof(wapt_light),sizeof(wapt_scene_desc),"
                   "sizeof(wapt_render_params),sizeof(wapt_stats),sizeof(wapt_ray),sizeof(wapt_hit),"
                   "sizeof(wapt_bvh_info),sizeof(wapt_trace_counters));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])  # plain C: the header is C-clean
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    mirrors = [abi.Camera, abi.Material, abi.Light, abi.SceneDesc, abi.RenderParams, abi.Stats, abi.Ray, abi.Hit,
               abi.BvhInfo, abi.TraceCounters]
    assert sizes == [C.sizeof(m) for m in mirrors]
    assert C.sizeof(abi.Ray) == 32 and C.sizeof(abi.Hit) == 24


def test_errors_without_a_gpu_are_codes_not_crashes():
    from pathtracer_amd import capi
    h = C.c_void_p()
    assert capi.wapt_waff_load(b"/nonexistent/scene.waff", C.byref(h)) == 2  # WAPT_E_IO
    assert b"cannot open" in capi.wapt_last_error()
    assert capi.wapt_scene_create(None, 0, C.byref(h)) == 1  # WAPT_E_INVALID
    assert capi.wapt_render(None, None, None, None) == 1
    assert capi.wapt_intersect(None, None, 5, None) == 1


def test_missing_library_fails_loudly(tmp_path):
    code = ("import os,sys; os.environ['WAPT_LIB']='does_not_exist.so'; sys.path.insert(0, %r)\n"
            "try:\n    import pathtracer_amd.capi\nexcept ImportError as e:\n    print('IMPORTERROR', e)\n" % ROOT)
    out = subprocess.check_output(["python", "-c", code], text=True)
    assert "IMPORTERROR" in out and "no CPU path" in out
