"""Host code under sanitizers: the scene-file and mesh readers, the host BVH builder, tone map and BMP writer of the
library and the CPU checker are compiled from their sources with AddressSanitizer + UndefinedBehaviorSanitizer (and the
checker's multi-threaded render with ThreadSanitizer) and driven by tests/sanitize/host_san.cpp, including truncated
and corrupted input files (the reference reads such files unchecked or loops forever, src/myRenderer.cpp:355-363, 651;
here every outcome but a status code is a bug).  No GPU code is involved."""
import os
import shutil
import subprocess

import pytest

from pathtracer_amd import meshes, scenes
from test_host_io import WAFF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = [os.path.join(ROOT, "tests", "sanitize", "host_san.cpp"),
           os.path.join(ROOT, "pathtracer_amd", "csrc", "scene_io.cpp"),
           os.path.join(ROOT, "pathtracer_amd", "csrc", "bvh_build.cpp"),
           os.path.join(ROOT, "oracle", "oracle.cpp")]


def build(tmp, name, flags):
    exe = str(tmp / name)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", *flags, *SOURCES, "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("cannot find" in r.stderr or "unrecognized" in r.stderr):
        pytest.skip("sanitizer runtime not installed: " + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


@pytest.fixture(scope="module")
def scene_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    for f in ("cornellBox.obj", "cornellBoxLeftWall.obj", "cornellBoxRightWall.obj"):
        shutil.copy(os.path.join(scenes.ASSET_DIR, f), d / f)
    shutil.copy(os.path.join(scenes.ASSET_DIR, "cornellBox.obj"), d / "box.obj")
    v, f = meshes.uv_sphere((100, 100, 100), 50, 8, 12)
    meshes.write_fast(d / "ball.fast", v, f)
    (d / "test.waff").write_text(WAFF)
    return d


def run(exe, mode, scene_dir, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([exe, mode, str(scene_dir)], capture_output=True, text=True, cwd=str(scene_dir), env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "runtime error" not in r.stderr and "Sanitizer" not in r.stderr, r.stderr[-3000:]
    return r.stdout


@pytest.fixture(scope="module")
def asan_exe(tmp_path_factory):
    return build(tmp_path_factory.mktemp("asan"), "host_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"])


ASAN_ENV = {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"}


def test_host_pipeline_under_asan_ubsan(asan_exe, scene_dir):
    run(asan_exe, "host", scene_dir, ASAN_ENV)


def test_damaged_scene_and_mesh_files_only_ever_yield_status_codes(asan_exe, scene_dir):
    out = run(asan_exe, "fuzz", scene_dir, ASAN_ENV)
    assert "accepted" in out


def test_checker_render_under_asan_ubsan(asan_exe, scene_dir):
    assert "rays" in run(asan_exe, "oracle", scene_dir, ASAN_ENV)


def test_checker_threads_under_tsan(tmp_path_factory, scene_dir):
    exe = build(tmp_path_factory.mktemp("tsan"), "host_tsan", ["-fsanitize=thread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, "threads", str(scene_dir)], capture_output=True, text=True, cwd=str(scene_dir), env=env, timeout=600)
    if r.returncode != 0 and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory on this kernel")
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "data race" not in r.stderr, r.stderr[-3000:]
