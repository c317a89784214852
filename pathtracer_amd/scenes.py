"""The benchmark/test scenes, built from the three bundled Cornell OBJ files (scenes/*.obj, copied
data of the reference's waff/ directory) and the procedural stand-ins of meshes.py.

Camera, render settings and the area light are those of the reference's only scene file
(waff/dragonChrome.waff:1-14); everything the reference does not ship (sphere placement, glass/milk/fog
coefficients) is fixed here and used identically by the GPU library and the CPU checker.
"""
import os

import numpy as np

from . import meshes
from .scene import Scene, area_light, camera, material, sphere_light

ASSET_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")

LIGHT = dict(P=(200, 550, 200), U=(150, 0, 0), V=(0, 0, 150), power=(4000000, 4000000, 4000000))
BOX_FILES = [("cornellBox.obj", (0.7, 0.7, 0.7)), ("cornellBoxLeftWall.obj", (0.7, 0.07, 0.07)),
             ("cornellBoxRightWall.obj", (0.07, 0.7, 0.07))]
SPHERE_A = dict(centre=(369.0, 120.0, 350.0), radius=120.0)   # specular, ks 1
SPHERE_B = dict(centre=(197.0, 106.0, 166.0), radius=106.0)   # diffuse 0.7
GLASS_IOR = 1.5
MILK = dict(ior=1.3, kt=(1.0, 1.0, 1.0), sa=(0.0005, 0.001, 0.002), ss=(0.03, 0.03, 0.03))
FOG = dict(ss=(0.001, 0.001, 0.001), sa=(0.0002, 0.0002, 0.0002))


def _cam(res):
    return camera(P=(275, 275, -755), at=(275, 275, 0), up=(0, 1, 0), fov=40.0, res=res, width=1.0)


def _box():
    mats, ms = [], []
    for name, kd in BOX_FILES:
        v, f = meshes.read_obj(os.path.join(ASSET_DIR, name))
        ms.append(meshes.expand(v, f))
        mats.append(material("diffuse", kd))
    return mats, ms


def cornell(res=(512, 512), spheres=True, sphere_segments=(64, 128), td=3):
    """BASELINE config 0: Cornell box, Lambert walls, area light, mirror sphere A and diffuse sphere B."""
    mats, ms = _box()
    if spheres:
        va, fa = meshes.uv_sphere(SPHERE_A["centre"], SPHERE_A["radius"], *sphere_segments)
        vb, fb = meshes.uv_sphere(SPHERE_B["centre"], SPHERE_B["radius"], *sphere_segments)
        ms += [meshes.expand(va, fa), meshes.expand(vb, fb)]
        mats += [material("specular", (1.0, 1.0, 1.0)), material("diffuse", (0.7, 0.7, 0.7))]
    return Scene(_cam(res), [area_light(**LIGHT)], mats, ms, td=td, me=0.0, mr=0.01)


def dragon(kind="chrome", res=(1920, 1080), n_along=1320, n_around=330, td=3, mesh=None):
    """BASELINE configs 1 and 2: the box with the dragon stand-in as chrome / glass / milk."""
    mats, ms = _box()
    if mesh is None:
        v, f = meshes.dragon_standin(n_along, n_around)
        mesh = meshes.expand(v, f)
    ms.append(mesh)
    if kind == "chrome":
        mats.append(material("specular", (1.0, 1.0, 1.0)))
    elif kind == "glass":
        mats.append(material("dielectric", ior=GLASS_IOR))
    elif kind == "milk":
        mats.append(material("dielectricsss", **MILK))
    else:
        raise ValueError(kind)
    return Scene(_cam(res), [area_light(**LIGHT)], mats, ms, td=td, me=0.0, mr=0.01)


def fog(res=(1920, 1080), td=3, lights="sphere", spheres=True):
    """BASELINE config 3: homogeneous isotropic global medium; sphere lights (or the area light)."""
    mats, ms = _box()
    if spheres:
        vb, fb = meshes.uv_sphere(SPHERE_B["centre"], SPHERE_B["radius"], 32, 64)
        ms.append(meshes.expand(vb, fb))
        mats.append(material("diffuse", (0.7, 0.7, 0.7)))
    if lights == "sphere":
        ls = [sphere_light((275, 450, 275), 40.0, (3000000, 3000000, 3000000)),
              sphere_light((120, 300, 420), 20.0, (800000, 500000, 200000))]
    else:
        ls = [area_light(**LIGHT)]
    return Scene(_cam(res), ls, mats, ms, global_medium=material("medium", **FOG), td=td, me=0.0, mr=0.01)


def stress(res=(3840, 2160), copies=64, n_along=1320, n_around=330, td=3, materials="mixed"):
    """BASELINE config 4: `copies` transformed dragons flattened into one soup, materials cycling
    (materials="opaque": diffuse and mirror only, the traversal-bound variant used for the HBM traffic profile)."""
    mats, ms = _box()
    v, f = meshes.dragon_standin(n_along, n_around)
    centre = np.array([307.0, 0.0, 274.0])
    side = int(np.ceil(np.sqrt(copies)))
    kinds = [material("diffuse", (0.6, 0.6, 0.7)), material("specular", (1.0, 1.0, 1.0)),
             material("dielectric", ior=GLASS_IOR), material("dielectricsss", **MILK)]
    if materials == "opaque":
        kinds = kinds[:2]
    for k in range(copies):
        a = 2.0 * np.pi * (k * 0.61803398875 % 1.0)
        rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        s = 0.9 / side
        off = np.array([(k % side + 0.5) * 550.0 / side, 0.0, (k // side + 0.5) * 550.0 / side])
        vv = ((v.astype(np.float64) - centre) * s) @ rot.T + off
        ms.append(meshes.expand(vv.astype(np.float32), f))
        mats.append(kinds[k % len(kinds)])
    return Scene(_cam(res), [area_light(**LIGHT)], mats, ms, td=td, me=0.0, mr=0.01)
