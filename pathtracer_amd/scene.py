"""Flattened scenes: what wa_content holds after init() (reference src/content.cpp:13-40).

A Scene keeps numpy copies of the triangle soup plus ctypes tables of materials and lights, so the
same description can be handed to libwapt.so and to the CPU checker used by the tests.
"""
import ctypes as C
import os

import numpy as np

from . import abi


def material(kind="diffuse", colour=(0.5, 0.5, 0.5), ior=1.0, kt=(0, 0, 0), sa=(0, 0, 0), ss=(0, 0, 0)):
    """Material table rows as the .waff parser produces them (reference src/myRenderer.cpp:525-639)."""
    m = abi.Material()
    z = (0.0, 0.0, 0.0)
    m.kd, m.ks, m.kt = abi.f3(*z), abi.f3(*z), abi.f3(*kt)
    m.sa, m.ss = abi.f3(*sa), abi.f3(*ss)
    m.st = abi.f3(*[a + s for a, s in zip(sa, ss)])
    m.ior = ior
    if kind == "diffuse":
        m.kd = abi.f3(*colour)
        m.flags = abi.MAT_OPAQUE | abi.MAT_CLEAR | abi.MAT_DIFFUSE
    elif kind == "specular":
        m.ks = abi.f3(*colour)
        m.flags = abi.MAT_OPAQUE | abi.MAT_CLEAR | abi.MAT_SPECULAR
    elif kind == "dielectric":
        m.ks, m.kt = abi.f3(1, 1, 1), abi.f3(1, 1, 1)
        m.flags = abi.MAT_CLEAR | abi.MAT_DIELECTRIC
    elif kind == "dielectricsss":
        m.ks = abi.f3(1, 1, 1)
        m.flags = abi.MAT_DIELECTRICSSS
    elif kind == "sss":
        m.ks = abi.f3(0.5, 0.5, 0.5)
        m.flags = 0
    elif kind == "light":  # placeholder object of a light: a 0.5 Lambert reflector that also emits
        m.kd, m.ks = abi.f3(0.5, 0.5, 0.5), abi.f3(0.5, 0.5, 0.5)
        m.flags = abi.MAT_OPAQUE | abi.MAT_CLEAR | abi.MAT_DIFFUSE | abi.MAT_LIGHT
    elif kind == "clear_medium":
        m.kd, m.ks = abi.f3(0.5, 0.5, 0.5), abi.f3(0.5, 0.5, 0.5)
        m.flags = abi.MAT_OPAQUE | abi.MAT_CLEAR | abi.MAT_DIFFUSE
    elif kind == "medium":  # global participating medium ("m 1 ss sa")
        m.kd, m.ks = abi.f3(0.5, 0.5, 0.5), abi.f3(0.5, 0.5, 0.5)
        m.flags = abi.MAT_OPAQUE | abi.MAT_DIFFUSE
    else:
        raise ValueError(kind)
    return m


def area_light(P, U, V, power):
    l = abi.Light()
    l.type, l.P, l.U, l.V, l.radius, l.power = 0, abi.f3(*P), abi.f3(*U), abi.f3(*V), 0.0, abi.f3(*power)
    return l


def sphere_light(P, radius, power):
    l = abi.Light()
    l.type, l.P, l.U, l.V, l.radius, l.power = 1, abi.f3(*P), abi.f3(0, 0, 0), abi.f3(0, 0, 0), radius, abi.f3(*power)
    return l


def camera(P, at, up=(0, 1, 0), fov=40.0, res=(600, 600), width=1.0, type=0):
    c = abi.Camera()
    P = np.asarray(P, np.float32)
    d = np.asarray(at, np.float32) - P
    d = d / np.float32(np.sqrt(np.float32((d * d).sum())))
    c.type, c.P, c.D, c.up = type, abi.f3(*P), abi.f3(*d), abi.f3(*up)
    c.fov_deg, c.resI, c.resJ, c.sensor_width = fov, res[0], res[1], width
    return c


class Scene:
    def __init__(self, cam, lights, mesh_materials, meshes, global_medium=None,
                 world_center=(275.0, 275.0, 275.0), world_radius=1100.0, td=3, me=0.0, mr=0.01):
        """meshes: list of float32 arrays [n,3,3] (one per mesh object, CCW front)."""
        self.camera = cam
        self.lights = list(lights)
        self.global_medium = global_medium if global_medium is not None else material("clear_medium")
        self.materials = [material("light") for _ in self.lights] + list(mesh_materials)
        self.td, self.me, self.mr = td, me, mr
        self.world_center, self.world_radius = tuple(world_center), world_radius
        verts, obj, prim = [], [], []
        for li, l in enumerate(self.lights):
            if l.type != 0:
                continue  # sphere lights own an object slot but no geometry
            P, U, V = (np.array(list(x), np.float32) for x in (l.P, l.U, l.V))
            v0, v1, v2, v3 = P, P + U, P + U + V, P + V
            verts.append(np.stack([np.stack([v0, v1, v3]), np.stack([v2, v3, v1])]))
            obj += [li, li]
            prim += [0, 0]
        for gi, m in enumerate(meshes):
            m = np.ascontiguousarray(m, np.float32).reshape(-1, 3, 3)
            verts.append(m)
            obj.append(np.full(len(m), len(self.lights) + gi, np.int32))
            prim.append(np.arange(len(m), dtype=np.int32))
        self.tri_verts = np.ascontiguousarray(np.concatenate(verts, 0), np.float32)
        self.tri_obj = np.ascontiguousarray(np.concatenate([np.atleast_1d(np.asarray(o, np.int32)) for o in obj]))
        self.tri_prim = np.ascontiguousarray(np.concatenate([np.atleast_1d(np.asarray(p, np.int32)) for p in prim]))

    @property
    def resolution(self):
        return self.camera.resI, self.camera.resJ

    def desc(self):
        """SceneDesc pointing into this object's arrays (keep the Scene alive while it is used)."""
        d = abi.SceneDesc()
        d.camera = self.camera
        d.global_medium = self.global_medium
        self._lights_arr = (abi.Light * max(len(self.lights), 1))(*self.lights)
        self._mats_arr = (abi.Material * len(self.materials))(*self.materials)
        d.n_lights, d.lights = len(self.lights), C.cast(self._lights_arr, C.POINTER(abi.Light))
        d.n_objects, d.materials = len(self.materials), C.cast(self._mats_arr, C.POINTER(abi.Material))
        d.n_tris = len(self.tri_obj)
        d.tri_verts = self.tri_verts.ctypes.data_as(C.POINTER(C.c_float))
        d.tri_obj = self.tri_obj.ctypes.data_as(C.POINTER(C.c_int32))
        d.tri_prim = self.tri_prim.ctypes.data_as(C.POINTER(C.c_int32))
        d.world_center = abi.f3(*self.world_center)
        d.world_radius = self.world_radius
        return d

    @classmethod
    def from_desc(cls, d, td, me, mr):
        """Deep copy of a SceneDesc produced by the C loader."""
        s = cls.__new__(cls)
        s.camera = abi.Camera.from_buffer_copy(d.camera)
        s.global_medium = abi.Material.from_buffer_copy(d.global_medium)
        s.lights = [abi.Light.from_buffer_copy(d.lights[i]) for i in range(d.n_lights)]
        s.materials = [abi.Material.from_buffer_copy(d.materials[i]) for i in range(d.n_objects)]
        n = int(d.n_tris)
        s.tri_verts = np.ctypeslib.as_array(d.tri_verts, shape=(n, 3, 3)).copy()
        s.tri_obj = np.ctypeslib.as_array(d.tri_obj, shape=(n,)).copy()
        s.tri_prim = np.ctypeslib.as_array(d.tri_prim, shape=(n,)).copy()
        s.world_center, s.world_radius = tuple(d.world_center), d.world_radius
        s.td, s.me, s.mr = td, me, mr
        return s

    @classmethod
    def from_waff(cls, path):
        """loadContentFromFile_waff through the library's C++ loader (reference src/myRenderer.cpp:649-683)."""
        from . import capi
        h = C.c_void_p()
        capi.check(capi.wapt_waff_load(os.fsencode(path), C.byref(h)))
        try:
            d = abi.SceneDesc()
            capi.check(capi.wapt_waff_desc(h, C.byref(d)))
            td, me, mr = C.c_int32(), C.c_float(), C.c_float()
            capi.check(capi.wapt_waff_settings(h, C.byref(td), C.byref(me), C.byref(mr)))
            return cls.from_desc(d, td.value, me.value, mr.value)
        finally:
            capi.wapt_waff_free(h)
