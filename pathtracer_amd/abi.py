"""ctypes mirrors of the plain-C structs declared in include/wapt.h (no library is loaded here)."""
import ctypes as C

f3 = C.c_float * 3

MAT_OPAQUE, MAT_CLEAR, MAT_LIGHT, MAT_DIFFUSE = 1, 2, 4, 8
MAT_SPECULAR, MAT_SSS, MAT_DIELECTRIC, MAT_DIELECTRICSSS = 16, 32, 64, 128


class Camera(C.Structure):
    _fields_ = [("type", C.c_int32), ("P", f3), ("D", f3), ("up", f3), ("fov_deg", C.c_float),
                ("resI", C.c_int32), ("resJ", C.c_int32), ("sensor_width", C.c_float)]


class Material(C.Structure):
    _fields_ = [("kd", f3), ("ks", f3), ("kt", f3), ("sa", f3), ("ss", f3), ("st", f3),
                ("ior", C.c_float), ("flags", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("P", f3), ("U", f3), ("V", f3), ("radius", C.c_float), ("power", f3)]


class SceneDesc(C.Structure):
    _fields_ = [("camera", Camera), ("global_medium", Material),
                ("n_lights", C.c_int32), ("lights", C.POINTER(Light)),
                ("n_objects", C.c_int32), ("materials", C.POINTER(Material)),
                ("n_tris", C.c_uint64), ("tri_verts", C.POINTER(C.c_float)),
                ("tri_obj", C.POINTER(C.c_int32)), ("tri_prim", C.POINTER(C.c_int32)),
                ("world_center", f3), ("world_radius", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("psamples", C.c_int32), ("hsamples", C.c_int32), ("lsamples", C.c_int32), ("max_depth", C.c_int32),
                ("me", C.c_float), ("mr", C.c_float), ("seed", C.c_uint64),
                ("sample_begin", C.c_int32), ("sample_count", C.c_int32), ("profile", C.c_int32),
                ("reserved", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("stack_overflows", C.c_uint64), ("seconds", C.c_double),
                ("ms_raygen", C.c_double), ("ms_trace", C.c_double), ("ms_head", C.c_double), ("ms_shade", C.c_double),
                ("ms_volume", C.c_double), ("ms_shadow", C.c_double), ("ms_probe", C.c_double),
                ("n_trace_launches", C.c_uint64), ("n_trace_rays", C.c_uint64),
                ("n_shadow_launches", C.c_uint64), ("n_shadow_rays", C.c_uint64),
                ("trace_node_visits", C.c_uint64), ("trace_tri_tests", C.c_uint64),
                ("trace_wave_steps", C.c_uint64), ("trace_node_steps", C.c_uint64), ("trace_tri_trips", C.c_uint64),
                ("shadow_node_visits", C.c_uint64), ("shadow_tri_tests", C.c_uint64), ("shadow_wave_steps", C.c_uint64),
                ("shadow_node_steps", C.c_uint64), ("shadow_tri_trips", C.c_uint64),
                ("queries_decided", C.c_uint64), ("n_probe_rays", C.c_uint64), ("probe_node_visits", C.c_uint64), ("probe_tri_tests", C.c_uint64),
                ("probe_wave_steps", C.c_uint64), ("probe_node_steps", C.c_uint64), ("probe_tri_trips", C.c_uint64),
                ("queue_bytes", C.c_uint64), ("queue_tasks", C.c_uint64)]


class Ray(C.Structure):
    _fields_ = [("org", f3), ("tnear", C.c_float), ("dir", f3), ("tfar", C.c_float)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("prim", C.c_int32), ("obj", C.c_int32), ("Ng", f3)]


class BvhInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_tris", C.c_uint64), ("node_bytes", C.c_uint64), ("tri_bytes", C.c_uint64),
                ("node_size", C.c_uint32), ("tri_size", C.c_uint32), ("max_depth", C.c_uint32), ("sah_cost", C.c_float),
                ("builder", C.c_uint32), ("build_seconds", C.c_float)]


class TraceCounters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("stack_spills", C.c_uint64),
                ("wave_steps", C.c_uint64), ("node_steps", C.c_uint64), ("tri_trips", C.c_uint64)]


class Tile(C.Structure):  # wapt_tile
    _fields_ = [("l", C.c_int32), ("r", C.c_int32), ("b", C.c_int32), ("t", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]

BUILDER_AUTO, BUILDER_HOST_SAH, BUILDER_DEVICE_LBVH = 0, 1, 2
