"""Deterministic procedural stand-ins for the meshes the reference repository does not ship
(sphereA/B.obj and the Stanford dragon, see its .MISSING_LARGE_BLOBS), plus OBJ / .fast file I/O.

All generators return (verts float32 [nv,3], faces int32 [nt,3]) with counter-clockwise = outward.
The .fast layout is the reference's (src/myRenderer.cpp:102-114): int32 nV, int32 nT,
nV x (float x,y,z,pad), nT x (int32 v0,v1,v2).
"""
import numpy as np


def expand(verts, faces):
    """Indexed mesh -> triangle soup float32 [nt,3,3]."""
    return np.ascontiguousarray(verts[faces], np.float32)


def uv_sphere(centre, radius, n_lat=64, n_lon=128):
    lat = np.linspace(0.0, np.pi, n_lat + 1)
    lon = np.linspace(0.0, 2.0 * np.pi, n_lon, endpoint=False)
    th, ph = np.meshgrid(lat, lon, indexing="ij")
    d = np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], -1)
    verts = (np.asarray(centre, np.float64) + radius * d).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    a = (i * n_lon + j).ravel()
    b = (i * n_lon + (j + 1) % n_lon).ravel()
    c = ((i + 1) * n_lon + j).ravel()
    d2 = ((i + 1) * n_lon + (j + 1) % n_lon).ravel()
    faces = np.concatenate([np.stack([a, b, c], -1), np.stack([b, d2, c], -1)]).astype(np.int32)
    faces = _orient_outward(verts, faces, np.asarray(centre, np.float64))
    return verts, _drop_degenerate(verts, faces)


def _drop_degenerate(verts, faces):
    t = verts[faces].astype(np.float64)
    n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    keep = (n * n).sum(-1) > 0
    return np.ascontiguousarray(faces[keep])


def _orient_outward(verts, faces, centre):
    t = verts[faces].astype(np.float64)
    n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    out = ((t.mean(1) - centre) * n).sum(-1) < 0
    faces = faces.copy()
    faces[out] = faces[out][:, ::-1]
    return faces


def dragon_standin(n_along=1320, n_around=330, seed=20261111):
    """Closed, bumpy, S-shaped tube of 2*n_along*n_around triangles (871 200 by default) that sits where
    the dragon sits in the reference's dragonChrome render: roughly x 150..480, y 0..230, z 150..400."""
    rng = np.random.default_rng(seed)
    u = np.linspace(0.0, 1.0, n_along + 1)
    v = np.linspace(0.0, 2.0 * np.pi, n_around, endpoint=False)
    ang = 2.2 * np.pi * u - 0.3
    c = np.stack([315.0 + 135.0 * np.cos(ang) * (1.0 - 0.3 * u),
                  95.0 + 55.0 * u + 35.0 * np.sin(3.0 * np.pi * u),
                  275.0 + 95.0 * np.sin(ang)], -1)
    t = np.gradient(c, u, axis=0)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    n1 = np.cross(t, up)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(n1, t)
    r0 = 4.0 + 46.0 * np.sin(np.pi * u) ** 0.6
    r0[0] = r0[-1] = 0.0  # closed at both tips
    # low-frequency lumps from a few random harmonics + a regular "scale" pattern
    k = rng.integers(1, 9, size=(6, 2))
    ph = rng.uniform(0, 2 * np.pi, size=(6, 2))
    amp = rng.uniform(0.01, 0.04, size=6)
    U, V = np.meshgrid(u, v, indexing="ij")
    bump = 0.06 * np.sin(60.0 * np.pi * U) * np.cos(10.0 * V)
    for i in range(6):
        bump += amp[i] * np.sin(2 * np.pi * k[i, 0] * U + ph[i, 0]) * np.cos(k[i, 1] * V + ph[i, 1])
    R = r0[:, None] * (1.0 + bump)
    p = c[:, None, :] + R[..., None] * (np.cos(V)[..., None] * n1[:, None, :] + np.sin(V)[..., None] * n2[:, None, :])
    yshift = p[..., 1].min()  # rest on the floor
    p[..., 1] -= yshift
    axis = c.copy()
    axis[:, 1] -= yshift
    verts = p.reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n_along), np.arange(n_around), indexing="ij")
    a = (i * n_around + j).ravel()
    b = (i * n_around + (j + 1) % n_around).ravel()
    cc = ((i + 1) * n_around + j).ravel()
    d = ((i + 1) * n_around + (j + 1) % n_around).ravel()
    faces = np.concatenate([np.stack([a, cc, b], -1), np.stack([b, cc, d], -1)]).astype(np.int32)
    # make the winding outward with respect to the tube axis
    tri = verts[faces].astype(np.float64)
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    axis_pt = np.concatenate([axis[i.ravel()], axis[i.ravel()]])
    flip = ((tri.mean(1) - axis_pt) * n).sum(-1) < 0
    faces[flip] = faces[flip][:, ::-1]
    return verts, np.ascontiguousarray(faces)


def write_fast(path, verts, faces):
    verts = np.asarray(verts, np.float32)
    faces = np.asarray(faces, np.int32)
    v4 = np.zeros((len(verts), 4), np.float32)
    v4[:, :3] = verts
    with open(path, "wb") as f:
        np.array([len(verts), len(faces)], np.int32).tofile(f)
        v4.tofile(f)
        faces.tofile(f)


def read_fast(path):
    with open(path, "rb") as f:
        nv, nt = np.fromfile(f, np.int32, 2)
        v4 = np.fromfile(f, np.float32, nv * 4).reshape(nv, 4)
        faces = np.fromfile(f, np.int32, nt * 3).reshape(nt, 3)
    return np.ascontiguousarray(v4[:, :3]), faces


def write_obj(path, verts, faces):
    with open(path, "w") as f:
        for x, y, z in np.asarray(verts, np.float64):
            f.write(f"v {x:.6f} {y:.6f} {z:.6f}\n")
        for a, b, c in np.asarray(faces, np.int64) + 1:
            f.write(f"f {a} {b} {c}\n")


def read_obj(path):
    """Positions and the faces of the first shape, polygons fan-triangulated (what the reference keeps of
    tinyobjloader's result, src/raytracer.cpp:138-165).  Independent of the C++ reader in libwapt.so."""
    verts, faces, closed = [], [], False
    with open(path) as f:
        for line in f:
            s = line.split()
            if not s or s[0].startswith("#"):
                continue
            if s[0] == "v":
                verts.append([float(x) for x in s[1:4]])
            elif s[0] == "f" and not closed:
                idx = []
                for tok in s[1:]:
                    k = int(tok.split("/")[0])
                    idx.append(k - 1 if k > 0 else len(verts) + k)
                for i in range(2, len(idx)):
                    faces.append([idx[0], idx[i - 1], idx[i]])
            elif s[0] in ("o", "g") and faces:
                closed = True
    return np.asarray(verts, np.float32), np.asarray(faces, np.int32)
