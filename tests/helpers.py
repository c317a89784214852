"""Shared test utilities: seeded ray sets and image metrics."""
import numpy as np


def random_rays(scene, n, seed, kinds=("camera", "inside", "outside")):
    """A fixed, seeded mixture of rays through the scene's bounding box.

    camera : from the camera position through random points of the box
    inside : random origins inside the box, uniform directions
    outside: origins on a sphere around the box aimed at random points inside it
    """
    rng = np.random.default_rng(seed)
    v = scene.tri_verts.reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    c, ext = 0.5 * (lo + hi), (hi - lo)
    out = []
    per = n // len(kinds)
    for k, kind in enumerate(kinds):
        m = per if k < len(kinds) - 1 else n - per * (len(kinds) - 1)
        target = lo + rng.random((m, 3)) * ext
        if kind == "camera":
            org = np.tile(np.array(list(scene.camera.P), np.float64), (m, 1))
        elif kind == "inside":
            org = lo + rng.random((m, 3)) * ext
            target = org + rng.normal(size=(m, 3))
        else:
            d = rng.normal(size=(m, 3))
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            org = c + d * np.linalg.norm(ext) * 1.5
        d = target - org
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        r = np.zeros((m, 8), np.float32)
        r[:, 0:3], r[:, 3], r[:, 4:7], r[:, 7] = org, 0.001, d, 1.0e7
        out.append(r)
    return np.ascontiguousarray(np.concatenate(out, 0))


def compare_hits(a, b, rel_t=1e-5):
    """Fraction of rays whose primitive ids agree (ties excused) and the worst relative t error among agreeing hits.

    A disagreement is a tie when both report a hit and the two distances differ by at most rel_t*t."""
    both_miss = (a["prim"] < 0) & (b["prim"] < 0)
    same = (a["prim"] == b["prim"]) & (a["obj"] == b["obj"])
    hit_both = (a["prim"] >= 0) & (b["prim"] >= 0)
    dt = np.abs(a["t"].astype(np.float64) - b["t"].astype(np.float64)) / np.maximum(np.abs(b["t"].astype(np.float64)), 1e-30)
    tie = hit_both & ~same & (dt <= rel_t)
    ok = both_miss | same | tie
    agree_hits = hit_both & same
    worst = float(dt[agree_hits].max()) if agree_hits.any() else 0.0
    return float(ok.mean()), worst, int((~ok).sum()), int(tie.sum())


def rel_rmse(a, b, mask=None):
    """sqrt(mean((a-b)^2)) / mean(b) over linear radiance."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if mask is not None:
        a, b = a[mask], b[mask]
    m = float(np.mean(b))
    return float(np.sqrt(np.mean((a - b) ** 2)) / m) if m else 0.0
