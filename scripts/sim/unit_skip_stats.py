"""How often a worker wave's unit of 64 consecutive triangles holds NO triangle that passes a stage of the
Moller-Trumbore test (measurement helper, CPU only): if whole units fail the u test, a wave-level early-out
(one ballot + branch per unit) skips qVec / v / t for that unit - same decisions, same values (the PERPIXEL
kernels' u-first argument, ptw_trace_common.h).  Rays: camera rays and one diffuse bounce from their hits.
usage: python scripts/sim/unit_skip_stats.py [scene] [rays]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "ce"
nrays = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
scene = pkg.Scene()
W = H = 256
cam = scene.build_named(name, W, H)
a = scene.arrays()
tri = a["tri_vertices"]
v0, e1, e2 = tri[:, 0], tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
nt = len(tri)
U = (nt + 63) // 64
print(name, nt, "triangles", U, "units")
rng = np.random.default_rng(1)
c = cam.as_array()
centre, ax, ay, az = c[0:3], c[3:6], c[6:9], c[9:12]
aspect, dist = c[12], c[13]


def stages(o, d):
    """per ray x triangle: pass masks of det, u, uv and the hit distance"""
    p = np.cross(d[:, None, :], e2[None])
    det = (e1[None] * p).sum(-1)
    ok = np.abs(det) >= 1e-7
    inv = 1.0 / np.where(ok, det, 1.0)
    tv = o[:, None, :] - v0[None]
    u = (tv * p).sum(-1) * inv
    pu = ok & (u >= 0) & (u <= 1)
    q = np.cross(tv, e1[None])
    v = (d[:, None, :] * q).sum(-1) * inv
    puv = pu & (v >= 0) & (u + v <= 1)
    t = (e2[None] * q).sum(-1) * inv
    hit = puv & (t > 1e-7)
    return ok, pu, puv, np.where(hit, t, np.inf)


def report(label, o, d):
    ok, pu, puv, t = stages(o, d)
    pad = U * 64 - nt
    def units(m):
        m = np.concatenate([m, np.zeros((len(m), pad), bool)], 1).reshape(len(m), U, 64)
        return m.any(-1)
    uu, uv = units(pu), units(puv)
    print(f"{label}: {len(o)} rays | triangles passing u {pu.mean():.3f}, u and v {puv.mean():.4f} | "
          f"units with NO lane past u {1 - uu.mean():.3f}, none past u and v {1 - uv.mean():.3f}")
    return t


px = rng.uniform(0, W, nrays)
py = rng.uniform(0, H, nrays)
x = (px / W * 2 - 1) * aspect
y = (1 - py / H * 2)
d = ax[None] * x[:, None] + ay[None] * y[:, None] + az[None] * dist
d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.repeat(centre[None], nrays, 0)
t = report("camera rays", o, d)
k = t.argmin(1)
hit = np.isfinite(t.min(1))
print("camera rays that hit a triangle:", hit.mean())
oh = o[hit] + d[hit] * t.min(1)[hit, None]
n = np.cross(e1[k[hit]], e2[k[hit]])
n /= np.linalg.norm(n, axis=1, keepdims=True)
n = np.where(((n * d[hit]).sum(-1) > 0)[:, None], -n, n)
r = rng.normal(size=oh.shape)
r /= np.linalg.norm(r, axis=1, keepdims=True)
db = n + r
db /= np.linalg.norm(db, axis=1, keepdims=True)
report("one diffuse bounce", oh + 1e-6 * n, db)
