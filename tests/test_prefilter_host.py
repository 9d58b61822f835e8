"""CPU: the fp32 PREFILTER of PTW_ACCEL_PREFILTER never rejects a triangle the reference's fp64 test accepts.

The device kernel (PixCtxT::intersectPrefiltered, csrc/ptw_pix_ctx.h) skips the fp64 Moller-Trumbore test of
src/dod/Scene.cpp:62-98 for a triangle only when its fp32 evaluation PROVES the rejection (host/prefilter.h,
DESIGN.md 3.4).  Here the very records the kernel reads (ptw_scene_prefilter_records) are run through the same
fp32 expression trees in numpy - unfused products and sums: the same number of roundings on the deepest path as
the device's fma form, so the same bound - against the fp64 test in the reference's operation order, on rays
chosen to sit ON the decision boundaries: aimed at edges and vertices, and a relative 1e-12 ... 1e-4 to
either side of them, from near and far, on triangles from 1e-6 to 1e4 in size, slivers, and the bundled meshes.
"""
import numpy as np
import pytest

EPS = 1e-9   # Epsilon, src/math/Epsilon.h:3 (kEpsilon)
f32 = np.float32


def exact_uv_accepts(o, d, v0, e1, e2):
    """Scene.cpp:62-98 up to the (u, v) rejection, vectorised over rays x triangles in fp64."""
    pvec = np.cross(d[:, None, :], e2[None, :, :])
    det = np.einsum("tk,rtk->rt", e1, pvec)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = 1.0 / det
        tvec = o[:, None, :] - v0[None, :, :]
        u = np.einsum("rtk,rtk->rt", tvec, pvec) * inv
        qvec = np.cross(tvec, e1[None, :, :])
        v = np.einsum("rk,rtk->rt", d, qvec) * inv
        rejected = (np.abs(det) < EPS) | (u < 0.0) | (u > 1.0) | (v < 0.0) | (u + v > 1.0)
    return ~rejected


def prefilter_keeps(o, d, records, ntri):
    """The device's fp32 look, from the library's own pair records: True where the triangle goes to the fp64 test."""
    rec = records.reshape(-1, 11, 2)                       # [pair][field][A|B]
    flat = rec.transpose(0, 2, 1).reshape(-1, 11)[:ntri]   # [triangle][v0x v0y v0z e1x.. e2z EA EB]
    v0, e1, e2, ea, eb = flat[:, 0:3], flat[:, 3:6], flat[:, 6:9], flat[:, 9], flat[:, 10]
    of, df = o.astype(f32), d.astype(f32)
    omax = np.max(np.abs(of), axis=1) * f32(1.0 + 2.0 ** -22)

    def cross(a, b):
        return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                         a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                         a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)

    def dot(a, b):
        return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]

    tvec = of[:, None, :] - v0[None, :, :]
    pvec = cross(df[:, None, :], e2[None, :, :])
    det = dot(e1[None, :, :], pvec)
    un = dot(tvec, pvec)
    qvec = cross(tvec, e1[None, :, :])
    vn = dot(df[:, None, :], qvec)
    wn = (det - un) - vn
    assert un.dtype == f32 and wn.dtype == f32
    e = omax[:, None] * eb[None, :] + ea[None, :]
    mn = np.minimum(np.minimum(un, vn), wn)
    mx = np.maximum(np.maximum(un, vn), wn)
    r = np.maximum(mn + e, e - mx)
    return ~(r < 0)


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def scene_of(pkg, tris):
    scene = pkg.Scene()
    mat = pkg.material("diffuse", (0.5, 0.5, 0.5))
    for t in tris:
        scene.add_triangle(t[0], t[1], t[2], mat)
    return scene


def boundary_rays(rng, tris, per_tri, scale_o):
    """Rays through points on / next to the edges and vertices of the triangles (barycentric coordinates at the
    decision boundaries u = 0, v = 0, u + v = 1, u = 1 and a relative 1e-12 .. 1e-4 to either side)."""
    n = len(tris)
    pick = rng.integers(0, n, per_tri * n)
    t = tris[pick]
    offs = np.concatenate([[0.0], 10.0 ** rng.uniform(-12, -4, 7)]) * rng.choice([-1.0, 1.0], 8)
    kind = rng.integers(0, 4, len(pick))
    a = rng.uniform(0, 1, len(pick))
    eps = rng.choice(offs, len(pick))
    u = np.where(kind == 0, eps, np.where(kind == 1, a, np.where(kind == 2, a, 1.0 + eps)))
    v = np.where(kind == 0, a, np.where(kind == 1, eps, np.where(kind == 2, 1.0 - a + eps, -eps * a)))
    target = t[:, 0] + u[:, None] * (t[:, 1] - t[:, 0]) + v[:, None] * (t[:, 2] - t[:, 0])
    o = target + unit(rng.normal(size=(len(pick), 3))) * (10.0 ** rng.uniform(-3, 1, (len(pick), 1))) * scale_o
    return o, unit(target - o)


def check(pkg, tris, o, d, min_reject=None):
    scene = scene_of(pkg, tris)
    records, usable = scene.prefilter_records()
    assert usable and records.shape == ((len(tris) + 1) // 2, 22)
    v0 = tris[:, 0]
    e1, e2 = tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    worst = 0
    rejected = total = 0
    for s in range(0, len(o), 512):                      # (rays x triangles in blocks: memory)
        oo, dd = o[s:s + 512], d[s:s + 512]
        acc = exact_uv_accepts(oo, dd, v0, e1, e2)
        keep = prefilter_keeps(oo, dd, records, len(tris))
        worst += int(np.count_nonzero(acc & ~keep))
        rejected += int(np.count_nonzero(~keep))
        total += keep.size
    assert worst == 0, f"{worst} (ray, triangle) pairs: fp32 rejected what the fp64 test accepts"
    if min_reject is not None:
        assert rejected / total >= min_reject, f"the prefilter proves only {rejected / total:.3f} of the tests"
    return rejected / total


@pytest.mark.parametrize("size", [1e-6, 1e-3, 1.0, 1e4])
def test_rays_on_the_decision_boundaries(pkg, size):
    rng = np.random.default_rng(int(-np.log10(size)) + 20)
    centres = rng.uniform(-3, 3, (120, 1, 3)) * max(size, 1.0)
    tris = centres + rng.uniform(-1, 1, (120, 3, 3)) * size
    o, d = boundary_rays(rng, tris, 40, max(size, 1e-3))
    check(pkg, tris, o, d)


def test_slivers_and_degenerate_triangles(pkg):
    rng = np.random.default_rng(5)
    a = rng.uniform(-2, 2, (200, 3))
    dirs = unit(rng.normal(size=(200, 3)))
    tris = np.stack([a, a + dirs, a + dirs * rng.uniform(0.2, 2.0, (200, 1)) +
                     rng.normal(size=(200, 3)) * 10.0 ** rng.uniform(-14, -3, (200, 1))], axis=1)
    tris[::17, 2] = tris[::17, 1]            # exactly degenerate: two vertices coincide
    o, d = boundary_rays(rng, tris, 30, 1.0)
    check(pkg, tris, o, d)


@pytest.mark.parametrize("name,min_reject", [("suzanne", 0.97), ("ce", 0.99), ("cornell", 0.5)])
def test_bundled_meshes_random_and_boundary_rays(pkg, name, min_reject):
    """The scenes of BASELINE cfg2-cfg4: random rays from inside the scene's bounds plus boundary rays - and the
    prefilter must be worth having: it proves the rejection of nearly all tests on the two large meshes."""
    rng = np.random.default_rng(3)
    scene = pkg.Scene()
    scene.build_named(name, 8, 8)
    tris = scene.arrays()["tri_vertices"]
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    o = rng.uniform(lo - 1.0, hi + 1.0, (1024, 3))
    d = unit(rng.normal(size=(1024, 3)))
    frac = check(pkg, tris, o, d, min_reject)
    sub = tris[rng.integers(0, len(tris), min(len(tris), 200))]
    bo, bd = boundary_rays(rng, sub, 10, 1.0)
    check(pkg, tris, bo, bd)
    print(f"{name}: the fp32 look proves {100 * frac:.2f} % of the tests of random rays")


def test_far_origins_and_large_coordinates(pkg):
    rng = np.random.default_rng(8)
    tris = rng.uniform(-1, 1, (100, 1, 3)) * 1e6 + rng.uniform(-1, 1, (100, 3, 3)) * 10.0 ** rng.uniform(-2, 4, (100, 1, 1))
    o, d = boundary_rays(rng, tris, 30, 1e5)
    check(pkg, tris, o, d)


def test_the_mode_is_refused_beyond_the_coordinate_bound(pkg):
    tris = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 5e12], [1, 0, 5e12], [0, 1, 5e12]]], dtype=float)
    _, usable = scene_of(pkg, tris).prefilter_records()
    assert not usable
    _, usable = scene_of(pkg, tris[:1]).prefilter_records()
    assert usable


def test_record_layout_and_rounding(pkg):
    rng = np.random.default_rng(1)
    tris = rng.uniform(-2, 2, (5, 3, 3))                  # odd count: the last record's B half repeats A
    rec, usable = scene_of(pkg, tris).prefilter_records()
    assert usable and rec.shape == (3, 22)
    geo = np.concatenate([tris[:, 0], tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]], axis=1)   # v0 e1 e2
    for i in range(5):
        k, h = divmod(i, 2)
        assert np.array_equal(rec[k, 0:18].reshape(9, 2)[:, h], geo[i].astype(f32))
        a1, a2 = np.abs(geo[i, 3:6]).sum(), np.abs(geo[i, 6:9]).sum()
        ea = 1e-6 * (a1 * a2 + 2 * np.abs(geo[i, 0:3]).max() * (a1 + a2)) + 1e-12
        eb = 2e-6 * (a1 + a2)
        assert rec[k, 18 + h] >= ea and rec[k, 20 + h] >= eb            # rounded UP ...
        assert rec[k, 18 + h] <= ea * (1 + 3e-7) and rec[k, 20 + h] <= eb * (1 + 3e-7)   # ... by at most an ulp
    assert np.array_equal(rec[2, 0:22].reshape(11, 2)[:, 1], rec[2, 0:22].reshape(11, 2)[:, 0])
