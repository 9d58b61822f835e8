"""GPU: the accelerated modes of SURVEY.md section 8 f4 - the BVH-culled search and the conservative fp32
PREFILTER - separate from the reference's brute force, but held to the same parity bar: they skip work, they
must not change a single bit of any sample (fp64 sums, counts and per-sample RNG word counts equal to the
brute-force kernels')."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_round2 import device_render  # noqa: E402


# (the prefilter has the PERPIXEL policy's two forms: persistent - the default - and lock step)
MODES = ["bvh", "prefilter", "prefilter-lockstep"]
KERNEL = {"bvh": "tracePerPixelBvh", "prefilter": "tracePerPixelPersistentPrefilter", "prefilter-lockstep": "tracePerPixelPrefilter"}


def accel_of(pkg, mode):
    if mode == "bvh":
        return dict(accel=pkg.ACCEL_BVH)
    return dict(accel=pkg.ACCEL_PREFILTER, pix_kernel=pkg.PIX_KERNEL_LOCKSTEP if mode.endswith("lockstep") else pkg.PIX_KERNEL_AUTO)


def both(pkg, scene, cam, mode="bvh", **kw):
    base = pkg.default_params(rng_policy=pkg.RNG_PERPIXEL, **kw)
    accel = pkg.default_params(rng_policy=pkg.RNG_PERPIXEL, **accel_of(pkg, mode), **kw)
    return device_render(pkg, scene, cam, base, want_words=True), device_render(pkg, scene, cam, accel, want_words=True)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name,w,h,spp", [("cornell", 40, 30, 4), ("suzanne", 48, 48, 4), ("ce", 24, 24, 2),
                                         ("example1", 32, 24, 3), ("bbc-owl", 32, 24, 3), ("multi-sphere", 24, 16, 3),
                                         ("single-sphere", 24, 16, 3)])
def test_accel_mode_is_bit_identical_to_brute_force(pkg, name, w, h, spp, mode):
    scene = pkg.Scene()
    cam = scene.build_named(name, w, h)
    (rgb, cnt, words), (rgb2, cnt2, words2) = both(pkg, scene, cam, mode, width=w, height=h, samples_per_pixel=spp, seed=5)
    assert np.array_equal(cnt, cnt2) and np.array_equal(words, words2)
    assert np.array_equal(rgb, rgb2)


@pytest.mark.parametrize("mode", MODES)
def test_accel_mode_matches_oracle_and_reports_itself(pkg, ob, mode):
    import torch
    scene = pkg.Scene()
    cam = scene.build_named("suzanne", 20, 20)
    p = pkg.default_params(width=20, height=20, samples_per_pixel=2, seed=9, rng_policy=1, **accel_of(pkg, mode))
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, p, threads=2)  # the oracle is brute force
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    ctx._has_scene = True
    ctx.enable_stats(True)
    rgb, cnt, words = device_render(pkg, scene, cam, p, want_words=True, ctx=ctx)
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words)
    assert float(np.max(np.abs(rgb - ref_rgb) / np.maximum(np.abs(ref_rgb), 1.0))) < 1e-12
    assert ctx.stats(reset=True).trace_kernel.decode() == KERNEL[mode]
    # SEQUENTIAL + BVH: refused, not silently ignored (SEQUENTIAL + PREFILTER is the worker-wave kernels' own form:
    # tests/test_gpu_round6.py)
    if mode == "bvh":
        with pytest.raises(pkg.PtwError) as e:
            pkg.render(scene, cam, pkg.default_params(width=20, height=20, samples_per_pixel=1, seed=9, **accel_of(pkg, mode)))
        assert e.value.status == 8


@pytest.mark.parametrize("mode", MODES)
def test_accel_mode_ties_and_degenerate_scenes(pkg, mode):
    """Coincident and duplicated triangles (exact ties in t: the lowest insertion index must win, and
    its material decides the image), a single triangle, spheres only, an empty scene."""
    rng = np.random.default_rng(3)
    scene = pkg.Scene()
    mats = [pkg.material("diffuse", (0.9, 0.1, 0.1)), pkg.material("light", (3, 2, 1)),
            pkg.material("diffuse", (0.1, 0.9, 0.1)), pkg.material("glossy", (0.5, 0.5, 0.9), 1.3, 20.0)]
    tris = rng.uniform(-1, 1, (40, 3, 3))
    tris[:, :, 2] += 3.0
    for k, t in enumerate(tris):
        scene.add_triangle(*t, mats[k % 4])
    for k in (3, 7, 11, 3):           # duplicates with OTHER materials, inserted later: must never win a tie
        scene.add_triangle(*tris[k], mats[(k + 1) % 4])
    scene.add_sphere((0.2, 0.1, 2.5), 0.4, mats[3])
    scene.set_environment_colour((0.2, 0.3, 0.4))
    cam = pkg.look_at((0, 0, -1), (0, 0, 3), (0, 1, 0), 36, 28, 60.0)
    (rgb, cnt, words), (rgb2, cnt2, words2) = both(pkg, scene, cam, mode, width=36, height=28, samples_per_pixel=3, seed=2)
    assert np.array_equal(rgb, rgb2) and np.array_equal(words, words2) and np.array_equal(cnt, cnt2)
    for build in ("one", "spheres", "empty"):
        s2 = pkg.Scene()
        if build == "one":
            s2.add_triangle((-1, -1, 3), (1, -1, 3), (0, 1, 3), mats[1])
        if build == "spheres":
            s2.add_sphere((0, 0, 3), 1.0, mats[0])
            s2.add_sphere((1, 0.5, 2), 0.3, mats[1])
        s2.set_environment_colour((0.1, 0.1, 0.1))
        (a, ac, aw), (b, bc, bw) = both(pkg, s2, cam, mode, width=36, height=28, samples_per_pixel=2, seed=4)
        assert np.array_equal(a, b) and np.array_equal(aw, bw) and np.array_equal(ac, bc)


def _wall_scene(pkg, n=16):
    """A triangulated wall z = 4 over [-1, 1]^2 (n x n cells, two triangles each, four materials), one very large
    and one very small triangle, inside a shell sphere."""
    scene = pkg.Scene()
    mats = [pkg.material("diffuse", (0.8, 0.3, 0.2)), pkg.material("diffuse", (0.2, 0.8, 0.3)),
            pkg.material("light", (1.5, 1.4, 1.2)), pkg.material("glossy", (0.4, 0.4, 0.9), 1.3, 15.0)]
    xs = np.linspace(-1.0, 1.0, n + 1)
    tris = []
    for i in range(n):
        for j in range(n):
            a, b = (xs[i], xs[j], 4.0), (xs[i + 1], xs[j], 4.0)
            c, d = (xs[i], xs[j + 1], 4.0), (xs[i + 1], xs[j + 1], 4.0)
            scene.add_triangle(a, b, c, mats[(i + j) % 4])
            scene.add_triangle(b, d, c, mats[(i + 2 * j + 1) % 4])
            tris += [(a, b, c), (b, d, c)]
    big = ((-2e5, -1e5, 9e4), (2e5, -1e5, 9e4), (0.0, 2e5, 9e4))
    small = ((1e-5, 1e-5, 2.0), (3e-5, 1e-5, 2.0), (1e-5, 3e-5, 2.0))
    scene.add_triangle(*big, mats[0])
    scene.add_triangle(*small, mats[1])
    tris += [big, small]
    scene.add_sphere((0, 0, 0), 3e5, mats[0])
    scene.set_environment_colour((0.05, 0.05, 0.1))
    return scene, np.asarray(tris, dtype=float)


def test_prefilter_on_rays_that_graze_edges(pkg, ob):
    """The adversarial case for the fp32 prefilter, through the device's known-answer entry
    (ptw_context_intersect with ptw_debug_options.intersect_accel): 60 000 rays aimed AT the edges and vertices of
    a triangulated wall - barycentric coordinates exactly 0 or 1 and a relative 1e-15 ... 1e-5 to either side -
    from near and far, plus a very large and a very small triangle.  Which triangle such a ray hits - the one on
    this side of the edge, the one on the other side, or neither - is decided in the last bits of the fp64 test
    (src/dod/Scene.cpp:89); the fp32 look must hand every such pair over.  The nine doubles of every hit record
    (distance, side, position, normal, material) equal the brute-force search's bit for bit; away from the exact
    boundary (offsets of 1e-10 and more, where the device's contracted fp64 and the oracle's strict fp64 cannot
    disagree about a decision) they equal the oracle's."""
    scene, tris = _wall_scene(pkg)
    rng = np.random.default_rng(12)
    n = 60000
    t = tris[rng.integers(0, len(tris), n)]
    offs = np.concatenate([[0.0, 0.0], 10.0 ** rng.uniform(-15, -5, 14)]) * np.tile([1.0, -1.0], 8)
    kind = rng.integers(0, 4, n)
    a = rng.uniform(0, 1, n)
    eps = rng.choice(offs, n)
    u = np.where(kind == 0, eps, np.where(kind == 1, a, np.where(kind == 2, a, 1.0 + eps)))
    v = np.where(kind == 0, a, np.where(kind == 1, eps, np.where(kind == 2, 1.0 - a + eps, -eps * a)))
    target = t[:, 0] + u[:, None] * (t[:, 1] - t[:, 0]) + v[:, None] * (t[:, 2] - t[:, 0])
    dirs = rng.normal(size=(n, 3))
    dirs[:, 2] = -np.abs(dirs[:, 2]) - 0.2                                  # origins in front of the wall
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    o = target + dirs * 10.0 ** rng.uniform(-2, 1.5, (n, 1))
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1)
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    brute = ctx.intersect(rays)
    ctx.set_debug(intersect_accel=pkg.ACCEL_PREFILTER)
    pre = ctx.intersect(rays)
    assert np.array_equal(brute, pre), f"{int(np.count_nonzero(np.any(brute != pre, axis=1)))} rays found another hit"
    hit_wall = np.count_nonzero(np.abs(brute[:, 4] - 4.0) < 1e-9)
    assert 0.3 * n < hit_wall < n                                            # (on the boundary: some hit, some slip by)
    view = scene.view()
    clear = np.flatnonzero((np.abs(eps) >= 1e-10) & (kind != 3))   # (kind 3 sits next to a vertex: |v| = |eps| a)
    for i in rng.choice(clear, 3000, replace=False):                         # the oracle: distance and material
        ref = ob.oracle_intersect(view, rays[i])
        if ref[0] < 0:
            assert pre[i, 0] < 0
        else:
            assert abs(pre[i, 0] - ref[0]) <= 1e-12 * max(1.0, abs(ref[0])) and pre[i, 8] == ref[8]


def test_prefilter_refuses_scenes_beyond_its_coordinate_bound(pkg):
    """fp32 products of coordinates beyond 1e12 could overflow, and an infinity would break the bound's argument:
    the mode says PTW_ERR_UNSUPPORTED instead of rendering (the brute-force kernels take the scene as it is)."""
    scene = pkg.Scene()
    mat = pkg.material("diffuse", (0.5, 0.5, 0.5))
    scene.add_triangle((-1, -1, 3), (1, -1, 3), (0, 1, 3), mat)
    scene.add_triangle((-1, -1, 5e12), (1, -1, 5e12), (0, 1, 5e12), mat)
    cam = pkg.look_at((0, 0, 0), (0, 0, 3), (0, 1, 0), 8, 8, 50.0)
    with pytest.raises(pkg.PtwError) as e:
        pkg.render(scene, cam, pkg.default_params(width=8, height=8, samples_per_pixel=1, seed=1, rng_policy=pkg.RNG_PERPIXEL,
                                                  accel=pkg.ACCEL_PREFILTER))
    assert e.value.status == 8 and "1e12" in str(e.value)
    rgb, cnt = pkg.render(scene, cam, pkg.default_params(width=8, height=8, samples_per_pixel=1, seed=1, rng_policy=pkg.RNG_PERPIXEL))
    assert int(cnt.sum()) == 64
