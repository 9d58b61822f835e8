"""GPU: the accelerated (BVH) mode of SURVEY.md section 8 f4 - separate from the reference's brute
force, but held to the same parity bar: it culls tests, it must not change a single bit of any
sample (fp64 sums, counts and per-sample RNG word counts equal to the brute-force kernels')."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_round2 import device_render  # noqa: E402


def both(pkg, scene, cam, **kw):
    base = pkg.default_params(rng_policy=pkg.RNG_PERPIXEL, **kw)
    accel = pkg.default_params(rng_policy=pkg.RNG_PERPIXEL, accel=pkg.ACCEL_BVH, **kw)
    return device_render(pkg, scene, cam, base, want_words=True), device_render(pkg, scene, cam, accel, want_words=True)


@pytest.mark.parametrize("name,w,h,spp", [("cornell", 40, 30, 4), ("suzanne", 48, 48, 4), ("ce", 24, 24, 2),
                                         ("example1", 32, 24, 3), ("bbc-owl", 32, 24, 3), ("multi-sphere", 24, 16, 3)])
def test_bvh_mode_is_bit_identical_to_brute_force(pkg, name, w, h, spp):
    scene = pkg.Scene()
    cam = scene.build_named(name, w, h)
    (rgb, cnt, words), (rgb2, cnt2, words2) = both(pkg, scene, cam, width=w, height=h, samples_per_pixel=spp, seed=5)
    assert np.array_equal(cnt, cnt2) and np.array_equal(words, words2)
    assert np.array_equal(rgb, rgb2)


def test_bvh_mode_matches_oracle_and_reports_itself(pkg, ob):
    import torch
    scene = pkg.Scene()
    cam = scene.build_named("suzanne", 20, 20)
    p = pkg.default_params(width=20, height=20, samples_per_pixel=2, seed=9, rng_policy=1, accel=pkg.ACCEL_BVH)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, p, threads=2)  # the oracle is brute force
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    ctx._has_scene = True
    ctx.enable_stats(True)
    rgb, cnt, words = device_render(pkg, scene, cam, p, want_words=True, ctx=ctx)
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words)
    assert float(np.max(np.abs(rgb - ref_rgb) / np.maximum(np.abs(ref_rgb), 1.0))) < 1e-12
    assert ctx.stats(reset=True).trace_kernel.decode() == "tracePerPixelBvh"
    # SEQUENTIAL + accel: refused, not silently ignored
    with pytest.raises(pkg.PtwError) as e:
        pkg.render(scene, cam, pkg.default_params(width=20, height=20, samples_per_pixel=1, seed=9, accel=1))
    assert e.value.status == 8


def test_bvh_mode_ties_and_degenerate_scenes(pkg):
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
    (rgb, cnt, words), (rgb2, cnt2, words2) = both(pkg, scene, cam, width=36, height=28, samples_per_pixel=3, seed=2)
    assert np.array_equal(rgb, rgb2) and np.array_equal(words, words2) and np.array_equal(cnt, cnt2)
    for build in ("one", "spheres", "empty"):
        s2 = pkg.Scene()
        if build == "one":
            s2.add_triangle((-1, -1, 3), (1, -1, 3), (0, 1, 3), mats[1])
        if build == "spheres":
            s2.add_sphere((0, 0, 3), 1.0, mats[0])
            s2.add_sphere((1, 0.5, 2), 0.3, mats[1])
        s2.set_environment_colour((0.1, 0.1, 0.1))
        (a, ac, aw), (b, bc, bw) = both(pkg, s2, cam, width=36, height=28, samples_per_pixel=2, seed=4)
        assert np.array_equal(a, b) and np.array_equal(aw, bw) and np.array_equal(ac, bc)
