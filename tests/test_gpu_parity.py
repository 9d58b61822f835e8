"""GPU parity: the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Tolerance: 1e-12 relative (to max(|ref|, 1)) on per-pixel fp64 radiance sums, plus EXACT
equality of the per-sample RNG word counts - equal word counts mean every hit/miss and lobe
decision of every path matched the reference algorithm.  The GPU differs from the strict-fp64
oracle only by FMA contraction and by ocml's sin/cos/acos (<= 2 ulp), cf. csrc/ptw_device.h.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))


def gpu_render_with_words(pkg, scene, cam, params):
    import torch
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    h, w, spp = params.height, params.width, params.samples_per_pixel
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    words = torch.zeros((spp, h, w), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), words.data_ptr(), stream)
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), cnt.cpu().numpy().astype(np.uint32), words.cpu().numpy().astype(np.uint32)


@pytest.mark.parametrize("name,w,h,spp,seed", [
    ("cornell", 32, 24, 4, 1),
    ("cornell", 17, 9, 3, 12345),
    ("suzanne", 24, 24, 3, 2),
    ("ce", 8, 8, 2, 3),
    ("single-sphere", 24, 16, 3, 4),
    ("multi-sphere", 24, 16, 3, 5),
    ("example1", 24, 16, 3, 6),
    ("bbc-owl", 24, 16, 2, 7),
])
def test_sequential_matches_oracle(pkg, ob, name, w, h, spp, seed):
    scene = pkg.Scene()
    cam = scene.build_named(name, w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=seed)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, params, threads=4)
    rgb, cnt, words = gpu_render_with_words(pkg, scene, cam, params)
    assert np.array_equal(cnt, ref_cnt)
    assert np.array_equal(words, ref_words), "RNG word counts differ: a path decision diverged"
    assert rel_err(rgb, ref_rgb) < TOL


@pytest.mark.parametrize("name,w,h,spp,seed", [
    ("cornell", 32, 24, 4, 1),
    ("suzanne", 24, 24, 3, 2),
    ("example1", 24, 16, 3, 6),
])
def test_perpixel_matches_oracle(pkg, ob, name, w, h, spp, seed):
    scene = pkg.Scene()
    cam = scene.build_named(name, w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=seed,
                                rng_policy=pkg.RNG_PERPIXEL)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, params, threads=4)
    rgb, cnt, words = gpu_render_with_words(pkg, scene, cam, params)
    assert np.array_equal(cnt, ref_cnt)
    assert np.array_equal(words, ref_words)
    assert rel_err(rgb, ref_rgb) < TOL


def test_host_buffer_entry_point(pkg, ob):
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 16, 16)
    params = pkg.default_params(width=16, height=16, samples_per_pixel=2, seed=9)
    rgb, cnt = pkg.render(scene, cam, params)
    ref_rgb, ref_cnt, _, _ = ob.oracle_render(scene.view(), cam, params, threads=2)
    assert np.array_equal(cnt, ref_cnt)
    assert rel_err(rgb, ref_rgb) < TOL


# ---- parameter space: both kernels of each policy, non-default RenderParams ------------------
@pytest.mark.parametrize("policy", [0, 1])
@pytest.mark.parametrize("name,over", [
    ("cornell", dict(max_depth=1)), ("cornell", dict(max_depth=2)), ("cornell", dict(max_depth=7)),
    ("cornell", dict(first_bounce_u=3, first_bounce_v=2)), ("cornell", dict(first_bounce_u=1, first_bounce_v=1)),
    ("cornell", dict(preview=1)), ("cornell", dict(max_depth=0)),
    ("suzanne", dict(max_depth=1)), ("suzanne", dict(max_depth=3, first_bounce_u=2, first_bounce_v=5)),
    ("suzanne", dict(preview=1)), ("ce", dict(max_depth=2)),
    ("single-sphere", dict(max_depth=6)),   # pinhole camera: 2 draws per primary ray
])
def test_non_default_params_match_oracle(pkg, ob, policy, name, over):
    w, h, spp = (12, 9, 2) if name != "ce" else (6, 5, 1)
    scene = pkg.Scene()
    cam = scene.build_named(name, w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=31, rng_policy=policy, **over)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, params, threads=2)
    rgb, cnt, words = gpu_render_with_words(pkg, scene, cam, params)
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words)
    assert rel_err(rgb, ref_rgb) < TOL


@pytest.mark.parametrize("fixture,scene_name", [
    ("f4_cornell_32x32", "cornell"), ("f4_suzanne_32x32", "suzanne"), ("f4_ce_8x8", "ce"),
    ("f4_example1_24x16", "example1"), ("f4_bbc_owl_24x16", "bbc-owl"),
    ("f4_multi_sphere_24x16", "multi-sphere"), ("f4_single_sphere_24x16", "single-sphere")])
def test_sequential_matches_reference_golden(pkg, golden_dir, fixture, scene_name):
    """Directly against the vectors the REFERENCE build produced (no oracle in between)."""
    z = np.load(golden_dir / f"{fixture}.npz")
    w, h, passes, *seeds = z["meta"].tolist()
    scene = pkg.Scene()
    cam = scene.build_named(scene_name, w, h)
    for seed in seeds:
        params = pkg.default_params(width=w, height=h, samples_per_pixel=passes, seed=seed)
        rgb, cnt, words = gpu_render_with_words(pkg, scene, cam, params)
        want = z[f"radiance_seed{seed}"]
        assert np.array_equal(words, z[f"words_seed{seed}"].astype(np.uint32))
        acc = np.zeros_like(want[0])
        for k in range(passes):  # pass-ordered accumulation, as ArrayOutput::operator+=
            acc += want[k]
        assert rel_err(rgb, acc) < TOL and np.all(cnt == passes)


def test_device_rng_known_answers(pkg, ob, golden_dir):
    z = np.load(golden_dir / "f1_rng.npz")
    ctx = pkg.Context(0)
    for seed in (1, 2, 5489, 0xFFFFFFFF):
        assert np.array_equal(ctx.rng_doubles(pkg.RNG_SEQUENTIAL, seed, 700), z[f"unit_{seed}"])
    w = ob.perpixel_words(77, 1234, 64).astype(np.float64)
    want = np.minimum((w[0::2] + w[1::2] * 4294967296.0) / 18446744073709551616.0, np.nextafter(1.0, 0.0))
    assert np.array_equal(ctx.rng_doubles(pkg.RNG_PERPIXEL, 77, 32, pixel=1234), want)


def test_device_intersect_known_answers(pkg, ob, golden_dir):
    """F2: the reference's own intersection tests (test/dod/*Tests.cpp) on the device."""
    z = np.load(golden_dir / "f2_intersect.npz")
    from test_oracle_golden import _scene_from_case
    for name in z["names"]:
        if str(z[f"{name}__which"]) != "all" and np.isfinite(float(z[f"{name}__limit"])):
            continue  # nearerThan-limited variants are host-API only in the reference
        scene = _scene_from_case(pkg, z, name)
        ctx = pkg.Context(0)
        ctx.set_scene(scene)
        got = ctx.intersect(z[f"{name}__ray"])[0]
        want = z[f"{name}__hit"]
        # spheres-only / triangles-only cases have only that primitive kind in the scene
        assert np.allclose(got[:8], want[:8], rtol=1e-14, atol=1e-14), name
        if want[0] >= 0:
            assert np.array_equal(scene.arrays()["materials"][int(got[8])], want[8:17]), name
    # a batch of camera rays against the full Cornell scene, vs the oracle
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 32, 32)
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    rays = np.array([ob.oracle_camera_ray(cam, x, y, 9 + x + 32 * y) for y in range(32) for x in range(32)])
    hits = ctx.intersect(rays)
    want = np.array([ob.oracle_intersect(scene.view(), r) for r in rays])
    assert np.array_equal(hits[:, [0, 1, 8]] >= 0, want[:, [0, 1, 8]] >= 0)
    assert np.allclose(hits, want, rtol=1e-13, atol=1e-13)


# ---- size-independent properties at larger sizes ---------------------------------------------
def _render(pkg, scene, cam, **kw):
    params = pkg.default_params(**kw)
    import torch
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    rgb = torch.zeros((params.height, params.width, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((params.height, params.width), dtype=torch.int32, device="cuda")
    ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), cnt.cpu().numpy()


@pytest.mark.parametrize("policy", [0, 1])
def test_determinism_and_seed_sensitivity(pkg, policy):
    """test/seed_tests.sh: same seed twice -> identical bytes; another seed -> different."""
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 64, 64)
    kw = dict(width=64, height=64, samples_per_pixel=16, rng_policy=policy)
    a, _ = _render(pkg, scene, cam, seed=1, **kw)
    b, _ = _render(pkg, scene, cam, seed=1, **kw)
    c, _ = _render(pkg, scene, cam, seed=2, **kw)
    assert a.tobytes() == b.tobytes()
    assert a.tobytes() != c.tobytes()


@pytest.mark.parametrize("policy", [0, 1])
def test_pass_ranges_compose(pkg, policy):
    """Linearity in passes: rendering passes [0,a) then [a,a+b) into the same framebuffer equals
    rendering [0,a+b) at once (first_pass continues the seed sequence) - bit for bit, because
    the accumulation is pass-ordered."""
    import torch
    scene = pkg.Scene()
    cam = scene.build_named("suzanne", 48, 40)
    whole, cnt = _render(pkg, scene, cam, width=48, height=40, samples_per_pixel=7, seed=5, rng_policy=policy)
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    rgb = torch.zeros((40, 48, 3), dtype=torch.float64, device="cuda")
    c = torch.zeros((40, 48), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for first, n in ((0, 3), (3, 4)):
        p = pkg.default_params(width=48, height=40, samples_per_pixel=n, first_pass=first, seed=5, rng_policy=policy)
        ctx.render(cam, p, rgb.data_ptr(), c.data_ptr(), 0, st)
    torch.cuda.synchronize()
    assert np.array_equal(rgb.cpu().numpy(), whole) and np.all(c.cpu().numpy() == 7) and np.all(cnt == 7)


@pytest.mark.parametrize("policy", [0, 1])
def test_band_size_does_not_change_the_result(pkg, policy, monkeypatch):
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 40, 30)
    kw = dict(width=40, height=30, samples_per_pixel=5, seed=8, rng_policy=policy)
    big, _ = _render(pkg, scene, cam, **kw)
    monkeypatch.setenv("PTW_STAGE_BUDGET_KB", "16")  # 16 KiB staging -> ~136-pixel bands
    small, _ = _render(pkg, scene, cam, **kw)
    assert big.tobytes() == small.tobytes()


def test_ce_is_a_constant_image_at_size(pkg):
    """`ce` as shipped: the camera sits inside the dull light, so every sample of every pixel is
    exactly (0.5675, 0.75, 0.7425) and every ray hits: a full-size
    property test of the nearest-hit search over 3445 primitives."""
    scene = pkg.Scene()
    cam = scene.build_named("ce", 96, 64)
    for policy in (0, 1):
        rgb, cnt = _render(pkg, scene, cam, width=96, height=64, samples_per_pixel=4, seed=1, rng_policy=policy)
        assert np.all(cnt == 4)
        assert np.allclose(rgb / 4, np.broadcast_to((0.5675, 0.75, 0.7425), rgb.shape), rtol=1e-15, atol=0)


def test_policies_agree_statistically(pkg):
    """PERPIXEL is the same estimator with other random numbers: image means agree within noise."""
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 48, 48)
    a, _ = _render(pkg, scene, cam, width=48, height=48, samples_per_pixel=64, seed=1, rng_policy=0)
    b, _ = _render(pkg, scene, cam, width=48, height=48, samples_per_pixel=64, seed=1, rng_policy=1)
    ma, mb = a.mean(axis=(0, 1)) / 64, b.mean(axis=(0, 1)) / 64
    assert np.all(np.abs(ma - mb) < 0.02 * np.maximum(ma, mb))


def test_row_window_perpixel(pkg, ob):
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 20, 12)
    full, _ = _render(pkg, scene, cam, width=20, height=12, samples_per_pixel=3, seed=4, rng_policy=1)
    part, cnt = _render(pkg, scene, cam, width=20, height=12, samples_per_pixel=3, seed=4, rng_policy=1,
                        row_begin=5, row_end=9)
    assert np.array_equal(part[5:9], full[5:9]) and not part[:5].any() and not part[9:].any()
    assert np.all(cnt[5:9] == 3) and not cnt[:5].any() and not cnt[9:].any()


# ---- every kernel variant: random triangle/sphere soups sized to hit each <SLOTS, WAVES> -----
def _soup(pkg, ntri, nsph, seed):
    rng = np.random.default_rng(seed)
    scene = pkg.Scene()
    mats = [pkg.material("diffuse", rng.uniform(0.2, 0.9, 3)),
            pkg.material("light", rng.uniform(0.5, 3.0, 3)),
            pkg.material("glossy", rng.uniform(0.2, 0.9, 3), 1.3, 20.0),
            pkg.material("reflective", rng.uniform(0.2, 0.9, 3), 0.5, 4.0),
            pkg.material("specular", rng.uniform(0.2, 0.9, 3), 1.0)]
    for i in range(ntri):
        c = rng.uniform(-3, 3, 3)
        v = c + rng.uniform(-0.6, 0.6, (3, 3))
        scene.add_triangle(v[0], v[1], v[2], mats[i % len(mats)])
    for i in range(nsph):
        scene.add_sphere(rng.uniform(-3, 3, 3), rng.uniform(0.05, 0.4), mats[(i + 2) % len(mats)])
    scene.add_sphere((0, 0, 0), 12.0, mats[0])  # enclosing shell: long paths
    scene.set_environment_colour((0.1, 0.2, 0.3))
    cam = pkg.set_focus(pkg.look_at((0, 0.5, 7), (0, 0, 0), (0, 1, 0), 10, 8, 45.0), (0, 0, 0), 0.02)
    return scene, cam


@pytest.mark.parametrize("ntri,nsph", [(50, 3), (100, 0), (200, 70), (400, 2), (900, 5), (1800, 1),
                                       (3500, 300), (5000, 9)])
@pytest.mark.parametrize("policy", [0, 1])
def test_kernel_variants_on_random_soups(pkg, ob, ntri, nsph, policy):
    scene, cam = _soup(pkg, ntri, nsph, seed=ntri + nsph)
    params = pkg.default_params(width=10, height=8, samples_per_pixel=2, seed=3, rng_policy=policy)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, params, threads=2)
    rgb, cnt, words = gpu_render_with_words(pkg, scene, cam, params)
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words)
    assert rel_err(rgb, ref_rgb) < TOL


# The speculative sequential kernel (scenes up to 64 triangles, depth up to 9): mixed materials,
# open and closed scenes, every depth, odd fan-outs, several bands.
@pytest.mark.parametrize("ntri,nsph,shell,over", [
    (1, 0, True, dict()), (7, 2, True, dict(max_depth=9)), (64, 0, False, dict(max_depth=3)),
    (40, 20, True, dict(first_bounce_u=5, first_bounce_v=3)), (33, 62, False, dict(max_depth=8, first_bounce_u=2, first_bounce_v=2)),
    (64, 62, True, dict(max_depth=4, first_bounce_u=1, first_bounce_v=7)), (20, 1, True, dict(max_depth=10)),
])
def test_speculative_kernel_on_small_soups(pkg, ob, ntri, nsph, shell, over, monkeypatch):
    monkeypatch.setenv("PTW_STAGE_BUDGET_KB", "40")  # park / resume the stream ring between bands
    rng = np.random.default_rng(1000 + ntri)
    scene = pkg.Scene()
    mats = [pkg.material("diffuse", rng.uniform(0.2, 0.9, 3)), pkg.material("light", rng.uniform(0.5, 3.0, 3)),
            pkg.material("glossy", rng.uniform(0.2, 0.9, 3), 1.3, 20.0),
            pkg.material("reflective", rng.uniform(0.2, 0.9, 3), 0.5, 4.0),
            pkg.material("specular", rng.uniform(0.2, 0.9, 3), 1.0)]
    for i in range(ntri):
        c = rng.uniform(-2, 2, 3)
        v = c + rng.uniform(-1.5, 1.5, (3, 3))
        scene.add_triangle(v[0], v[1], v[2], mats[i % len(mats)])
    for i in range(nsph):
        scene.add_sphere(rng.uniform(-3, 3, 3), rng.uniform(0.1, 0.8), mats[(i + 2) % len(mats)])
    if shell:
        scene.add_sphere((0, 0, 0), 12.0, mats[0])
    scene.set_environment_colour((0.3, 0.2, 0.1))
    cam = pkg.set_focus(pkg.look_at((0, 0.5, 7), (0, 0, 0), (0, 1, 0), 24, 16, 45.0), (0, 0, 0), 0.02)
    params = pkg.default_params(width=24, height=16, samples_per_pixel=5, seed=77, **over)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, params, threads=4)
    rgb, cnt, words = gpu_render_with_words(pkg, scene, cam, params)
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words)
    assert rel_err(rgb, ref_rgb) < TOL


@pytest.mark.parametrize("name,w,h,spp", [("cornell", 96, 96, 64), ("suzanne", 48, 48, 64)])
def test_repeated_renders_are_bytewise_identical(pkg, name, w, h, spp):
    """The sequential kernels pass data between waves through LDS; a race would show up as a run
    that differs (scripts/stress_determinism.py is the longer form of this)."""
    scene = pkg.Scene()
    cam = scene.build_named(name, w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=9)
    first = None
    for _ in range(4):
        rgb, cnt, words = gpu_render_with_words(pkg, scene, cam, params)
        blob = rgb.tobytes() + words.tobytes()
        first = first or blob
        assert blob == first


def test_full_width_strip_matches_oracle(pkg, ob):
    """The headline frame's width (1024) with a few rows: the same pixel -> (x, y) mapping, stream
    lengths of tens of thousands of generator blocks per pass, several bands."""
    w, h, spp = 1024, 40, 3
    scene = pkg.Scene()
    cam = scene.build_named("cornell", w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=1)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, params, threads=3)
    rgb, cnt, words = gpu_render_with_words(pkg, scene, cam, params)
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words)
    assert rel_err(rgb, ref_rgb) < TOL


# ---- edge cases -------------------------------------------------------------------------------
@pytest.mark.parametrize("policy", [0, 1])
def test_edge_cases(pkg, ob, policy):
    # empty scene: every ray leaves to the environment (Scene.cpp:131-133), 8 words per sample
    scene = pkg.Scene()
    scene.set_environment_colour((0.25, 0.5, 0.75))
    cam = pkg.set_focus(pkg.look_at((0, 0, 5), (0, 0, 0), (0, 1, 0), 7, 5, 40.0), (0, 0, 0), 0.05)
    params = pkg.default_params(width=7, height=5, samples_per_pixel=3, seed=2, rng_policy=policy)
    rgb, cnt, words = gpu_render_with_words(pkg, scene, cam, params)
    assert np.all(words == 8) and np.all(cnt == 3)
    assert np.array_equal(rgb, np.broadcast_to(np.array([0.25, 0.5, 0.75]) * 3, rgb.shape))
    # 1 x 1 frame, one pass; and a tall 1-pixel-wide frame
    for (w, h, spp) in ((1, 1, 1), (1, 9, 2), (13, 1, 2)):
        scene = pkg.Scene()
        cam = scene.build_named("cornell", w, h)
        params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=11, rng_policy=policy)
        ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, params, threads=1)
        rgb, cnt, words = gpu_render_with_words(pkg, scene, cam, params)
        assert np.array_equal(words, ref_words) and np.array_equal(cnt, ref_cnt)
        assert rel_err(rgb, ref_rgb) < TOL
    # zero passes: nothing happens, buffers untouched
    params = pkg.default_params(width=13, height=1, samples_per_pixel=0, seed=11, rng_policy=policy)
    rgb0, cnt0 = pkg.render(scene, cam, params, rgb_sum=np.full((1, 13, 3), 2.5), counts=np.full((1, 13), 4, np.uint32))
    assert np.all(rgb0 == 2.5) and np.all(cnt0 == 4)


def test_many_passes_small_frame(pkg, ob):
    """More passes than the chip has SIMDs (queued workgroups), accumulated in pass order."""
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 4, 3)
    params = pkg.default_params(width=4, height=3, samples_per_pixel=1500, seed=1)
    rgb, cnt = pkg.render(scene, cam, params)
    ref_rgb, ref_cnt, _, _ = ob.oracle_render(scene.view(), cam, params, threads=8, want_words=False)
    assert np.array_equal(cnt, ref_cnt) and rel_err(rgb, ref_rgb) < TOL


def test_argument_errors(pkg):
    scene = pkg.Scene()
    cam = scene.build_named("single-sphere", 4, 4)
    for over, status in ((dict(max_depth=65), 8), (dict(samples_per_pixel=-1), 1),
                         (dict(rng_policy=7), 1), (dict(row_begin=3, row_end=2), 1), (dict(device=99), 2)):
        kw = dict(width=4, height=4, samples_per_pixel=1, seed=1)
        kw.update(over)
        with pytest.raises(pkg.PtwError) as e:
            pkg.render(scene, cam, pkg.default_params(**kw),
                       rgb_sum=np.zeros((4, 4, 3)), counts=np.zeros((4, 4), np.uint32))
        assert e.value.status == status, (over, e.value)
    bad = pkg.default_params(width=0, height=4, samples_per_pixel=1, seed=1)
    view = scene.view()
    buf = np.zeros(64)
    assert pkg.lib.ptw_render(C.byref(view), C.byref(cam), C.byref(bad), buf.ctypes.data, buf.ctypes.data,
                              None, None) == 1
    ctx = pkg.Context(0)
    with pytest.raises(pkg.PtwError):  # render before set_scene
        ctx.render(cam, pkg.default_params(width=4, height=4, samples_per_pixel=1, seed=1), 1, 1)
