"""CPU: the C restatement (oracle/ptw_oracle.c) against the committed golden vectors that
oracle/make_golden.py generated from the reference's own compiled sources.

The strict oracle build and the strict reference build execute the same IEEE-754 operations in
the same order and call the same libm, so the comparison is bit-exact; EXACT is relaxed to 1e-13
relative only if the host's libm differs from the one the fixtures were generated with.
"""
import ctypes as C

import numpy as np
import pytest


def close(a, b, tol=1e-13):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if np.array_equal(a, b):
        return True
    return bool(np.all(np.abs(a - b) <= tol * np.maximum(np.abs(b), 1.0)))


def test_f1_rng_words_and_doubles(ob, golden_dir):
    z = np.load(golden_dir / "f1_rng.npz")
    for seed in (1, 2, 5489, 0xFFFFFFFF):
        assert np.array_equal(ob.mt_words(seed, 1400), z[f"words_{seed}"])
        assert np.array_equal(ob.mt_unit_doubles(seed, 700), z[f"unit_{seed}"])
    # anchors recorded in SURVEY.md section 8c
    assert ob.mt_words(1, 4).tolist() == [1791095845, 4282876139, 3093770124, 4005303368]
    assert ob.mt_unit_doubles(1, 2).tolist() == [0.99718480823026556, 0.93255736136816547]


def test_canonical_is_below_one_and_uses_two_words(ob):
    d = ob.mt_unit_doubles(12345, 5000)
    w = ob.mt_words(12345, 10000).astype(np.float64)
    expect = (w[0::2] + w[1::2] * 4294967296.0) / 18446744073709551616.0
    expect = np.minimum(expect, np.nextafter(1.0, 0.0))
    assert np.array_equal(d, expect)
    assert d.max() < 1.0 and d.min() >= 0.0


def _scene_from_case(pkg, z, name):
    scene = pkg.Scene()
    mat_a = pkg.material("diffuse", (1, 1, 1))
    mat_b = pkg.material("diffuse", (1, 0, 0))
    for i, s in enumerate(z[f"{name}__spheres"]):
        scene.add_sphere(s[:3], s[3], mat_a if i == 0 else mat_b)
    for t in z[f"{name}__tris"]:
        scene.add_triangle(t[0], t[1], t[2], mat_a)
    return scene


def test_f2_intersection_known_answers(pkg, ob, golden_dir):
    z = np.load(golden_dir / "f2_intersect.npz")
    for name in z["names"]:
        scene = _scene_from_case(pkg, z, name)
        view = scene.view()
        which = str(z[f"{name}__which"])
        got = ob.oracle_intersect(view, z[f"{name}__ray"], which, float(z[f"{name}__limit"]))
        want = z[f"{name}__hit"]
        assert close(got[:8], want[:8]), name
        if want[0] >= 0:  # material: the reference returns the MaterialSpec, we return an index
            mats = scene.arrays()["materials"]
            assert np.array_equal(mats[int(got[8])], want[8:17]), name


def test_f2_reference_test_assertions(golden_dir):
    """The Catch2 assertions of test/dod/*Tests.cpp, applied to the recorded reference output."""
    z = np.load(golden_dir / "f2_intersect.npz")
    approx = lambda a, b: np.allclose(a, b, rtol=1e-5, atol=1e-4)  # Approx / ApproxVec3 (1e-4)
    assert z["sphere_miss_up__hit"][0] < 0 and z["sphere_miss_behind__hit"][0] < 0
    h = z["sphere_hit__hit"]
    assert approx(h[0], 22.416738) and approx(h[2:5], (5.99108, 11.9822, 17.9732))
    assert approx(h[5:8], (-0.267261, -0.534522, -0.801784)) and h[1] == 0
    assert z["sphere_hit_limited__hit"][0] < 0
    h = z["sphere_known_point__hit"]
    assert h[0] == 20 and approx(h[2:5], (0, 0, 20)) and approx(h[5:8], (0, 0, -1)) and h[1] == 0
    h = z["sphere_from_inside__hit"]
    assert h[0] == 10 and approx(h[2:5], (0, 0, 20)) and approx(h[5:8], (0, 0, 1)) and h[1] == 1
    assert z["two_spheres_first_nearer__hit"][0] == 20 and z["two_spheres_first_nearer__hit"][11:14].tolist() == [1, 1, 1]
    assert z["two_spheres_second_nearer__hit"][0] == 20 and z["two_spheres_second_nearer__hit"][11:14].tolist() == [1, 0, 0]
    for name in ("tri_cw_hit", "tri_ccw_hit"):
        h = z[f"{name}__hit"]
        assert approx(h[0], 3.0) and approx(h[2:5], (0, 0, 3)) and approx(h[5:8], (0, 0, -1))
    assert z["tri_cw_miss_up__hit"][0] < 0 and z["tri_cw_miss_behind__hit"][0] < 0
    assert z["tri_cw_hit_limited__hit"][0] < 0


def test_f8_camera_rays(pkg, ob, golden_dir):
    z = np.load(golden_dir / "f8_camera.npz")
    for name, d in ob.SCENE_CAMERAS.items():
        cam = ob.oracle_camera(d["eye"], d["look_at"], d["up"], 64, 48, d["fov"],
                               d.get("focus"), d.get("aperture", 0.0))
        for row in z[name]:
            px, py, seed = int(row[0]), int(row[1]), int(row[2])
            assert close(ob.oracle_camera_ray(cam, px, py, seed), row[3:]), (name, px, py)


F4 = [("f4_cornell_32x32", "cornell", {}), ("f4_suzanne_32x32", "suzanne", {}),
      ("f4_ce_8x8", "ce", {}), ("f4_example1_24x16", "example1", {}),
      ("f4_bbc_owl_24x16", "bbc-owl", {}), ("f4_multi_sphere_24x16", "multi-sphere", {}),
      ("f4_single_sphere_24x16", "single-sphere", {})]


@pytest.mark.parametrize("fixture,scene_name,over", F4)
def test_f4_radiance_and_word_counts(pkg, ob, golden_dir, fixture, scene_name, over):
    z = np.load(golden_dir / f"{fixture}.npz")
    w, h, passes, *seeds = z["meta"].tolist()
    scene = pkg.Scene()
    cam = scene.build_named(scene_name, w, h)
    view = scene.view()
    for seed in seeds:
        params = pkg.default_params(width=w, height=h, samples_per_pixel=passes, seed=seed, **over)
        for k in range(passes):
            rad, words = ob.oracle_render_pass(view, cam, params, k)
            assert np.array_equal(words, z[f"words_seed{seed}"][k]), (fixture, seed, k)
            assert close(rad, z[f"radiance_seed{seed}"][k]), (fixture, seed, k)


@pytest.mark.parametrize("prefix,over", [
    ("fb3x2", dict(first_bounce_u=3, first_bounce_v=2)), ("depth7", dict(max_depth=7)),
    ("depth1", dict(max_depth=1)), ("preview", dict(preview=1))])
def test_f4_non_default_params(pkg, ob, golden_dir, prefix, over):
    z = np.load(golden_dir / "f4_cornell_params.npz")
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 16, 16)
    params = pkg.default_params(width=16, height=16, samples_per_pixel=1, seed=7, **over)
    rad, words = ob.oracle_render_pass(scene.view(), cam, params, 0)
    assert np.array_equal(words, z[f"{prefix}_words_seed7"][0])
    assert close(rad, z[f"{prefix}_radiance_seed7"][0])


def test_survey_recorded_known_answers(pkg, ob):
    """Values SURVEY.md section 8c recorded from the unmodified reference (its own OBJ loader
    included): they pin this repository's loader + scene constants + oracle end to end."""
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 16, 16)
    params = pkg.default_params(width=16, height=16, samples_per_pixel=1, seed=1)
    rad, words = ob.oracle_render_pass(scene.view(), cam, params, 0)
    assert close(rad[0, 0], (0.0159814638671875, 0.012415037812499998, 0.010729992000000001), 1e-15)
    assert close(rad[0, 1], (0.057355542968750006, 0.016830008625, 0.0048317332000000004), 1e-15)
    assert close(rad[8, 8], (0.0072672789306640626, 0.0067291741887499992, 0.0057478169600000007), 1e-15)
    assert (words[0, 0], words[0, 1], words[8, 8]) == (374, 416, 440)
    assert close(rad.reshape(-1, 3).sum(0), (80.489992458842224, 53.051581188476284, 17.039326212370096), 1e-13)
    ray = ob.oracle_camera_ray(cam, 0, 0, 1)
    assert close(ray, (-0.0069242485471704461, 1.0072015954628901, 3, -0.35146831429374065,
                       0.35464433437652504, -0.86642796592800952), 1e-15)
    scene = pkg.Scene()
    cam = scene.build_named("suzanne", 16, 16)
    rad, words = ob.oracle_render_pass(scene.view(), cam, params, 0)
    assert close(rad[8, 8], (0.30874999999999997, 0.9850000000000001, 0.3725), 1e-15) and words[8, 8] == 152
    assert close(rad.reshape(-1, 3).sum(0), (17.395530914770262, 49.561542638784076, 23.069347823604009), 1e-13)
    scene = pkg.Scene()
    cam = scene.build_named("ce", 6, 4)
    rad, words = ob.oracle_render_pass(scene.view(), cam, params.__class__.from_buffer_copy(
        pkg.default_params(width=6, height=4, samples_per_pixel=1, seed=1)), 0)
    assert close(rad, np.broadcast_to((0.5675, 0.75, 0.7425), rad.shape), 1e-15)
    assert np.all(words == 488)


def test_perpixel_policy_is_tile_independent(pkg, ob):
    """PERPIXEL: a row window renders exactly the pixels the full frame renders there."""
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 12, 10)
    full = pkg.default_params(width=12, height=10, samples_per_pixel=2, seed=3, rng_policy=pkg.RNG_PERPIXEL)
    rgb, cnt, _, _ = ob.oracle_render(scene.view(), cam, full, threads=2)
    part = pkg.default_params(width=12, height=10, samples_per_pixel=2, seed=3, rng_policy=pkg.RNG_PERPIXEL,
                              row_begin=3, row_end=7)
    prgb, pcnt, _, _ = ob.oracle_render(scene.view(), cam, part, threads=2)
    assert np.array_equal(prgb[3:7], rgb[3:7]) and np.all(pcnt[3:7] == 2)
    assert not prgb[:3].any() and not prgb[7:].any() and not pcnt[:3].any() and not pcnt[7:].any()


def test_fast_oracle_build_stays_within_reference_flag_spread(pkg, ob):
    """The cpu_baseline build (reference's flags: FMA contraction, unsafe-math) differs from the
    strict build only in the last bits and consumes the same RNG words."""
    if ob.oracle_fast is None:
        pytest.skip("fast oracle build missing")
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 24, 24)
    params = pkg.default_params(width=24, height=24, samples_per_pixel=2, seed=1)
    a = ob.oracle_render(scene.view(), cam, params, threads=2)
    b = ob.oracle_render(scene.view(), cam, params, threads=2, lib=ob.oracle_fast)
    assert np.array_equal(a[2], b[2])
    assert close(b[0], a[0], 1e-12)
