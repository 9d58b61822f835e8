"""CPU, only where oracle/_ref exists (the container with /root/reference): the C restatement
against the reference's own compiled sources on fresh seeded inputs, bit for bit."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def need_ref(ob):
    if not ob.HAVE_REF:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")


def test_libstdcxx_random(ob, need_ref):
    for seed in (0, 1, 7, 2 ** 31, 2 ** 32 - 1):
        assert np.array_equal(ob.mt_words(seed, 3000), ob.ref_mt_words(seed, 3000))
        assert np.array_equal(ob.mt_unit_doubles(seed, 1500), ob.ref_unit_doubles(seed, 1500))


@pytest.mark.parametrize("name,w,h,seed,over", [
    ("cornell", 20, 14, 11, {}), ("suzanne", 20, 20, 12, {}), ("ce", 5, 5, 13, {}),
    ("multi-sphere", 16, 12, 14, {}), ("example1", 16, 12, 15, {}), ("bbc-owl", 16, 12, 16, {}),
    ("cornell", 10, 10, 17, dict(first_bounce_u=5, first_bounce_v=1, max_depth=3)),
    ("cornell", 10, 10, -3, dict(first_bounce_u=1, first_bounce_v=1, max_depth=9)),
])
def test_pass_matches_reference_bitwise(pkg, ob, need_ref, name, w, h, seed, over):
    scene = pkg.Scene()
    cam = scene.build_named(name, w, h)
    view = scene.view()
    rs = ob.RefScene(view)
    desc = ob.cam_desc(**ob.SCENE_CAMERAS[name])
    params = pkg.default_params(width=w, height=h, samples_per_pixel=2, seed=seed, **over)
    for k in range(2):
        rad, words = ob.oracle_render_pass(view, cam, params, k)
        rrad, rwords = rs.render_pass(desc, params, k)
        assert np.array_equal(words, rwords)
        assert np.array_equal(rad, rrad)


def test_accumulated_render_matches_reference(pkg, ob, need_ref):
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 12, 12)
    params = pkg.default_params(width=12, height=12, samples_per_pixel=5, seed=21)
    rgb, cnt, _, _ = ob.oracle_render(scene.view(), cam, params, threads=3)
    rs = ob.RefScene(scene.view())
    rrgb, rcnt = rs.render(ob.cam_desc(**ob.SCENE_CAMERAS["cornell"]), params, threads=2)
    assert np.array_equal(cnt, rcnt) and np.array_equal(rgb, rrgb)


def test_random_rays_against_random_soup(pkg, ob, need_ref):
    rng = np.random.default_rng(5)
    scene = pkg.Scene()
    rs = ob.RefScene()
    mats = [pkg.material("diffuse", rng.random(3)) for _ in range(4)]
    for i in range(40):
        v = rng.uniform(-2, 2, (3, 3))
        scene.add_triangle(v[0], v[1], v[2], mats[i % 4])
        rs.add_triangle(v[0], v[1], v[2], mats[i % 4])
    for i in range(6):
        c, r = rng.uniform(-2, 2, 3), rng.uniform(0.1, 0.8)
        scene.add_sphere(c, r, mats[i % 4])
        rs.add_sphere(c, r, mats[i % 4])
    view = scene.view()
    marr = scene.arrays()["materials"]
    hits = 0
    for _ in range(400):
        p1, p2 = rng.uniform(-3, 3, 3), rng.uniform(-1, 1, 3)
        ray = ob.ref_ray_from_two_points(p1, p2)
        want = rs.intersect(p1, p2)
        got = ob.oracle_intersect(view, ray)
        assert np.array_equal(got[:8], want[:8])
        if want[0] >= 0:
            hits += 1
            assert np.array_equal(marr[int(got[8])], want[8:17])
    assert hits > 50
