"""CPU: the per-sample pick checksum of the oracle (oracle/ptw_oracle.c; include/ptw.h,
ptw_debug_options.d_picks) - what it is, that it can tell primitives apart, and what it cannot show
on BASELINE cfg4's scene.

pick(sample) = sum over the sample's intersect() calls r = 0, 1, ... (the reference's call order) of
(r + 1) * (combined primitive index + 1) mod 2^32, misses counting 0; combined index = position in
Scene::intersect's scan (src/dod/Scene.cpp:115-122): spheres first, then triangles."""
import numpy as np

import pick_helpers


def _coloured_scene(pkg):
    """Three spheres and five triangles, every primitive with its own diffuse colour."""
    scene = pkg.Scene()
    colour = lambda i: (0.1 + 0.1 * i, 0.9 - 0.1 * i, 0.5)  # noqa: E731
    k = 0
    for c, r in [((-1.2, 0, 0), 0.5), ((0, 0.2, 0.3), 0.4), ((1.3, -0.1, 0), 0.45)]:
        scene.add_sphere(c, r, pkg.material("diffuse", colour(k)))
        k += 1
    rng = np.random.default_rng(4)
    for _ in range(5):
        c = rng.uniform(-1.5, 1.5, 3) * (1, 1, 0.2)
        v = c + rng.uniform(-0.7, 0.7, (3, 3))
        scene.add_triangle(v[0], v[1], v[2], pkg.material("diffuse", colour(k)))
        k += 1
    scene.set_environment_colour((0, 0, 0))
    cam = pkg.look_at((0, 0, 5), (0, 0, 0), (0, 1, 0), 24, 18, 40.0)
    return scene, cam, [colour(i) for i in range(k)]


def test_the_combined_index_is_the_scan_order_of_scene_intersect(pkg, ob):
    """Preview mode returns the diffuse colour of the primary hit (Scene.cpp:137-138): with one colour per
    primitive the pick of a one-call sample can be checked against the colour the sample returned."""
    scene, cam, colours = _coloured_scene(pkg)
    params = pkg.default_params(width=24, height=18, samples_per_pixel=1, seed=3, preview=1)
    rad, _, picks = ob.oracle_render_pass_picks(scene.view(), cam, params, 0)
    seen = set()
    for y in range(18):
        for x in range(24):
            if picks[y, x] == 0:
                assert np.all(rad[y, x] == 0)                      # a miss: the (black) environment
                continue
            prim = int(picks[y, x]) - 1                            # call 0 has weight 1
            assert np.allclose(rad[y, x], colours[prim], atol=0, rtol=0), (x, y, prim)
            seen.add(prim)
    assert len([p for p in seen if p < 3]) >= 2 and len([p for p in seen if p >= 3]) >= 3   # spheres [0, 3), triangles from 3


def test_the_checksum_weights_calls_by_their_ordinal(pkg, ob):
    """A closed one-sphere scene: every call hits primitive 0, a sample makes 1 + 16 x 4 calls at depth 5:
    sum (r + 1) = 65 * 66 / 2."""
    scene = pkg.Scene()
    scene.add_sphere((0, 0, 0), 5.0, pkg.material("diffuse", (0.5, 0.5, 0.5)))
    cam = pkg.look_at((0, 0, 1), (0, 0, 0), (0, 1, 0), 4, 3, 40.0)
    params = pkg.default_params(width=4, height=3, samples_per_pixel=2, seed=1)
    _, _, words, picks = ob.oracle_render_picks(scene.view(), cam, params, threads=2)
    assert np.all(picks == 65 * 66 // 2) and np.all(words == 2 * (2 + 16 * 15))   # pinhole camera: 2 draws


def test_dropping_a_unit_of_triangles_changes_the_picks_of_a_soup(pkg, ob):
    rng = np.random.default_rng(11)
    scene = pkg.Scene()
    mat = [pkg.material("diffuse", (0.4, 0.5, 0.6)), pkg.material("light", (1, 1, 1))]
    for i in range(300):
        c = rng.uniform(-3, 3, 3)
        v = c + rng.uniform(-0.8, 0.8, (3, 3))
        scene.add_triangle(v[0], v[1], v[2], mat[i % 2])
    scene.add_sphere((0, 0, 0), 12.0, mat[0])
    cam = pkg.look_at((0, 0.5, 7), (0, 0, 0), (0, 1, 0), 6, 4, 45.0)
    params = pkg.default_params(width=6, height=4, samples_per_pixel=2, seed=5)
    unit = pick_helpers.find_sensitive_unit(pkg, ob, scene, cam, params)
    assert 0 <= unit < 5


def test_ce_as_shipped_never_hits_a_triangle(pkg, ob):
    """BASELINE cfg4's scene: the camera (0.27, 1.15, 0.36) sits inside the light sphere c = (0, 1.6, 0),
    r = 1 and inside the dull light c = (-0.2, 5.9, -0.3), r = 5 (src/main/main.cpp:122-129); the mesh lies
    in the slab y in [-0.18, 0.02], outside both.  Every ray starts inside a sphere it cannot leave without
    hitting it: the 3442 triangles are tested 65 times per sample and never win.  The frame WITHOUT the
    mesh is the same frame - radiance, word counts and picks.  (So on ce no comparison of outputs can
    tell whether all triangles were searched; the instantiation cfg4 runs is held to triangle soups for
    that, tests/test_gpu_round3.py.)"""
    import ctypes as C
    scene = pkg.Scene()
    cam = scene.build_named("ce", 16, 16)
    params = pkg.default_params(width=16, height=16, samples_per_pixel=2, seed=1)
    rgb, cnt, words, picks = ob.oracle_render_picks(scene.view(), cam, params, threads=4)
    arr = scene.arrays()
    bare = pkg.Scene()
    for (cx, cy, cz, r), m in zip(arr["sph_centre_radius"], arr["sph_material"]):
        mat = pkg.Material()
        C.memmove(C.byref(mat), np.ascontiguousarray(arr["materials"][int(m)], np.float64).ctypes.data, C.sizeof(mat))
        bare.add_sphere((cx, cy, cz), r, mat)
    bare.set_environment_colour(arr["environment"])
    rgb2, cnt2, words2, picks2 = ob.oracle_render_picks(bare.view(), cam, params, threads=4)
    assert np.array_equal(rgb, rgb2) and np.array_equal(words, words2) and np.array_equal(picks, picks2)
    assert np.all(words == 488) and np.all(picks < 65 * 66 // 2 * 4)     # only spheres 0..2 are ever picked
