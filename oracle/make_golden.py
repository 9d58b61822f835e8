#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE build (oracle/_ref/libptw_ref.so).

TEST INFRASTRUCTURE.  Run in the container where /root/reference exists:

    make -C oracle ref && python oracle/make_golden.py

Every vector below is produced by the reference's own compiled code (dod::Scene, Camera,
ArrayOutput, libstdc++ <random>) driven by oracle/ref_driver.cpp; scenes are fed to it through
its own addTriangle/addSphere from this repository's loader (the reference's ObjLoader needs the
un-vendored CTRE header and is not buildable here).  The fixtures are data only: inputs and
expected outputs.  The strict (-ffp-contract=off) reference build is used so the numbers are
reproducible by any IEEE-754 fp64 implementation.
"""
import hashlib
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import oracle_binding as ob  # noqa: E402

pkg = ob.pkg
OUT = ROOT / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)
assert ob.HAVE_REF, "build oracle/_ref first (needs /root/reference)"


def save(name, **arrays):
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"wrote {path.relative_to(ROOT)} ({path.stat().st_size} bytes)")


# F1: RNG known answers straight from libstdc++ (std::mt19937, uniform_real_distribution).
f1 = {}
for seed in (1, 2, 5489, 0xFFFFFFFF):
    f1[f"words_{seed}"] = ob.ref_mt_words(seed, 1400)      # crosses two regenerations
    f1[f"unit_{seed}"] = ob.ref_unit_doubles(seed, 700)
save("f1_rng", **f1)

# F2: intersection known answers.  Inputs restate test/dod/{Sphere,Triangle,Scene}Tests.cpp as
# data; expected values are what the reference build returns for them (the Catch2 assertions of
# those tests - distance ~ 22.416738 etc. - are re-checked in tests/test_oracle_golden.py).
INF = float("inf")
cases = [
    # (name, spheres[(c, r)], triangles[(v0, v1, v2)], p1, p2, which, nearer_than)
    ("sphere_miss_up", [((10, 20, 30), 15)], [], (0, 0, 0), (0, 1, 0), "spheres", INF),
    ("sphere_miss_behind", [((10, 20, 30), 15)], [], (0, 0, 0), (-10, -20, -30), "spheres", INF),
    ("sphere_hit", [((10, 20, 30), 15)], [], (0, 0, 0), (10, 20, 30), "spheres", INF),
    ("sphere_hit_limited", [((10, 20, 30), 15)], [], (0, 0, 0), (10, 20, 30), "spheres", 22.0),
    ("sphere_known_point", [((0, 0, 30), 10)], [], (0, 0, 0), (0, 0, 2), "spheres", INF),
    ("sphere_from_inside", [((0, 0, 30), 10)], [], (0, 0, 30), (0, 0, 2), "spheres", INF),
    ("two_spheres_first_nearer", [((0, 0, 30), 10), ((0, 0, 90), 10)], [], (0, 0, 0), (0, 0, 2), "all", INF),
    ("two_spheres_second_nearer", [((0, 0, 90), 10), ((0, 0, 30), 10)], [], (0, 0, 0), (0, 0, 2), "all", INF),
    ("tri_cw_miss_up", [], [((0, 0, 3), (0, 1, 3), (1, 1, 3))], (0, 0, 0), (0, 1, 0), "triangles", INF),
    ("tri_cw_miss_behind", [], [((0, 0, 3), (0, 1, 3), (1, 1, 3))], (0, 0, 0), (0, 0, -1), "triangles", INF),
    ("tri_cw_hit", [], [((0, 0, 3), (0, 1, 3), (1, 1, 3))], (0, 0, 0), (0, 0, 1), "triangles", INF),
    ("tri_cw_hit_limited", [], [((0, 0, 3), (0, 1, 3), (1, 1, 3))], (0, 0, 0), (0, 0, 1), "triangles", 2.999),
    ("tri_ccw_hit", [], [((0, 0, 3), (1, 1, 3), (0, 1, 3))], (0, 0, 0), (0, 0, 1), "triangles", INF),
    ("mixed_triangle_in_front", [((0, 0, 30), 10)], [((-5, -5, 3), (5, -5, 3), (0, 5, 3))], (0, 0, 0), (0, 0, 1), "all", INF),
    ("mixed_sphere_in_front", [((0, 0, 30), 10)], [((-50, -50, 60), (50, -50, 60), (0, 50, 60))], (0, 0, 0), (0, 0, 1), "all", INF),
]
f2 = {"names": np.array([c[0] for c in cases])}
mat_a = pkg.material("diffuse", (1, 1, 1))
mat_b = pkg.material("diffuse", (1, 0, 0))
for name, spheres, tris, p1, p2, which, limit in cases:
    rs = ob.RefScene()
    for i, (c, r) in enumerate(spheres):
        rs.add_sphere(c, r, mat_a if i == 0 else mat_b)
    for (v0, v1, v2) in tris:
        rs.add_triangle(v0, v1, v2, mat_a)
    f2[f"{name}__spheres"] = np.array([list(c) + [r] for c, r in spheres], dtype=np.float64).reshape(-1, 4)
    f2[f"{name}__tris"] = np.array(tris, dtype=np.float64).reshape(-1, 3, 3)
    f2[f"{name}__p1p2"] = np.array([p1, p2], dtype=np.float64)
    f2[f"{name}__which"] = np.array(which)
    f2[f"{name}__limit"] = np.array(limit)
    f2[f"{name}__ray"] = ob.ref_ray_from_two_points(p1, p2)
    f2[f"{name}__hit"] = rs.intersect(p1, p2, which, limit)
save("f2_intersect", **f2)

# F3: scene dumps.  Full arrays for Cornell, sha256 of the arrays for the big ones.  These pin
# the loader + scene catalogue; that the reference renders them to the expected radiance is
# pinned by F4 (the reference consumes exactly these arrays).
f3 = {}
for name in ("cornell", "suzanne", "ce", "single-sphere", "multi-sphere", "example1", "bbc-owl"):
    scene = pkg.Scene()
    cam = scene.build_named(name, 64, 48)
    arr = scene.arrays()
    blob = b"".join(np.ascontiguousarray(arr[k]).tobytes() for k in
                    ("tri_vertices", "tri_material", "sph_centre_radius", "sph_material", "materials", "environment"))
    f3[f"{name}__sha256"] = np.array(hashlib.sha256(blob).hexdigest())
    f3[f"{name}__counts"] = np.array([arr["tri_vertices"].shape[0], arr["sph_centre_radius"].shape[0], arr["materials"].shape[0]])
    if name == "cornell":
        for k, v in arr.items():
            f3[f"cornell__{k}"] = v
save("f3_scenes", **f3)

# F8: camera known answers: 16 primary rays per scene from the reference's Camera.
f8 = {}
for name, d in ob.SCENE_CAMERAS.items():
    desc = ob.cam_desc(**d)
    w, h = 64, 48
    rays = []
    for i in range(16):
        px, py, seed = (i * 5) % w, (i * 7) % h, 100 + i
        rays.append(np.concatenate([[px, py, seed], ob.ref_camera_ray(desc, w, h, px, py, seed)]))
    f8[name] = np.array(rays)
save("f8_camera", **f8)

# F4: per-pass per-pixel radiance + RNG word counts from the reference's radiance().
def render_fixture(name, w, h, seeds, passes, **over):
    scene = pkg.Scene()
    scene.build_named(name, w, h)
    rs = ob.RefScene(scene.view())
    desc = ob.cam_desc(**ob.SCENE_CAMERAS[name])
    out = {}
    for seed in seeds:
        params = pkg.default_params(width=w, height=h, samples_per_pixel=passes, seed=seed, **over)
        rads, words = [], []
        for k in range(passes):
            r, wd = rs.render_pass(desc, params, k)
            rads.append(r)
            words.append(wd)
        out[f"radiance_seed{seed}"] = np.array(rads)
        out[f"words_seed{seed}"] = np.array(words).astype(np.uint16 if np.max(words) < 65536 else np.uint32)
    out["meta"] = np.array([w, h, passes] + list(seeds))
    return out

save("f4_cornell_32x32", **render_fixture("cornell", 32, 32, (1, 2, 3, 4), 2))
save("f4_suzanne_32x32", **render_fixture("suzanne", 32, 32, (1, 2), 2))
save("f4_ce_8x8", **render_fixture("ce", 8, 8, (1,), 1))
save("f4_example1_24x16", **render_fixture("example1", 24, 16, (1,), 2))
save("f4_bbc_owl_24x16", **render_fixture("bbc-owl", 24, 16, (1,), 1))
save("f4_multi_sphere_24x16", **render_fixture("multi-sphere", 24, 16, (1,), 2))
save("f4_single_sphere_24x16", **render_fixture("single-sphere", 24, 16, (1,), 2))
# non-default parameters: odd first-bounce fan-out (true division), deeper recursion, preview
save("f4_cornell_params", **{
    **{f"fb3x2_{k}": v for k, v in render_fixture("cornell", 16, 16, (7,), 1, first_bounce_u=3, first_bounce_v=2).items()},
    **{f"depth7_{k}": v for k, v in render_fixture("cornell", 16, 16, (7,), 1, max_depth=7).items()},
    **{f"depth1_{k}": v for k, v in render_fixture("cornell", 16, 16, (7,), 1, max_depth=1).items()},
    **{f"preview_{k}": v for k, v in render_fixture("cornell", 16, 16, (7,), 1, preview=1).items()},
})

# F6/F7: the .raw byte format and the 8-bit conversion, written by the reference's ArrayOutput.
scene = pkg.Scene()
scene.build_named("cornell", 16, 16)
rs = ob.RefScene(scene.view())
desc = ob.cam_desc(**ob.SCENE_CAMERAS["cornell"])
params = pkg.default_params(width=16, height=16, samples_per_pixel=15, seed=1)
rgb, counts = rs.render(desc, params, threads=1)
with tempfile.TemporaryDirectory() as tmp:
    path = str(Path(tmp) / "ref.raw")
    assert ob.ref.ref_raw_save(path.encode(), 16, 16, rgb.ctypes.data, counts.ctypes.data) == 0
    raw_bytes = np.frombuffer(Path(path).read_bytes(), dtype=np.uint8)
rgb8 = np.zeros((16, 16, 3), dtype=np.uint8)
ob.ref.ref_pixels_rgb8(16, 16, rgb.ctypes.data, counts.ctypes.data, rgb8.ctypes.data)
# a second buffer with hand-picked values around the clamp/rounding edges of componentToInt
edge = np.zeros((1, 16, 3))
edge[0, :, 0] = [-1.0, 0.0, 1e-9, 0.001, 0.0031308, 0.01, 0.2, 0.21404, 0.5, 0.73, 0.9999, 1.0, 1.5, 2.0, 0.25, 0.75]
edge[0, :, 1] = edge[0, ::-1, 0]
edge[0, :, 2] = 0.5 * edge[0, :, 0]
edge_counts = np.array([[1, 1, 1, 2, 3, 1, 7, 1, 2, 1, 1, 1, 1, 4, 0, 0]], dtype=np.uint32)
edge_sum = edge * np.maximum(edge_counts, 1)[..., None]
edge8 = np.zeros((1, 16, 3), dtype=np.uint8)
ob.ref.ref_pixels_rgb8(16, 1, edge_sum.ctypes.data, edge_counts.ctypes.data, edge8.ctypes.data)
save("f6_raw_png", rgb_sum=rgb, counts=counts, raw_bytes=raw_bytes, rgb8=rgb8,
     edge_sum=edge_sum, edge_counts=edge_counts, edge_rgb8=edge8)
