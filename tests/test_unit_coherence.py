"""CPU: ptw_scene_unit_coherence - the host-side statistic that switches the worker waves' unit-level u-first early-out
on (csrc/ptw_trace_common.h testTriangleUnit; threshold 0.4, host/precompute.h).  It decides a schedule, never a
result: the GPU suite holds both forms to the oracle (tests/test_gpu_round6.py)."""
import numpy as np


def test_meshes_score_high_soups_low(pkg):
    scores = {}
    for name in ("ce", "suzanne", "cornell"):
        scene = pkg.Scene()
        scene.build_named(name, 32, 32)
        scores[name] = scene.unit_coherence()
    assert scores["ce"] >= 0.55, scores            # faces follow each other in space: most units fail u as a whole
    assert 0.05 <= scores["suzanne"] < 0.4, scores   # (measured: the early-out costs this scene 1 %)
    assert scores["cornell"] == 0.0                 # fewer than two units
    assert scores["ce"] == pkg_again(pkg, "ce")     # a fixed sample: the same number every time


def pkg_again(pkg, name):
    scene = pkg.Scene()
    scene.build_named(name, 8, 8)
    return scene.unit_coherence()


def test_soup_of_large_triangles_scores_low_small_triangles_high(pkg):
    """The statistic is the early-out's own success rate: a unit fails the u test as a whole when its triangles are
    far from the ray - because they sit together (a mesh) or because they are small; large triangles all over the
    scene (the soups of the GPU suite) almost never do."""
    rng = np.random.default_rng(5)
    m = pkg.material("diffuse", (0.5, 0.5, 0.5))
    ctr = rng.uniform(-3, 3, (2000, 3))
    scores = []
    for size in (1.5, 0.05):
        scene = pkg.Scene()
        for t in ctr[:, None, :] + rng.uniform(-size, size, (2000, 3, 3)):
            scene.add_triangle(t[0], t[1], t[2], m)
        scores.append(scene.unit_coherence())
    assert scores[0] < 0.2 < 0.4 < scores[1], scores


def test_debug_default_leaves_the_rule_to_the_library(pkg):
    assert pkg.debug_options().seq_unit_ufirst == -1
    assert pkg.debug_options(seq_unit_ufirst=1).seq_unit_ufirst == 1
