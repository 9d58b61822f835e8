"""CPU: the parts of bench.py that need no GPU - the BASELINE presets, the metric strings, the
roofline arithmetic, and the choice of the parity reference (oracle/_ref when it is on the box,
the pinned restatement otherwise)."""
import importlib.util
import sys
import types
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _args(bench, monkeypatch, *argv):
    monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
    return bench.parse_args()


def test_default_is_the_configuration_the_metric_is_quoted_on(bench, monkeypatch):
    a = _args(bench, monkeypatch)
    assert (a.scene, a.width, a.height, a.spp, a.rows, a.policy) == ("cornell", 1024, 1024, 256, "", "sequential")
    assert a.gpus == 1 and a.steps == 1 and a.warmup == 0 and a.is_default_workload
    # BASELINE.json's metric string, verbatim
    import json
    baseline = json.loads((ROOT / "BASELINE.json").read_text())
    assert bench.metric_name(a.scene, a.width, a.height, a.spp) == baseline["metric"]


def test_config_presets_name_the_baseline_sizes(bench, monkeypatch):
    a = _args(bench, monkeypatch, "--config", "cfg3")
    assert (a.scene, a.width, a.height, a.spp, a.rows) == ("suzanne", 1024, 1024, 512, "") and not a.is_default_workload
    assert bench.metric_name(a.scene, a.width, a.height, a.spp).startswith("Msamples/sec suzanne 1024x1024@512spp")
    a = _args(bench, monkeypatch, "--config", "cfg4")
    assert (a.scene, a.width, a.height, a.spp) == ("ce", 2048, 2048, 1024)
    r0, r1 = (int(v) for v in a.rows.split(":"))
    assert r0 == 0 and 0 < r1 < 2048, "cfg4 is a stated PREFIX sub-run of the frame"
    a = _args(bench, monkeypatch, "--config", "cfg2")
    assert a.is_default_workload
    # a smaller frame is not the default workload: no side legs
    assert not _args(bench, monkeypatch, "--width", "64", "--height", "64").is_default_workload
    assert set(bench.SIDE_PARITY) == {"cfg3", "cfg4"} and set(bench.CONFIGS) == {"cfg2", "cfg3", "cfg4"}


def test_roofline_arithmetic(bench):
    stats = types.SimpleNamespace(trace_launches=2, trace_ms=4000.0, rays=10_000_000, samples=200_000,
                                  trace_kernel=b"traceSequentialSpec")
    r = bench.roofline_of(stats, ntri=38, nsph=1)
    flop_per_ray = 38 * 45.0 + 19.0
    assert r["kernel"] == "traceSequentialSpec" and r["launches"] == 2 and r["avg_launch_ms"] == 2000.0
    assert abs(r["achieved"] - (5_000_000 * flop_per_ray / 2.0 / 1e12)) < 1e-15
    assert abs(r["frac"] - r["achieved"] / 78.6) < 1e-15 and r["rays_per_sample"] == 50.0


def test_parity_reference_falls_back_to_the_pinned_restatement(bench):
    """A clean checkout has no oracle/_ref (git-ignored): the metric's second half must not vanish."""
    with_ref = types.SimpleNamespace(ref_fast=object(), SCENE_CAMERAS={"cornell": {}})
    without = types.SimpleNamespace(ref_fast=None, SCENE_CAMERAS={"cornell": {}})
    assert bench.reference_kind(with_ref, "cornell") == "reference"
    assert bench.reference_kind(with_ref, "a-scene-the-ref-driver-has-no-camera-for") == "port"
    assert bench.reference_kind(without, "cornell") == "port"


def test_ref_passes_runs_the_restatement_when_the_reference_is_absent(bench, pkg, ob, monkeypatch):
    """The same pass loop through either checker: with oracle/_ref hidden, ref_passes() drives the C
    restatement, in pass order, and its radiance equals the strict oracle's (and, where oracle/_ref
    exists, the reference's own, bit for bit - tests/test_oracle_vs_ref.py)."""
    import numpy as np
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 12, 8)
    params = pkg.default_params(width=12, height=8, samples_per_pixel=3, seed=1)
    seen = []
    hidden = types.SimpleNamespace(ref_fast=None, ref=None, SCENE_CAMERAS=ob.SCENE_CAMERAS, oracle=ob.oracle,
                                   oracle_fast=ob.oracle_fast, oracle_render_pass=ob.oracle_render_pass)
    bench.ref_passes(hidden, "cornell", scene.view(), cam, params, [0, 1, 2], 2, True,
                     lambda k, rad, words: seen.append((k, rad.copy(), words.copy())), strict=True)
    assert [k for k, _, _ in seen] == [0, 1, 2]
    for k, rad, words in seen:
        want_rad, want_words = ob.oracle_render_pass(scene.view(), cam, params, k)
        assert np.array_equal(rad, want_rad) and np.array_equal(words, want_words)
    # a prefix of the rows (the bounded parity windows of the large frames): same values on those rows
    prefix = pkg.default_params(width=12, height=8, samples_per_pixel=3, seed=1, row_begin=0, row_end=3)
    rad_p, words_p = ob.oracle_render_pass(scene.view(), cam, prefix, 1)
    assert np.array_equal(rad_p[:3], seen[1][1][:3]) and np.array_equal(words_p[:3], seen[1][2][:3])
    assert not rad_p[3:].any()
    if ob.HAVE_REF:
        desc = ob.cam_desc(**ob.SCENE_CAMERAS["cornell"])
        rs = ob.RefScene(scene.view())
        rad_r, words_r = rs.render_pass(desc, prefix, 1)
        assert np.array_equal(rad_r[:3], seen[1][1][:3]) and np.array_equal(words_r[:3], seen[1][2][:3])
        assert not rad_r[3:].any()


@pytest.mark.parametrize("env,taken", [
    ({}, True),                                    # nothing set
    ({"NCCL_DEBUG": "VERSION"}, True),             # what the GPU boxes export: no channel lines at that level
    ({"NCCL_DEBUG": "warn"}, True),
    ({"NCCL_DEBUG": "INFO"}, False),               # the caller's own choice of level ...
    ({"NCCL_DEBUG": "TRACE"}, False),
    ({"NCCL_DEBUG_FILE": "/tmp/mine.log"}, False),                       # ... or of a file is left alone
    ({"NCCL_DEBUG": "VERSION", "NCCL_DEBUG_FILE": "/tmp/mine.log"}, False),
])
def test_rccl_log_is_routed_to_a_file_unless_the_caller_chose(bench, env, taken):
    """`rccl_transport.rccl_log` needs RCCL's INFO channel lines in a file ptw_comm_describe can read
    (round 5: the first two-rank line said null because the box exports NCCL_DEBUG=VERSION)."""
    before = dict(env)
    assert bench.route_rccl_log(env) is taken
    if taken:
        assert env["NCCL_DEBUG"] == "INFO" and "%h" in env["NCCL_DEBUG_FILE"] and "%p" in env["NCCL_DEBUG_FILE"]
    else:
        assert env == before
