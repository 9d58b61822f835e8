"""CPU: the dispatcher's table (ptw_dispatch_plan: the rules of csrc/dispatch.hip and the family launchers, asked
without a device) - which kernel runs for a scene of N triangles and a launch of P passes.

* monotone in N: a larger scene never gets a kernel with FEWER resident triangles per workgroup, and the families
  follow each other in one order (speculative / single wave -> worker waves);
* the thresholds are where DESIGN.md says they are (64 / 128 triangles, more passes than CUs, 12 / 8 units for the
  shares by place with two masters / one, the LDS budget of the shading tables);
* every name the table can produce is a kernel the GPU suite holds to the oracle somewhere (the names are the ones
  ptw_kernel_stats.trace_kernel reports).
"""
import re

import pytest

SIZES = [1, 8, 32, 63, 64, 65, 96, 127, 128, 129, 256, 384, 385, 512, 640, 768, 970, 1000, 1024, 1343, 1344, 1345,
         1537, 2000, 2048, 2689, 3000, 3442, 3584, 4000, 4225, 5376, 5377, 8192, 24202, 32768, 100000]


def resident(name):
    """(family rank, triangles resident in one workgroup's registers) of a SEQUENTIAL kernel name."""
    if name.startswith("traceSequentialSpec"):
        return 0, 64
    m = re.match(r"traceSequential<(\d+),(\d+),(lds|global),(reg|stack)(,2 masters)?>$", name)
    assert m, name
    slots, waves = int(m.group(1)), int(m.group(2))
    return (0 if waves == 1 else 1), slots * waves * 64


@pytest.mark.parametrize("passes", [1, 100, 256, 257, 512, 1024, 4096])
@pytest.mark.parametrize("cus", [256, 64])
def test_sequential_table_is_monotone_in_the_scene_size(pkg, passes, cus):
    prev = (0, 0)
    for n in SIZES:
        name = pkg.dispatch_plan(n, samples_per_pixel=passes, compute_units=cus)
        rank = resident(name)
        assert rank >= prev, (n, passes, name, prev)
        # the kernel holds the scene, or it is the largest of its family (the rest is streamed)
        if rank[1] < n:
            assert name.startswith(("traceSequential<12,7,", "traceSequential<11,6,")), (n, name)
        prev = rank


def test_thresholds(pkg):
    plan = pkg.dispatch_plan
    # at most 64 triangles: the speculative kernel while a pass can have a CU to itself, one wave per pass beyond
    assert plan(64, samples_per_pixel=256) == "traceSequentialSpec"
    assert plan(64, samples_per_pixel=257) == "traceSequential<1,1,lds,reg>"
    assert plan(64, samples_per_pixel=65, compute_units=64) == "traceSequential<1,1,lds,reg>"
    assert plan(64, num_spheres=64) == "traceSequential<1,1,lds,stack>"          # 128 primitives: no byte-per-level stack
    assert plan(40, max_depth=10) == "traceSequential<1,1,lds,stack>"
    assert plan(65) == plan(128) == "traceSequential<2,1,lds,stack>"
    # beyond 128: worker waves; two passes per workgroup with more passes than CUs
    assert plan(129, samples_per_pixel=256) == "traceSequential<1,7,lds,stack>"
    assert plan(129, samples_per_pixel=257) == "traceSequential<1,6,lds,stack,2 masters>"
    assert plan(970, samples_per_pixel=512) == "traceSequential<3,6,lds,stack,2 masters>"     # BASELINE cfg3
    assert plan(3442, num_spheres=3, samples_per_pixel=1024) == "traceSequential<10,6,global,stack,2 masters>"  # cfg4
    # one master: shares by place from 8 units on (below 31 inside the equal shares' instantiation); ce 9 / 6 / 9
    assert plan(3442, samples_per_pixel=256) == "traceSequential<9,7,global,stack>"
    assert plan(970, samples_per_pixel=256) == "traceSequential<3,7,lds,stack>"
    assert plan(3700, samples_per_pixel=256) == "traceSequential<10,7,global,stack>"
    assert plan(4600, samples_per_pixel=256) == "traceSequential<12,7,global,stack>"
    # shares by place (two masters) from 12 units of 64 triangles on; below 31 units they stay inside the instantiation
    # the equal shares choose: 17 units -> 3 slots (4 / 2 / 4 would need four), 30 -> 6 / 4 / 6 in the 6-slot kernel
    assert plan(17 * 64, samples_per_pixel=512) == "traceSequential<3,6,lds,stack,2 masters>"
    assert plan(30 * 64, samples_per_pixel=512) == "traceSequential<6,6,global,stack,2 masters>"
    assert plan(31 * 64, samples_per_pixel=512) == "traceSequential<6,6,global,stack,2 masters>"
    # the shading tables leave LDS when they outgrow the budget (96 B per triangle + the two generators)
    assert ",lds," in plan(1200, samples_per_pixel=512) and ",global," in plan(1400, samples_per_pixel=512)
    # forced roads
    assert plan(970, samples_per_pixel=512, seq_two_masters=0) == "traceSequential<3,7,lds,stack>"
    assert plan(40, samples_per_pixel=1024, seq_small_kernel=2) == "traceSequentialSpec"
    assert plan(40, seq_small_kernel=3) == "traceSequentialSpec<no cross-pixel candidate>"
    assert plan(40, samples_per_pixel=512, seq_small_kernel=4) == "traceSequentialSpec<2 waves>"
    assert plan(200, samples_per_pixel=512, seq_lds_tables=0) == "traceSequential<1,6,global,stack,2 masters>"
    # the unit-level u-first form of the worker-wave kernels is the scene's decision (ptw_scene_unit_coherence): forced
    assert plan(3442, samples_per_pixel=1024, seq_unit_ufirst=1) == "traceSequential<10,6,global,stack,2 masters,unit>"   # cfg4
    assert plan(3442, samples_per_pixel=256, seq_unit_ufirst=1) == "traceSequential<9,7,global,stack,unit>"
    assert plan(100, samples_per_pixel=256, seq_unit_ufirst=1) == "traceSequential<2,1,lds,stack>"     # (worker waves only)
    assert plan(3442, samples_per_pixel=1024, accel=2, seq_unit_ufirst=1).endswith(",prefilter>")


def test_perpixel_table(pkg):
    plan = pkg.dispatch_plan
    for n in SIZES:
        assert plan(n, rng_policy=1) == "tracePerPixelPersistent"                   # (AUTO, nothing calibrated)
        assert plan(n, rng_policy=1, pix_kernel=1) == "tracePerPixel"
        assert plan(n, rng_policy=1, accel=1) == "tracePerPixelBvh"
        assert plan(n, rng_policy=1, accel=2) == "tracePerPixelPersistentPrefilter"
        assert plan(n, rng_policy=1, accel=2, pix_kernel=1) == "tracePerPixelPrefilter"
    with pytest.raises(pkg.PtwError):
        plan(100, rng_policy=0, accel=2)
    with pytest.raises(pkg.PtwError):
        plan(1000, rng_policy=0, accel=1)
    # the prefilter form of the worker-wave kernels (SEQUENTIAL, beyond 128 triangles)
    assert plan(970, samples_per_pixel=512, accel=2) == "traceSequential<3,6,lds,stack,2 masters,prefilter>"
    assert plan(3442, samples_per_pixel=1024, accel=2) == "traceSequential<10,6,global,stack,2 masters,prefilter>"
    assert plan(3442, samples_per_pixel=256, accel=2) == "traceSequential<8,7,global,stack,prefilter>"


def test_every_planned_kernel_is_tested_on_the_gpu(pkg):
    """The names the table can produce, against the names the GPU tests assert (so that no instantiation ships
    that the suite never ran)."""
    from conftest import ROOT
    produced = set()
    for passes in (1, 256, 512, 4096):
        for n in SIZES:
            for extra in ({}, {"seq_lds_tables": 0}, {"seq_two_masters": 1}, {"seq_two_masters": 0}):
                produced.add(pkg.dispatch_plan(n, samples_per_pixel=passes, **extra))
                if n > 128:
                    produced.add(pkg.dispatch_plan(n, samples_per_pixel=passes, accel=2, **extra))
                    # ... and its unit-level u-first form (picked by the scene's statistic): held to the oracle by
                    # test_worker_wave_kernels_with_the_unit_early_out_forced_on_match_oracle over the same case lists
                    unit = pkg.dispatch_plan(n, samples_per_pixel=passes, seq_unit_ufirst=1, **extra)
                    assert unit.replace(",unit>", ">") == pkg.dispatch_plan(n, samples_per_pixel=passes, **extra)
    text = "".join(p.read_text() for p in (ROOT / "tests").glob("test_gpu_*.py"))
    assert "UFIRST_CASES = [(n, t, k + \">\", 1) for n, t, k in __import__(\"test_gpu_round3\").TWO_MASTER_CASES]" in text
    missing = [k for k in sorted(produced) if k.rstrip(">") not in text and k not in text]
    assert not missing, missing
