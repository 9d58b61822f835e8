"""GPU: round-2 additions of the boundary and full-size parity.

* the WHOLE 1024 x 1024 headline frame against the oracle (every pixel, every sample's RNG word
  count) - the largest frame compared so far was 1024 x 40;
* PERPIXEL policy against the reference's policy as a STATISTICAL test (per-pixel z-scores from
  batch variances, unbiasedness per channel, RMSE ~ 1/sqrt(spp));
* interleaved-row shards, empty shards, argument validation, the update callback that
  hands back the running framebuffer inside one render, the kernel-variant name in the stats,
  ptw_render_ex's multi-device decomposition (on one device) and the RCCL communicator.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))


def device_render(pkg, scene, cam, params, want_words=False, ctx=None):
    import torch
    ctx = ctx or pkg.Context(0)
    if ctx is not None and not getattr(ctx, "_has_scene", False):
        ctx.set_scene(scene)
        ctx._has_scene = True
    h, w, spp = params.height, params.width, params.samples_per_pixel
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    words = torch.zeros((max(spp, 1), h, w), dtype=torch.int32, device="cuda") if want_words else None
    ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), words.data_ptr() if want_words else 0,
               torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = (rgb.cpu().numpy(), cnt.cpu().numpy().astype(np.uint32))
    return out + ((words.cpu().numpy().astype(np.uint32),) if want_words else ())


def test_full_headline_frame_matches_oracle(pkg, ob):
    """cornell 1024 x 1024, the frame BASELINE.json's metric is quoted on, 2 passes: every pixel's
    fp64 sum and every sample's RNG word count."""
    w = h = 1024
    scene = pkg.Scene()
    cam = scene.build_named("cornell", w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=2, seed=1)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, params, threads=2)
    rgb, cnt, words = device_render(pkg, scene, cam, params, want_words=True)
    assert np.array_equal(cnt, ref_cnt)
    assert int(np.count_nonzero(words != ref_words)) == 0, "a path decision diverged somewhere in the frame"
    assert rel_err(rgb, ref_rgb) < TOL
    identical = np.all(rgb == ref_rgb, axis=2)
    assert identical.mean() > 0.999, f"only {identical.mean():.6f} of the pixels are bit-identical"
    mean = rgb / 2.0
    rmse = np.sqrt(np.mean((mean - ref_rgb / 2.0) ** 2, axis=(0, 1)))
    assert np.all(rmse < 1e-13)


@pytest.mark.parametrize("name,edge,batches,per_batch", [("cornell", 64, 16, 32), ("suzanne", 64, 16, 32)])
def test_perpixel_policy_is_the_same_estimator(pkg, name, edge, batches, per_batch):
    """PERPIXEL (independent stream per sample) against SEQUENTIAL (the reference's streams, which
    the other tests pin bit-for-bit to the reference): per-pixel z-scores of the difference of the
    two means, with variances estimated from `batches` independent batches of `per_batch` passes
    each.  Same estimator => z ~ Student t (mean 0, unit-ish variance): the mean z is ~0 within
    its standard error, the tails stay within generous bounds, every channel's image mean agrees
    within its standard error, and the RMSE between the policies falls like 1/sqrt(spp)."""
    scene = pkg.Scene()
    cam = scene.build_named(name, edge, edge)
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    ctx._has_scene = True

    def batch_means(policy):
        out = []
        for b in range(batches):
            p = pkg.default_params(width=edge, height=edge, samples_per_pixel=per_batch, seed=1,
                                   first_pass=b * per_batch, rng_policy=policy)
            rgb, cnt = device_render(pkg, scene, cam, p, ctx=ctx)
            assert np.all(cnt == per_batch)
            out.append(rgb / per_batch)
        return np.stack(out)  # [batches, h, w, 3]

    seq, pp = batch_means(0), batch_means(1)
    m_seq, m_pp = seq.mean(0), pp.mean(0)
    var = (seq.var(0, ddof=1) + pp.var(0, ddof=1)) / batches  # variance of the difference of means
    live = var > 0                                             # constant pixels (e.g. pure emitters) carry no noise
    z = (m_pp - m_seq)[live] / np.sqrt(var[live])
    assert live.mean() > 0.5
    n = z.size
    assert abs(z.mean()) < 5.0 / np.sqrt(n) * z.std() + 0.02, f"biased: mean z = {z.mean():.4f}"
    assert 0.8 < z.std() < 1.35, f"z spread {z.std():.3f}"
    assert np.mean(np.abs(z) > 3.0) < 0.02 and np.mean(np.abs(z) > 6.0) < 1e-3
    # where a pixel carries no noise in either policy the values agree exactly
    assert np.allclose(m_pp[~live], m_seq[~live], rtol=1e-12, atol=0)
    # per-channel image means within 5 standard errors
    for c in range(3):
        d = (m_pp[..., c] - m_seq[..., c]).mean()
        se = np.sqrt(var[..., c].sum()) / (edge * edge)
        assert abs(d) < 5 * se + 1e-15, (c, d, se)
    # RMSE between the two policies ~ 1/sqrt(spp): a quarter of the batches vs all of them
    q = batches // 4
    rmse_q = np.sqrt(np.mean((pp[:q].mean(0) - seq[:q].mean(0)) ** 2))
    rmse_all = np.sqrt(np.mean((m_pp - m_seq) ** 2))
    assert 1.5 < rmse_q / rmse_all < 2.7, (rmse_q, rmse_all)


@pytest.mark.parametrize("name,w,h", [("cornell", 20, 13), ("suzanne", 12, 11)])
def test_interleaved_rows_perpixel(pkg, ob, name, w, h):
    """row_stride / row_phase: both PERPIXEL kernels, against the oracle's shard and against the
    full frame (disjoint supports: the union of the shards IS the frame, bit for bit)."""
    scene = pkg.Scene()
    cam = scene.build_named(name, w, h)
    full, _ = device_render(pkg, scene, cam, pkg.default_params(width=w, height=h, samples_per_pixel=3, seed=4,
                                                                rng_policy=1))
    acc = np.zeros_like(full)
    for phase in range(3):
        p = pkg.default_params(width=w, height=h, samples_per_pixel=3, seed=4, rng_policy=1, row_stride=3,
                               row_phase=phase)
        rgb, cnt, words = device_render(pkg, scene, cam, p, want_words=True)
        ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, p, threads=2)
        assert np.array_equal(cnt, ref_cnt) and rel_err(rgb, ref_rgb) < TOL
        own = np.arange(h) % 3 == phase
        assert np.all(cnt[own] == 3) and not cnt[~own].any() and not rgb[~own].any()
        assert np.array_equal(words[:, own], ref_words[:, own])
        acc += rgb
    assert np.array_equal(acc, full)
    # a window combined with a stride
    p = pkg.default_params(width=w, height=h, samples_per_pixel=2, seed=4, rng_policy=1, row_begin=2, row_end=9,
                           row_stride=2, row_phase=1)
    rgb, cnt = device_render(pkg, scene, cam, p)
    assert [int(r[0]) for r in cnt] == [2 if (2 <= y < 9 and y % 2 == 1) else 0 for y in range(h)]


def test_empty_shard_and_argument_errors(pkg):
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 10, 8)
    # begin == end != 0: an empty shard adds nothing (it used to fall back to the whole frame)
    rgb, cnt = device_render(pkg, scene, cam, pkg.default_params(width=10, height=8, samples_per_pixel=2, seed=1,
                                                                 rng_policy=1, row_begin=8, row_end=8))
    assert not cnt.any() and not rgb.any()
    def status(**over):
        with pytest.raises(pkg.PtwError) as e:
            pkg.render(scene, cam, pkg.default_params(width=10, height=8, samples_per_pixel=1, seed=1, **over))
        return e.value.status
    assert status(first_bounce_u=0) == 1 and status(first_bounce_v=0) == 1   # PTW_ERR_INVALID
    assert status(first_bounce_u=1 << 11, first_bounce_v=1 << 11) == 1
    assert status(row_begin=2, row_end=4) == 8                               # SEQUENTIAL + window: UNSUPPORTED
    # ... except a prefix of the frame: exactly the rows the full render produces
    full, _ = pkg.render(scene, cam, pkg.default_params(width=10, height=8, samples_per_pixel=3, seed=1))
    part, pcnt = pkg.render(scene, cam, pkg.default_params(width=10, height=8, samples_per_pixel=3, seed=1, row_end=5))
    assert np.array_equal(part[:5], full[:5]) and not part[5:].any() and np.all(pcnt[:5] == 3) and not pcnt[5:].any()
    assert status(row_stride=2, row_phase=0) == 8
    assert status(rng_policy=1, row_stride=2, row_phase=2) == 1


def test_update_callback_hands_back_the_running_framebuffer(pkg):
    """updateFunc(output), src/dod/Scene.cpp:245: one render, one context, the caller's buffers
    valid at every call (finished rows complete, the others untouched), final result unchanged."""
    w, h, spp = 32, 24, 4
    scene = pkg.Scene()
    cam = scene.build_named("cornell", w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=2)
    plain, plain_cnt = pkg.render(scene, cam, params)
    seen = []

    def update(done, total, rgb, cnt):
        assert total == w * h * spp and 0 < done <= total
        complete = cnt == spp
        assert np.all((cnt == 0) | complete)
        assert np.array_equal(rgb[complete], plain[complete]) and not rgb[~complete].any()
        assert int(complete.sum()) * spp >= done - w * spp  # whole rows are copied: at most one row ahead
        seen.append(done)
        return False

    rgb, cnt = pkg.render(scene, cam, params, update=update, min_updates=6)
    assert len(seen) >= 6 and seen == sorted(seen) and seen[-1] == w * h * spp
    assert np.array_equal(rgb, plain) and np.array_equal(cnt, plain_cnt)
    # cancelling from the callback stops the render and reports it
    with pytest.raises(pkg.PtwError):
        pkg.render(scene, cam, params, update=lambda *a: True, min_updates=6)
    # accumulation into non-empty buffers is preserved (ArrayOutput::operator+=)
    rgb2, cnt2 = pkg.render(scene, cam, params, rgb_sum=plain.copy(), counts=plain_cnt.copy(),
                            update=lambda *a: False)
    assert np.array_equal(cnt2, 2 * plain_cnt) and rel_err(rgb2, plain + plain) < 1e-14  # (x + p0) + p1 ...


def test_kernel_variant_is_reported_by_the_library(pkg):
    import torch
    cases = [("cornell", 0, "traceSequentialSpec"), ("cornell", 1, "tracePerPixelPersistent"),
             ("suzanne", 0, "traceSequential<3,7,lds,stack>"), ("suzanne", 1, "tracePerPixelPersistent")]
    for name, policy, want in cases:
        scene = pkg.Scene()
        cam = scene.build_named(name, 8, 8)
        ctx = pkg.Context(0)
        ctx.set_scene(scene)
        ctx.enable_stats(True)
        rgb = torch.zeros((8, 8, 3), dtype=torch.float64, device="cuda")
        cnt = torch.zeros((8, 8), dtype=torch.int32, device="cuda")
        ctx.render(cam, pkg.default_params(width=8, height=8, samples_per_pixel=2, seed=1, rng_policy=policy),
                   rgb.data_ptr(), cnt.data_ptr())
        torch.cuda.synchronize()
        st = ctx.stats(reset=True)
        assert st.trace_kernel.decode() == want and st.trace_launches == 1 and st.rays > 0


def test_render_ex_multi_device_decomposition_on_one_device(pkg):
    """ptw_render_ex(num_devices = N, share_device): the shards the N-GPU render would run (pass
    ranges / interleaved rows), one after another on this box's GPU."""
    w, h, spp = 24, 16, 7
    scene = pkg.Scene()
    cam = scene.build_named("cornell", w, h)
    for policy in (0, 1):
        params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=3, rng_policy=policy)
        one, one_cnt = pkg.render(scene, cam, params)
        for n in (2, 3):
            rgb, cnt = pkg.render(scene, cam, params, num_devices=n, share_device=True)
            assert np.array_equal(cnt, one_cnt)
            if policy == 1:
                assert np.array_equal(rgb, one)          # disjoint rows
            else:
                assert rel_err(rgb, one) < 1e-14         # pass ranges: order of the fp64 additions
    # more devices than the box has, without sharing: a clear error, not a hang
    import torch
    if torch.cuda.device_count() == 1:
        with pytest.raises(pkg.PtwError) as e:
            pkg.render(scene, cam, params, num_devices=2)
        assert "device" in e.value.message.lower()


def test_comm_single_rank_collectives(pkg):
    """ptw_comm_*: RCCL behind the C ABI.  A one-rank communicator on this box: the reduce and the
    gather are identities and must leave the buffers as they are."""
    import torch
    uid = pkg.Comm.unique_id()
    assert len(uid) == pkg.COMM_ID_BYTES
    comm = pkg.Comm.create(uid, 1, 0, 0)
    rgb = torch.arange(6 * 5 * 3, dtype=torch.float64, device="cuda").reshape(6, 5, 3)
    cnt = torch.arange(6 * 5, dtype=torch.int32, device="cuda").reshape(6, 5)
    keep_rgb, keep_cnt = rgb.clone(), cnt.clone()
    stream = torch.cuda.current_stream().cuda_stream
    comm.reduce_framebuffer(rgb.data_ptr(), cnt.data_ptr(), 30, 0, stream)
    comm.gather_rows(rgb.data_ptr(), cnt.data_ptr(), 5, 6, 0, stream)
    torch.cuda.synchronize()
    assert torch.equal(rgb, keep_rgb) and torch.equal(cnt, keep_cnt)
    comm.close()
    with pytest.raises(pkg.PtwError):
        pkg.Comm.create(uid, 2, 5, 0)   # rank out of range
