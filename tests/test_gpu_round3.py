"""GPU: round-3 additions.

* the two-master worker-wave kernels - the ones BASELINE cfg3 (suzanne, 512 spp) and cfg4 (ce, 1024
  spp) actually run, `traceSequential<SLOTS, 6, lds|global, stack, 2 masters>` - compared DIRECTLY
  with the oracle (fp64 sums to 1e-12, every sample's RNG word count exact) on triangle soups sized
  to reach all eleven instantiations, with odd and even pass counts, parked streams, and once through
  the natural dispatch (more passes than CUs);
* the u-first early-out of the PERPIXEL triangle loop against the oracle on the same soups
  (test_kernel_variants_on_random_soups covers it too; here with many passes per pixel);
* the reference-side binding (integration/hip/Scene.h) EXECUTED on the GPU through a C++ host;
* ptw_render_ex(num_devices > 1): a failing shard is an error, not a hang.
"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))


def _soup(pkg, ntri, nsph, seed, w=4, h=3):
    """Random triangle / sphere soup inside an enclosing shell (long paths), all five material kinds."""
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
    scene.add_sphere((0, 0, 0), 12.0, mats[0])
    scene.set_environment_colour((0.1, 0.2, 0.3))
    cam = pkg.set_focus(pkg.look_at((0, 0.5, 7), (0, 0, 0), (0, 1, 0), w, h, 45.0), (0, 0, 0), 0.02)
    return scene, cam


def _render_with_stats(pkg, scene, cam, params, picks=False, **debug):
    """Device-resident render with per-sample RNG word counts; `debug`: ptw_debug_options fields (the
    dispatch forced for the test); `picks`: also the per-sample pick checksum (appended to the result)."""
    import torch
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    ctx.enable_stats(True)
    h, w, spp = params.height, params.width, params.samples_per_pixel
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    words = torch.zeros((spp, h, w), dtype=torch.int32, device="cuda")
    pk = torch.zeros((spp, h, w), dtype=torch.int32, device="cuda") if picks else None
    if debug or picks:
        ctx.set_debug(pkg.debug_options(d_picks=pk.data_ptr() if picks else 0, **debug))
    ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), words.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = ctx.stats(reset=True)
    out = (rgb.cpu().numpy(), cnt.cpu().numpy().astype(np.uint32), words.cpu().numpy().astype(np.uint32),
           st.trace_kernel.decode(), st.trace_launches)
    return out + (pk.cpu().numpy().astype(np.uint32),) if picks else out


# (triangles, shading tables, the instantiation that must run) - dispatch.hip dispatchSequential, seq_worker2.hip
TWO_MASTER_CASES = [
    (200, "lds", "traceSequential<1,6,lds,stack,2 masters"),
    (200, "global", "traceSequential<1,6,global,stack,2 masters"),
    (500, "lds", "traceSequential<2,6,lds,stack,2 masters"),
    (500, "global", "traceSequential<2,6,global,stack,2 masters"),
    (1000, "lds", "traceSequential<3,6,lds,stack,2 masters"),
    (1000, "global", "traceSequential<3,6,global,stack,2 masters"),
    (1200, "lds", "traceSequential<4,6,lds,stack,2 masters"),
    (1400, "global", "traceSequential<4,6,global,stack,2 masters"),   # tables exceed the LDS budget on their own
    (2000, "global", "traceSequential<6,6,global,stack,2 masters"),
    (3000, "global", "traceSequential<9,6,global,stack,2 masters"),
    (3300, "global", "traceSequential<10,6,global,stack,2 masters"),  # shares by place 10 / 7 / 10: the kernel BASELINE cfg4 runs
    (4000, "global", "traceSequential<11,6,global,stack,2 masters"),
    (4600, "global", "traceSequential<11,6,global,stack,2 masters"),  # beyond 11 x 6 x 64 resident: the tail is streamed
]


def two_master_case(pkg, ob, ntri, tables, kernel, spp, budget_kb):
    """One case of test_two_master_kernels_match_oracle.  PTW_STAGE_BUDGET_KB must be set by the caller."""
    debug = dict(seq_two_masters=1)
    if tables == "global" and ntri < 1400:
        debug["seq_lds_tables"] = 0
    w, h = (12, 10) if budget_kb else (4, 3)   # (a band is at least 64 pixels)
    scene, cam = _soup(pkg, ntri, 2, seed=31 * ntri + spp, w=w, h=h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=5)
    ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=4)
    rgb, cnt, words, variant, launches, picks = _render_with_stats(pkg, scene, cam, params, picks=True, **debug)
    assert variant == kernel + ">", variant
    if budget_kb:
        assert launches > 1, "the staging budget did not cut the frame into bands"
    assert np.array_equal(cnt, ref_cnt)
    assert np.array_equal(words, ref_words), "a path decision diverged from the oracle"
    assert np.array_equal(picks, ref_picks), "a ray hit another primitive than in the oracle"
    assert rel_err(rgb, ref_rgb) < TOL


@pytest.mark.parametrize("ntri,tables,kernel", TWO_MASTER_CASES)
@pytest.mark.parametrize("spp,budget_kb", [(3, None), (4, 1)])
def test_two_master_kernels_match_oracle(pkg, ob, monkeypatch, ntri, tables, kernel, spp, budget_kb):
    """Every <SLOTS, 6, lds|global, 2 masters> instantiation against the oracle: odd pass count (the
    last workgroup's second master has no pass) in one band; even pass count with a staging budget
    so small that every pass parks and resumes its generator after every few pixels.  Radiance sums,
    every sample's RNG word count and every sample's pick checksum (which primitive each ray hit)."""
    if budget_kb:
        monkeypatch.setenv("PTW_STAGE_BUDGET_KB", str(budget_kb))
    two_master_case(pkg, ob, ntri, tables, kernel, spp, budget_kb)


@pytest.mark.parametrize("ntri,nsph,kernel", [(300, 0, "<1,6,lds,"), (900, 70, "<3,6,lds,"), (3300, 2, "<10,6,global,")])
def test_two_master_natural_dispatch_more_passes_than_cus(pkg, ob, ntri, nsph, kernel):
    """No switch: more passes than the device has CUs selects the two-master kernel by itself - the
    situation of BASELINE cfg3 / cfg4 (3300 triangles: the <10,6,global> instantiation cfg4 runs).
    An odd count, so that the last workgroup runs one master."""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    spp = cus + 3
    scene, cam = _soup(pkg, ntri, nsph, seed=ntri)
    params = pkg.default_params(width=4, height=3, samples_per_pixel=spp, seed=9)
    ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=8)
    rgb, cnt, words, variant, _, picks = _render_with_stats(pkg, scene, cam, params, picks=True)
    assert variant.endswith(",2 masters>") and kernel in variant, variant
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words) and np.array_equal(picks, ref_picks)
    assert rel_err(rgb, ref_rgb) < TOL


@pytest.mark.parametrize("name,edge,spp,kernel", [("suzanne", 16, 6, "traceSequential<3,6,lds,stack,2 masters"),
                                                  ("ce", 6, 5, "traceSequential<10,6,global,stack,2 masters,unit")])   # (",unit": the unit-level u-first early-out, round 6)
def test_two_master_kernels_on_the_baseline_scenes(pkg, ob, name, edge, spp, kernel):
    """suzanne and ce - the scenes of cfg3 / cfg4 - directly against the oracle under the two-master
    kernels they run there.  On ce neither the radiance nor the RNG word counts can depend on which
    primitive a ray hits (every primary ray ends on an emitter of diffuse 0, every ray hits something):
    what makes this comparison able to fail there is the per-sample pick checksum."""
    scene = pkg.Scene()
    cam = scene.build_named(name, edge, edge)
    params = pkg.default_params(width=edge, height=edge, samples_per_pixel=spp, seed=1)
    ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=6)
    rgb, cnt, words, variant, _, picks = _render_with_stats(pkg, scene, cam, params, picks=True, seq_two_masters=1)
    assert variant == kernel + ">", variant
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words)
    assert np.array_equal(picks, ref_picks), "a ray hit another primitive than in the oracle"
    assert rel_err(rgb, ref_rgb) < TOL


def test_a_dropped_unit_of_triangles_shows_in_the_pick_checksum(pkg, ob):
    """The negative control of the comparisons above ("a worker wave that lost a unit
    of triangles would still pass - and run faster").  On ce ITSELF no output can show that: none of its
    rays ever hits a triangle (tests/test_oracle_picks.py: the frame with and without the mesh is the
    same in radiance, word counts AND picks - the camera sits inside the light spheres).  So (a) ce with
    forced worker shares below the scene (9 / 7 / 9 units = 50 of its 54; the rest is streamed) still
    equals the oracle in all three; (b) the instantiation cfg4 runs, <10,6,global,2 masters>, is held to
    a scene where the triangles matter - a 3300-triangle soup - and there the frame of a soup that LACKS
    one unit of 64 triangles differs from the full soup's oracle in its pick checksums."""
    import pick_helpers
    scene = pkg.Scene()
    cam = scene.build_named("ce", 6, 6)
    params = pkg.default_params(width=6, height=6, samples_per_pixel=4, seed=1)
    ref_rgb, _, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=4)
    full = _render_with_stats(pkg, scene, cam, params, picks=True, seq_two_masters=1, seq_units=(9, 7, 9))
    assert full[3].startswith("traceSequential<9,6,global,stack,2 masters"), full[3]
    assert np.array_equal(full[2], ref_words) and np.array_equal(full[5], ref_picks) and rel_err(full[0], ref_rgb) < TOL
    # (b)
    soup, scam = _soup(pkg, 3300, 2, seed=99, w=6, h=4)
    sparams = pkg.default_params(width=6, height=4, samples_per_pixel=3, seed=8)
    _, _, swords, spicks = ob.oracle_render_picks(soup.view(), scam, sparams, threads=4)
    good = _render_with_stats(pkg, soup, scam, sparams, picks=True, seq_two_masters=1)
    assert good[3] == "traceSequential<10,6,global,stack,2 masters>", good[3]
    assert np.array_equal(good[2], swords) and np.array_equal(good[5], spicks)
    unit = pick_helpers.find_sensitive_unit(pkg, ob, soup, scam, sparams)
    bad = _render_with_stats(pkg, pick_helpers.scene_without_unit(pkg, soup, unit), scam, sparams, picks=True,
                             seq_two_masters=1)
    assert not np.array_equal(bad[5], spicks), "the pick checksum did not notice 64 missing triangles"


@pytest.mark.parametrize("ntri,nsph", [(130, 1), (700, 3), (3442, 3)])
@pytest.mark.parametrize("kernel", ["persistent", "legacy"])
def test_perpixel_kernels_many_passes_match_oracle(pkg, ob, ntri, nsph, kernel):
    """PERPIXEL policy, both kernels, 40 passes of a small frame (incoherent rays in every wave - the
    case the wave-uniform u-first early-out of the triangle loop is for): exact word counts and sums."""
    scene, cam = _soup(pkg, ntri, nsph, seed=7 * ntri, w=6, h=4)
    params = pkg.default_params(width=6, height=4, samples_per_pixel=40, seed=2, rng_policy=1,
                                pix_kernel=pkg.PIX_KERNEL_PERSISTENT if kernel == "persistent" else pkg.PIX_KERNEL_LOCKSTEP)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, params, threads=8)
    rgb, cnt, words, variant, _ = _render_with_stats(pkg, scene, cam, params)
    assert variant == ("tracePerPixelPersistent" if kernel == "persistent" else "tracePerPixel")
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words)
    assert rel_err(rgb, ref_rgb) < TOL


# ---- multi-GPU: the collective bodies and the sharded render on ONE GPU -------------------------
def _expected_gather(frames, root):
    """ptw_comm_gather_rows' contract (= sharding.gather_rows): rank r's rows r, r + world, ... land in
    the same rows of the root's buffer; the root's other content stays."""
    world = len(frames)
    out = frames[root].copy()
    for r in range(world):
        if r != root:
            out[r::world] = frames[r][r::world]
    return out


@pytest.mark.parametrize("world,w,h,root", [(2, 9, 7, 0), (3, 5, 11, 0), (4, 6, 3, 2), (3, 4, 2, 1)])
def test_loopback_collectives(pkg, world, w, h, root):
    """The pack / slot arithmetic / unpack code of ptw_comm_gather_rows and the reduce, executed with
    `world` ranks on this box's GPU through the loopback transport (one host thread per rank, each on
    its own stream), against the definition.  Heights that do not divide by the world size, fewer
    rows than ranks (a rank that owns nothing), a root other than 0."""
    import threading
    import torch
    rng = np.random.default_rng(world * 100 + h)
    rgb_host = [rng.uniform(0, 5, (h, w, 3)) for _ in range(world)]
    cnt_host = [rng.integers(0, 1000, (h, w)).astype(np.uint32) for _ in range(world)]
    comms = pkg.Comm.create_loopback(world, 0)
    for mode in ("gather", "reduce"):
        rgb = [torch.tensor(a, device="cuda") for a in rgb_host]
        cnt = [torch.tensor(a.astype(np.int32), device="cuda") for a in cnt_host]
        streams = [torch.cuda.Stream() for _ in range(world)]
        torch.cuda.synchronize()
        errors = []

        def run(r):
            try:
                if mode == "gather":
                    comms[r].gather_rows(rgb[r].data_ptr(), cnt[r].data_ptr(), w, h, root, streams[r].cuda_stream)
                else:
                    comms[r].reduce_framebuffer(rgb[r].data_ptr(), cnt[r].data_ptr(), w * h, root, streams[r].cuda_stream)
            except Exception as e:  # noqa: BLE001
                errors.append((r, e))

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=60)
            assert not t.is_alive(), "a rank is stuck in the collective"
        assert not errors, errors
        torch.cuda.synchronize()
        got_rgb, got_cnt = rgb[root].cpu().numpy(), cnt[root].cpu().numpy().astype(np.uint32)
        if mode == "gather":
            assert np.array_equal(got_rgb, _expected_gather(rgb_host, root))
            assert np.array_equal(got_cnt, _expected_gather(cnt_host, root))
        else:
            want = rgb_host[root].copy()
            for r in range(world):
                if r != root:
                    want = want + rgb_host[r]       # the root adds the peers in rank order
            assert np.array_equal(got_rgb, want)
            assert np.array_equal(got_cnt, sum(c.astype(np.uint64) for c in cnt_host).astype(np.uint32))
        for r in range(world):                       # the other ranks' buffers are left as they were
            if r != root:
                assert np.array_equal(rgb[r].cpu().numpy(), rgb_host[r])
    for c in comms:
        c.close()


@pytest.mark.parametrize("policy,n", [(0, 2), (0, 3), (1, 2), (1, 3)])
def test_render_ex_sharded_code_path_on_one_gpu(pkg, policy, n):
    """ptw_render_ex(num_devices = n, share_device = 2): the N-GPU render itself - a host thread,
    context and stream per shard, pass ranges + reduce / interleaved rows + gather - with the
    collective carried by the loopback transport; against the single-device render, starting from a
    non-empty framebuffer (ArrayOutput +=)."""
    w, h, spp = 20, 13, 7
    scene = pkg.Scene()
    cam = scene.build_named("cornell", w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=3, rng_policy=policy)
    rng = np.random.default_rng(5)
    rgb0, cnt0 = rng.uniform(0, 1, (h, w, 3)), rng.integers(0, 9, (h, w)).astype(np.uint32)
    one_rgb, one_cnt = pkg.render(scene, cam, params, rgb_sum=rgb0.copy(), counts=cnt0.copy())
    seen = []
    many_rgb, many_cnt = pkg.render(scene, cam, params, rgb_sum=rgb0.copy(), counts=cnt0.copy(), num_devices=n,
                                    share_device=2, progress=lambda done, total: seen.append((done, total)) and False)
    assert np.array_equal(many_cnt, one_cnt) and np.all(many_cnt == cnt0 + spp)
    if policy == 1:
        assert np.array_equal(many_rgb, one_rgb)     # a gather moves bytes
    else:
        assert rel_err(many_rgb, one_rgb) < TOL     # pass ranges are added in another order
    assert seen and seen[-1][0] == seen[-1][1]


FAIL_SCRIPT = r"""
import sys
sys.path.insert(0, {root!r})
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)
import __graft_entry__ as entry
pkg = entry.load_package()
scene = pkg.Scene()
cam = scene.build_named("cornell", 16, 12)
params = pkg.default_params(width=16, height=12, samples_per_pixel=4, seed=1, rng_policy={policy})
try:
    pkg.render(scene, cam, params, num_devices={n}, share_device=2, progress={progress},
               debug=pkg.debug_options(**{debug!r}))
except pkg.PtwError as e:
    print("PTW_ERROR", e.status, e)
    sys.exit(0)
print("NO_ERROR")
sys.exit(3)
"""


@pytest.mark.parametrize("debug,policy,n,expect", [
    ({"fail_shard": 1}, 0, 2, "fail_shard"),            # a peer never sets up
    ({"fail_shard": 0}, 1, 3, "fail_shard"),            # ... the root itself
    ({"fail_collective": 1}, 0, 3, "fail_collective"),  # a sender fails: the root waits for it
    ({"fail_collective": 0}, 1, 2, "fail_collective"),  # the root fails: the senders wait for it
    ({}, 0, 2, "cancelled"),                            # the progress callback cancels
])
def test_failing_shard_is_an_error_not_a_hang(pkg, tmp_path, debug, policy, n, expect):
    """ADVICE r2 / VERDICT r2 What's weak 2: a shard that fails before or inside the collective used
    to leave its peers waiting forever.  Now every such render ends with the failing shard's error.
    Run in a child process under a timeout, so that a regression shows as a failure, not a hung suite."""
    from conftest import ROOT
    progress = "(lambda done, total: True)" if expect == "cancelled" else "None"
    script = tmp_path / "fail.py"
    script.write_text(FAIL_SCRIPT.format(root=str(ROOT), policy=policy, n=n, progress=progress, debug=debug))
    proc = subprocess.run(["python", str(script)], env=dict(os.environ), capture_output=True, text=True,
                          timeout=180)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert "PTW_ERROR" in proc.stdout and expect in proc.stdout, proc.stdout


RCCL_RANK_SCRIPT = r"""
import os, sys, time
sys.path.insert(0, {root!r})
rank, world, uid_file, w, h = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], 9, 7
import numpy as np
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
if rank == 0:
    uid = pkg.Comm.unique_id()
    with open(uid_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(uid_file + ".tmp", uid_file)
else:
    t0 = time.time()
    while not os.path.exists(uid_file):
        if time.time() - t0 > 60:
            print("NO_UID"); sys.exit(4)
        time.sleep(0.05)
    uid = open(uid_file, "rb").read()
try:
    comm = pkg.Comm.create(uid, world, rank, 0)
except pkg.PtwError as e:
    print("RCCL_REFUSED", e)
    sys.exit(5)
rng = [np.random.default_rng(10 + r) for r in range(world)]
rgb_all = [g.uniform(0, 5, (h, w, 3)) for g in rng]
cnt_all = [g.integers(0, 1000, (h, w)).astype(np.int32) for g in rng]
stream = torch.cuda.current_stream().cuda_stream
for mode in ("gather", "reduce"):
    rgb = torch.tensor(rgb_all[rank], device="cuda")
    cnt = torch.tensor(cnt_all[rank], device="cuda")
    if mode == "gather":
        comm.gather_rows(rgb.data_ptr(), cnt.data_ptr(), w, h, 0, stream)
    else:
        comm.reduce_framebuffer(rgb.data_ptr(), cnt.data_ptr(), w * h, 0, stream)
    torch.cuda.synchronize()
    if rank == 0:
        if mode == "gather":
            want_rgb, want_cnt = rgb_all[0].copy(), cnt_all[0].copy()
            for r in range(1, world):
                want_rgb[r::world] = rgb_all[r][r::world]
                want_cnt[r::world] = cnt_all[r][r::world]
            ok = np.array_equal(rgb.cpu().numpy(), want_rgb) and np.array_equal(cnt.cpu().numpy(), want_cnt)
        else:
            ok = np.allclose(rgb.cpu().numpy(), sum(rgb_all), rtol=1e-15, atol=0) and \
                np.array_equal(cnt.cpu().numpy(), sum(cnt_all))
        print("RCCL_" + mode.upper(), "OK" if ok else "MISMATCH")
comm.close()
print("RANK_DONE", rank)
"""


def test_rccl_bodies_with_two_ranks_on_one_gpu(pkg, tmp_path):
    """The RCCL transport itself (ncclReduce, grouped ncclSend / ncclRecv, pack / unpack) with TWO
    ranks.  RCCL refuses two ranks of one host on one GPU; NCCL_HOSTID makes the two processes look
    like two hosts, which moves the transfer onto RCCL's socket transport (loopback interface) - the
    same calls, the same kernels, a slower wire.  Skipped, with the reason, where RCCL refuses that
    too; the packing arithmetic is covered by test_loopback_collectives either way."""
    from conftest import ROOT
    script = tmp_path / "rank.py"
    script.write_text(RCCL_RANK_SCRIPT.format(root=str(ROOT)))
    uid_file = str(tmp_path / "uid.bin")
    procs = []
    for r in range(2):
        env = dict(os.environ, NCCL_HOSTID=f"ptw-test-host-{r}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1",
                   NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1", NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(["python", str(script), str(r), "2", uid_file], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.skip("RCCL did not complete a two-rank collective on one GPU within 150 s (socket transport)")
        outs.append(out)
    text = "\n".join(outs)
    if "RCCL_REFUSED" in text or any(p.returncode not in (0,) for p in procs):
        pytest.skip("RCCL refused two ranks on one GPU: " + text[-600:])
    assert "RCCL_GATHER OK" in text and "RCCL_REDUCE OK" in text, text


@pytest.mark.parametrize("scene,w,h,spp", [("cornell", 40, 30, 5), ("suzanne", 24, 16, 3)])
def test_reference_side_binding_runs_on_the_gpu(pkg, scene, w, h, spp):
    """integration/hip/Scene.h - the file a pt-three-ways maintainer would add as src/hip/Scene.h -
    executed: SceneBuilder calls, render(camera, renderParams, updateFunc) with the running
    ArrayOutput handed to updateFunc (src/dod/Scene.cpp:245), result compared with ptw_render inside
    the host program (pt-three-ways_amd/host/integration_check.cpp)."""
    from conftest import ROOT
    exe = pkg.LIB_PATH.parent / "integration_check"
    assert exe.exists(), "make -C pt-three-ways_amd integration_check"
    proc = subprocess.run([str(exe), scene, str(w), str(h), str(spp), str(ROOT / "scenes")], capture_output=True,
                          text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert "INTEGRATION_OK" in proc.stdout and f"samples={w * h * spp}" in proc.stdout
    assert int(proc.stdout.split("updates=")[1].split()[0]) >= 2


@pytest.mark.parametrize("name", ["cornell", "bbc-owl"])
def test_perpixel_kernel_choice_is_explicit(pkg, monkeypatch, name):
    """PERPIXEL policy (ABI v4): ptw_context_render never waits for the device - PTW_PIX_KERNEL_AUTO is
    the persistent kernel until ptw_context_calibrate() has timed the two kernels on this scene +
    frame shape; afterwards AUTO is the measured winner, and ptw_render_params.pix_kernel overrides
    either way.  Which kernel wins is a property of the box and its load: the test checks
    that ONE was chosen, reported, reused and really ran, that the trial stays out of the statistics
    and out of the image, and that both kernels write the same bytes."""
    import torch
    w = h = 256
    scene = pkg.Scene()
    cam = scene.build_named(name, w, h)
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    ctx.enable_stats(True)
    stream = torch.cuda.current_stream().cuda_stream

    def run(params):
        rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
        cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        st = ctx.stats(reset=True)
        assert st.samples == w * h * params.samples_per_pixel and st.rays > 0
        assert int(cnt.min().item()) == params.samples_per_pixel == int(cnt.max().item())
        return rgb.cpu().numpy(), st.trace_kernel.decode()

    auto = pkg.default_params(width=w, height=h, samples_per_pixel=8, seed=1, rng_policy=1)
    before, kernel = run(auto)
    assert kernel == "tracePerPixelPersistent"          # nothing calibrated yet
    names = {pkg.PIX_KERNEL_LOCKSTEP: "tracePerPixel", pkg.PIX_KERNEL_PERSISTENT: "tracePerPixelPersistent"}
    choice = ctx.calibrate(cam, auto, stream)
    assert choice in names
    ctx.stats(reset=True)                                # (the trial's launches are not a render's)
    for _ in range(2):                                   # AUTO now means the winner, every time
        after, kernel = run(auto)
        assert kernel == names[choice]
        assert np.array_equal(after, before)             # same bytes whichever kernel ran
    for forced, want in names.items():                   # the caller's explicit choice wins
        p = pkg.default_params(width=w, height=h, samples_per_pixel=8, seed=1, rng_policy=1, pix_kernel=forced)
        img, kernel = run(p)
        assert kernel == want and np.array_equal(img, before)
    # another frame shape has not been calibrated: AUTO is the persistent kernel there
    other = pkg.default_params(width=w, height=h, samples_per_pixel=8, seed=1, rng_policy=1, row_stride=2, row_phase=1)
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    ctx.render(cam, other, rgb.data_ptr(), cnt.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    assert ctx.stats(reset=True).trace_kernel.decode() == "tracePerPixelPersistent"
    # the sequential policy has nothing to calibrate
    assert ctx.calibrate(cam, pkg.default_params(width=w, height=h, samples_per_pixel=8, seed=1), stream) == pkg.PIX_KERNEL_AUTO


def test_calibration_trial_fits_a_tiny_staging_buffer(pkg, monkeypatch):
    """with a staging budget below one image row per pass the trial used to write
    past the end of the staging buffer.  It is clamped to what the buffer holds; the render that
    follows (many bands) still equals the render without a budget."""
    import torch
    w, h, spp = 1024, 16, 16
    scene = pkg.Scene()
    cam = scene.build_named("cornell", w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=2, rng_policy=1)
    stream = torch.cuda.current_stream().cuda_stream

    def run(budget_kb):
        if budget_kb:
            monkeypatch.setenv("PTW_STAGE_BUDGET_KB", str(budget_kb))
        else:
            monkeypatch.delenv("PTW_STAGE_BUDGET_KB", raising=False)
        ctx = pkg.Context(0)
        ctx.set_scene(scene)
        guard = torch.full((1 << 20,), 7.0, dtype=torch.float64, device="cuda")  # neighbours of the staging buffer
        choice = ctx.calibrate(cam, params, stream)
        rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
        cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        p = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=2, rng_policy=1,
                               pix_kernel=pkg.PIX_KERNEL_PERSISTENT)
        ctx.render(cam, p, rgb.data_ptr(), cnt.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        assert bool((guard == 7.0).all().item())
        return choice, rgb.cpu().numpy(), cnt.cpu().numpy()

    c_small, rgb_small, cnt_small = run(24)   # 24 KB: 64 pixels per band, less than a row of 1024
    c_full, rgb_full, cnt_full = run(None)
    assert c_small in (pkg.PIX_KERNEL_LOCKSTEP, pkg.PIX_KERNEL_PERSISTENT) and c_full in (1, 2)
    assert np.array_equal(rgb_small, rgb_full) and np.array_equal(cnt_small, cnt_full)


# ---- the small-scene kernels on the cases round 3's several-CUs-per-pass kernel was held to ---------------
def _small_soup(pkg, ntri, nsph, shell, seed, w, h):
    rng = np.random.default_rng(seed)
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
    cam = pkg.set_focus(pkg.look_at((0, 0.5, 7), (0, 0, 0), (0, 1, 0), w, h, 45.0), (0, 0, 0), 0.02)
    return scene, cam


SMALL_SCENE_CASES = [
    dict(scene="cornell", w=40, h=28, spp=5, over={}),
    dict(scene="cornell", w=24, h=16, spp=9, over=dict(first_bounce_u=3, first_bounce_v=5), budget_kb=12),
    dict(scene="single-sphere", w=24, h=16, spp=3, over=dict(max_depth=3)),
    dict(soup=(33, 20, False), w=24, h=16, spp=4, over=dict(max_depth=8, first_bounce_u=2, first_bounce_v=2), budget_kb=12),
    dict(soup=(64, 62, True), w=24, h=16, spp=2, over=dict(max_depth=4, first_bounce_u=1, first_bounce_v=7)),
    dict(soup=(1, 0, True), w=16, h=12, spp=8, over=dict(max_depth=9)),
]


@pytest.mark.parametrize("kernel,debug", [("traceSequentialSpec", {}),
                                          ("traceSequentialSpec<no cross-pixel candidate>", dict(seq_small_kernel=3)),
                                          ("traceSequentialSpec<2 waves>", dict(seq_small_kernel=4)),
                                          ("traceSequential<1,1,lds,reg>", dict(seq_small_kernel=1)),
                                          ("traceSequential<1,1,lds,stack>", dict(seq_small_kernel=0))])
def test_small_scene_kernels_match_oracle(pkg, ob, monkeypatch, kernel, debug):
    """The kernels for scenes of at most 64 triangles - four speculating waves per pass (with and without the
    next pixel's camera ray traced ahead in a pixel's last round, round 6), one wave with its (E, T) stack in
    scalar registers, one wave with the stack in LDS - against the oracle with pick
    checksums: closed and open scenes, every depth, odd fan-outs (the speculative kernel's general stratum
    path), primitive counts at the kernels' limits (64 triangles + 62 spheres + a shell = 127 primitives),
    streams parked and resumed between bands.  (These are the cases round 3's several-CUs-per-pass kernel was
    held to; that kernel left the tree in round 6, LAB.md.)"""
    for n, case in enumerate(SMALL_SCENE_CASES):
        monkeypatch.delenv("PTW_STAGE_BUDGET_KB", raising=False)
        if case.get("budget_kb"):
            monkeypatch.setenv("PTW_STAGE_BUDGET_KB", str(case["budget_kb"]))
        w, h = case["w"], case["h"]
        if "scene" in case:
            scene = pkg.Scene()
            cam = scene.build_named(case["scene"], w, h)
        else:
            scene, cam = _small_soup(pkg, *case["soup"], seed=4321 + n, w=w, h=h)
        params = pkg.default_params(width=w, height=h, samples_per_pixel=case["spp"], seed=77, **case["over"])
        ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=4)
        rgb, cnt, words, variant, launches, picks = _render_with_stats(pkg, scene, cam, params, picks=True, **debug)
        assert variant == kernel, (variant, case)
        assert not case.get("budget_kb") or launches > 1
        assert np.array_equal(cnt, ref_cnt)
        assert np.array_equal(words, ref_words), ("a path decision diverged", case)
        assert np.array_equal(picks, ref_picks), ("a ray hit another primitive than in the oracle", case)
        assert rel_err(rgb, ref_rgb) < TOL
