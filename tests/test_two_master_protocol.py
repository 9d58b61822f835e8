"""CPU model of the barrier protocol of the two-master worker kernel
(pt-three-ways_amd/csrc/ptw_seq_ctx.h: SeqCtx<..., MASTERS = 2>::intersect / workerLoop / stopWorkers, and
the role set-up in traceSequential, ptw_seq_kernel.h): the LOCK STEP the shipped kernels run.  (Round 4's
decoupled protocol - request and answer numbers in LDS, no barrier - left the tree in round 5 with its model,
last revision 916a1dc; round 5's paired form - one barrier per tick, two-ray requests - in round 6, last
revision 89c2934; LAB.md.)

Every wave of the workgroup is a generator that yields at each workgroup barrier; the scheduler
releases a barrier only when ALL live waves have arrived (what s_barrier does), so a protocol in
which one wave waits at a barrier another one never reaches shows up as a deadlock here instead of
as a hung GPU.  Checked for every combination of ray counts of the two passes (including a master
without a pass - an odd pass count - and passes that trace nothing at all):

* nobody deadlocks, and all waves leave after the same barrier;
* every ray a master publishes is searched by every worker exactly once, between the master's two
  barriers, and the master reads the answers of THAT ray;
* while one master's ray is searched, the other master is between its barriers B2 and B1, i.e.
  free to shade (the point of the arrangement).
"""
import itertools
import random

LIVE = 0xFFFFFFFF
WORKERS = 6


class Group:
    def __init__(self):
        self.op = [LIVE, LIVE]          # SeqCommand::op of the two masters
        self.ray = [None, None]         # the published ray (an id)
        self.partials = [[None] * WORKERS, [None] * WORKERS]
        self.log = []                   # (barrier index, master, ray id) per worker search
        self.reads = []                 # (master, ray id, answers)


def master(g, m, rays, has_pass):
    """traceSequential's master branch + intersect() + stopWorkers() for MASTERS == 2."""
    tick = 0
    if not has_pass:
        g.op[m] = 0                     # set before the initial __syncthreads()
    else:
        if m == 1:                      # the second master runs one barrier behind the first
            yield "B"
            tick = 1
        for r in range(rays):
            g.ray[m] = (m, r)           # publish (op stays LIVE)
            yield "B"                   # B1
            yield "B"                   # B2
            tick += 2
            g.reads.append((m, (m, r), list(g.partials[m])))
        g.op[m] = tick                  # no more rays as of my next barrier
    n = tick
    while True:                         # keep the cadence until the other one is done too
        yield "B"
        if g.op[0] <= n and g.op[1] <= n:
            return
        n += 1


def worker(g, w):
    """workerLoop() for MASTERS == 2."""
    n = 0
    while True:
        yield "B"
        m = n & 1
        mine, other = g.op[m], g.op[m ^ 1]
        if mine <= n:
            if other <= n:
                return
            n += 1
            continue
        g.partials[m][w] = (g.ray[m], w)
        if w == 0:
            g.log.append((n, m, g.ray[m]))
        n += 1


def run(rays0, rays1, has1=True, order_seed=None):
    """Between two barriers the waves run concurrently: `order_seed` picks the order in which the
    model runs their code there (None: as listed), so that a result depending on who is first - a race
    - shows up as a difference between seeds."""
    rnd = random.Random(order_seed) if order_seed is not None else None
    g = Group()
    waves = [master(g, 0, rays0, True), master(g, 1, rays1, has1)] + [worker(g, w) for w in range(WORKERS)]
    live = list(range(len(waves)))
    barriers = 0
    left_at = {}
    for _ in range(10 * (rays0 + rays1) + 20):
        arrived, done = [], []
        if rnd:
            rnd.shuffle(live)
        for i in live:
            try:
                next(waves[i])
                arrived.append(i)
            except StopIteration:
                done.append(i)
        for i in done:
            left_at[i] = barriers
        live = arrived
        if not live:
            break
        # s_barrier releases only when every wave of the workgroup that still runs has arrived;
        # a wave that left earlier than the others would let them through with stale data
        assert not done or not live, f"waves {done} left at barrier {barriers} while {live} still wait"
        barriers += 1
    assert not live, "deadlock"
    assert len(set(left_at.values())) == 1
    return g, barriers


def test_lock_step_protocol_all_ray_counts():
    for rays0, rays1, has1 in itertools.product(range(0, 7), range(0, 7), (True, False)):
        if not has1 and rays1:
            continue
        g, barriers = run(rays0, rays1, has1)
        expect = [(0, r) for r in range(rays0)] + ([(1, r) for r in range(rays1)] if has1 else [])
        # every published ray searched exactly once, and read by its master with all answers in
        assert sorted(ray for _, _, ray in g.log) == sorted(expect)
        assert sorted(ray for _, ray, _ in g.reads) == sorted(expect)
        for m, ray, answers in g.reads:
            assert answers == [(ray, w) for w in range(WORKERS)]
        # master m's rays are searched after barriers of parity m, one per two barriers
        for n, m, ray in g.log:
            assert n & 1 == m
        per_master = {0: [n for n, m, _ in g.log if m == 0], 1: [n for n, m, _ in g.log if m == 1]}
        for ns in per_master.values():
            assert all(b - a == 2 for a, b in zip(ns, ns[1:]))
        # the cost: two barriers per ray of the longer pass (plus the stagger and the exit)
        assert barriers <= 2 * max(rays0, rays1) + 3
        # no race: any interleaving of the waves between barriers gives the same schedule
        for seed in range(6):
            g2, b2 = run(rays0, rays1, has1, order_seed=seed)
            assert b2 == barriers and sorted(g2.log) == sorted(g.log) and sorted(g2.reads) == sorted(g.reads)


def worker_rank(wave, masters, workers):
    """traceSequential's `workerRank` (ptw_seq_kernel.h): `tid` order of the worker waves - the waves
    that share a SIMD with a master (waves go to the four SIMDs round robin: wave 4 sits with wave 0,
    wave 5 with wave 1) come last."""
    r = wave - masters
    first_shared, n_shared = 4 - masters, masters
    if r >= first_shared + n_shared:
        r -= n_shared
    elif r >= first_shared:
        r += workers - first_shared - n_shared
    return r


def test_slot_major_assignment_gives_the_empty_slots_to_the_waves_beside_a_master():
    """SeqCtx::slotTriangle: slot s of lane `tid` holds triangle s * lanes + tid.  Every triangle has
    exactly one holder, and the slots beyond the last triangle are empty in whole waves - the ones
    with the highest rank, which are the ones on a master's SIMD."""
    for masters, workers in ((1, 7), (2, 6)):
        ranks = {wave: worker_rank(wave, masters, workers) for wave in range(masters, masters + workers)}
        assert sorted(ranks.values()) == list(range(workers))
        shared = [w for w in ranks if w % 4 < masters and w >= 4]      # same SIMD as a master wave
        assert sorted(ranks[w] for w in shared) == list(range(workers - masters, workers))
        lanes = 64 * workers
        for slots, ntri in ((3, 970), (9, 3442), (2, 129), (4, 1536)):
            if slots * lanes < ntri:
                continue
            holder = {}
            for wave, rank in ranks.items():
                for lane in range(64):
                    tid = rank * 64 + lane
                    for s in range(slots):
                        k = s * lanes + tid
                        if k < ntri:
                            assert k not in holder
                            holder[k] = (wave, s)
            assert len(holder) == ntri
            # a slot is either full in a wave, or empty in it, except in one wave per slot at most
            for s in range(slots):
                partial = [w for w in ranks if 0 < sum(1 for k, (hw, hs) in holder.items() if hw == w and hs == s) < 64]
                assert len(partial) <= 1
            # the waves beside a master never hold more than any other wave
            load = {w: sum(1 for hw, _ in holder.values() if hw == w) for w in ranks}
            assert max(load[w] for w in shared) <= min(load[w] for w in ranks if w not in shared)
