"""Time-boxed rounds for the four-wave speculative kernel: a workgroup barrier every Q rays; tasks (sub-sample s at
stream offset D) keep their path state across barriers; finished / contradicted tasks are replaced at the barrier."""
import sys, heapq, itertools
import numpy as np
c = np.load(__import__('pathlib').Path(__file__).parent / 'counts_cornell.npy')
rows = c[c[:, 1] > 0][:, 1:].astype(int)
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rows = rows[:NP]
vals, cnts = np.unique(rows, return_counts=True)
prior = {int(v): n / cnts.sum() for v, n in zip(vals, cnts)}
T_RAY, T_SCATTER, T_PRIMARY = 2070, 300, 1250
rng = np.random.default_rng(1)
allc = rows.reshape(-1)

def rays_of(count): return min(count // 3, 4)

def lockstep(rows, T_OVER):
    total = 0; rounds = 0; commits = 0
    m1, m2 = 15, 6
    for row in rows:
        total += T_PRIMARY
        j = 0
        while j < 16:
            cands = [(0, 0), (1, m1), (1, m2), (2, 2 * m1)]
            t = 0
            for io, d in cands:
                if j + io < 16:
                    # a wrong-offset task traces some path: sample a count
                    t = max(t, T_SCATTER + rays_of(row[j + io]) * T_RAY)
            total += t + T_OVER; rounds += 1
            D = row[j]; m = 1
            if j + 1 < 16 and D in (m1, m2):
                ok1 = D == m1
                D2 = D + row[j + 1]; m = 2
                if ok1 and j + 2 < 16 and D == m1 and D2 == 2 * m1: m = 3
            j += m; commits += m
    return total / len(rows), rounds / len(rows), commits / rounds

def quantum(rows, Q, T_OVER, waves=4, depth=3):
    total = 0; nq = 0; ntasks = 0
    for row in rows:
        total += T_PRIMARY
        cum = np.concatenate([[0], np.cumsum(row)])
        cur = 0
        done = set()
        tasks = [None] * waves   # (s, D, need, donerays, started)
        while cur < 16:
            # knowledge about the frontier task
            minc = 3
            for t in tasks:
                if t and t[0] == cur and t[1] == cum[cur]:
                    minc = 3 * (t[3] + 1) if t[3] < t[2] or t[2] == 4 else 3  # still running after t[3] rays -> count > 3*t[3]
                    if t[3] >= 1: minc = 3 * (t[3] + 1)
            # assign idle waves
            running = {(t[0], t[1]) for t in tasks if t}
            # candidate list: best-first over (s, D) with probability
            cand = []
            base = cum[cur]
            cand.append((1.0, cur, base))
            def poss(first):
                return [(v, p) for v, p in prior.items() if (not first) or v >= minc]
            lvl = [(1.0, cur, base)]
            for d in range(1, depth + 1):
                nxt = []
                for pr, s, D in lvl:
                    if s + 1 >= 16: continue
                    ps = poss(s == cur)
                    z = sum(p for _, p in ps)
                    for v, p in ps:
                        nxt.append((pr * p / z, s + 1, D + v))
                # merge same (s, D)
                agg = {}
                for pr, s, D in nxt: agg[(s, D)] = agg.get((s, D), 0) + pr
                lvl = [(pr, s, D) for (s, D), pr in agg.items()]
                cand += lvl
            cand.sort(key=lambda x: -x[0])
            ci = 0
            for w in range(waves):
                if tasks[w] is None:
                    while ci < len(cand) and ((cand[ci][1], cand[ci][2]) in running or (cand[ci][1], cand[ci][2]) in done):
                        ci += 1
                    if ci >= len(cand): break
                    pr, s, D = cand[ci]; ci += 1
                    true = (D == cum[s])
                    need = rays_of(row[s]) if true else rays_of(int(allc[rng.integers(len(allc))]))
                    tasks[w] = [s, D, need, 0, False]
                    running.add((s, D)); ntasks += 1
            # run one quantum
            tq = 0
            for t in tasks:
                if not t: continue
                work = 0
                if not t[4]: work += T_SCATTER; t[4] = True
                r = min(Q, t[2] - t[3])
                work += r * T_RAY; t[3] += r
                tq = max(tq, work)
            total += tq + T_OVER; nq += 1
            # publish finished, commit, kill dead
            for w, t in enumerate(tasks):
                if t and t[3] >= t[2]:
                    done.add((t[0], t[1])); tasks[w] = None
            while cur < 16 and (cur, cum[cur]) in done: cur += 1
            for w, t in enumerate(tasks):
                if not t: continue
                s, D = t[0], t[1]
                dead = s < cur or (s == cur and D != cum[cur]) or (s > cur and (D < cum[cur] + 3 * (s - cur) or D > cum[cur] + 15 * (s - cur)))
                if dead: tasks[w] = None
    return total / len(rows), nq / len(rows), ntasks / len(rows)

ls = lockstep(rows, 900)
print("lock-step (commit+barrier 900): %.0f cycles/sample, %.2f rounds, %.2f commits/round" % ls)
for Q in (1, 2, 4):
    for over in (300, 500, 900):
        q = quantum(rows, Q, over)
        print(f"quantum {Q} rays, overhead {over}: {q[0]:.0f} cycles/sample ({ls[0]/q[0]:.3f}x), {q[1]:.1f} quanta, {q[2]:.1f} tasks per sample")
