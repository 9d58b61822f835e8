"""Lock-step rounds vs an asynchronous speculative scheduler for the four-wave sequential kernel
(measurement helper; timings calibrated on the phase counters of traceSequentialSpec on Cornell).

Lock-step (the shipped kernel): per round wave 0 traces the frontier sub-sample, waves 1-3 the
guesses (j+1 @ m1, j+1 @ m2 or j+3 @ 3 m1, j+2 @ 2 m1); the round ends when the slowest is done.
Asynchronous: no rounds.  A wave that finishes publishes its result, commits whatever the cursor can
now take, and picks the most useful task nobody is working on; a wave notices after every ray that
its task can no longer be reached from the cursor and abandons it.
Outcome (round 3): the model gives the asynchronous scheduler 1.10x over lock-step rounds with four
waves (1.00x without abandoning unreachable tasks) - not built.  Guesses taken from the previous
pixel's strata, +5 % here, were built and measured: the first guess got better (ok1 0.53 -> 0.57)
at the expense of the second (0.19 -> 0.16), commits per round 2.07 -> 2.09, and the extra scalar
work per round cost more than that bought (Cornell 7.72 -> 7.48 Msamples/s,
profiles/r03f_spatial_guess_probe.txt) - reverted.
usage: python scripts/sim/async_sim.py [scene] [pixels]"""
import heapq
import sys
from pathlib import Path

import numpy as np

name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
npx = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
c = np.load(Path(__file__).parent / f"counts_{name}.npy")
rows = c[c[:, 1] > 0][:npx, 1:].astype(int)   # draws per sub-sample, pixels whose primary ray hit

T_RAY, T_SCATTER, T_PRIMARY = 2100, 600, 2500
T_ROUND_OVERHEAD = 950          # barrier + commit of the lock-step kernel
T_TASK_OVERHEAD, T_COMMIT = 300, 150   # async: lock + pick a task; per committed sub-sample


def rays_of(count):       # rays a sub-sample of `count` draws traces (the level at the depth cap traces none)
    return min(count // 3, 4)


def duration(count):
    return T_SCATTER + rays_of(count) * T_RAY


def lockstep(rows):
    total = 0
    hist = {}
    rounds = commits = 0
    for row in rows:
        total += T_PRIMARY
        order = sorted(hist, key=lambda k: -hist[k])
        m1 = order[0] if order else 15
        m2 = order[1] if len(order) > 1 else m1
        j = 0
        while j < 16:
            one = m2 == m1
            cands = [(0, 0), (1, m1), (3, 3 * m1) if one else (1, m2), (2, 2 * m1)]
            # what each wave would trace: sub-sample j+ioff assumed at frontier+delta; its true count if the
            # assumption holds is row[j+ioff]; a wrong start offset traces SOME path - price it like the true one
            t = max(duration(row[j + io]) for io, _ in cands if j + io < 16)
            total += t + T_ROUND_OVERHEAD
            rounds += 1
            # commit chain
            D = row[j]; m = 1
            ok1 = j + 1 < 16 and D == m1
            ok2a = j + 1 < 16 and not ok1 and not one and D == m2
            if ok1 or ok2a:
                D += row[j + 1]; m = 2
                if j + 2 < 16 and D == 2 * m1:
                    D += row[j + 2]; m = 3
                    if one and j + 3 < 16 and D == 3 * m1:
                        m = 4
            for q in range(m):
                hist[row[j + q]] = hist.get(row[j + q], 0) + 1
            j += m; commits += m
    return total / len(rows), rounds / len(rows), commits / rounds


def asynchronous(rows, waves=4, abort=True):
    total = 0
    hist = {}
    tasks_run = 0
    for row in rows:
        t0 = total + T_PRIMARY                 # all waves compute the primary ray + first surface
        order = sorted(hist, key=lambda k: -hist[k])
        m1 = order[0] if order else 15
        m2 = order[1] if len(order) > 1 else m1
        cum = np.concatenate([[0], np.cumsum(row)])   # true start offset of sub-sample s
        cur = 0                                # cursor: next sub-sample to commit (its offset cum[cur] is known)
        done = {}                              # (s, D) -> finish time of a completed task
        running = {}                           # wave -> (s, D, start, end)
        free = [(t0, w) for w in range(waves)]
        heapq.heapify(free)
        now = t0
        while cur < 16:
            now, w = heapq.heappop(free)
            # finish this wave's task
            if w in running:
                s, D, st, en = running.pop(w)
                done[(s, D)] = en
            # commit everything the cursor can take
            while cur < 16 and (cur, cum[cur]) in done:
                hist[row[cur]] = hist.get(row[cur], 0) + 1
                cur += 1
                now += T_COMMIT
            if cur >= 16:
                break
            # tasks that became unreachable are abandoned at their next ray boundary
            if abort:
                for ww, (s, D, st, en) in list(running.items()):
                    dead = s < cur or (s == cur and D != cum[cur]) or D < cum[cur] + 3 * (s - cur)
                    if dead:
                        k = -(-(now - st - T_SCATTER) // T_RAY) if now > st + T_SCATTER else 0
                        stop = max(now, st + T_SCATTER + k * T_RAY)
                        if stop < en:
                            running.pop(ww)
                            free = [(t, x) for t, x in free if x != ww]
                            heapq.heapify(free)
                            heapq.heappush(free, (stop, ww))
            # pick a task: the true one first, then the guesses, nearest first
            base = cum[cur]
            wish = [(cur, base), (cur + 1, base + m1), (cur + 1, base + m2), (cur + 2, base + 2 * m1),
                    (cur + 2, base + m1 + m2), (cur + 3, base + 3 * m1), (cur + 2, base + 2 * m2), (cur + 4, base + 4 * m1)]
            busy = {(s, D) for s, D, _, _ in running.values()}
            pick = next(((s, D) for s, D in wish if s < 16 and (s, D) not in done and (s, D) not in busy), None)
            if pick is None:
                nxt = min(en for _, _, _, en in running.values()) if running else now + 200
                heapq.heappush(free, (max(nxt, now + 50), w))
                continue
            s, D = pick
            # a task on the true path takes its true duration; a wrong guess traces some path: price it by a sample
            dur = duration(row[s])
            st = now + T_TASK_OVERHEAD
            running[w] = (s, D, st, st + dur)
            tasks_run += 1
            heapq.heappush(free, (st + dur, w))
        total = now
    return total / len(rows), tasks_run / len(rows)


ls, rounds, cpr = lockstep(rows)
print(f"{name}: lock-step   {ls:8.0f} cycles/sample  rounds/sample {rounds:.2f} commits/round {cpr:.2f}")
for ab in (False, True):
    a, tasks = asynchronous(rows, abort=ab)
    print(f"{name}: async abort={ab!s:5s} {a:8.0f} cycles/sample  tasks/sample {tasks:.1f}  speed-up {ls / a:.2f}x")
for w in (5, 6, 8):
    a, tasks = asynchronous(rows, waves=w, abort=True)
    print(f"{name}: async {w} waves {a:8.0f} cycles/sample  tasks/sample {tasks:.1f}  speed-up {ls / a:.2f}x (same per-wave speed assumed)")


def lockstep_spatial(rows):
    """Lock-step rounds with per-stratum guesses taken from the previous pixel's sub-sample of the same
    stratum (first guess: its count if it is one of the two most frequent values, else the mode)."""
    total = 0
    hist = {}
    rounds = commits = 0
    prev = None
    for row in rows:
        total += T_PRIMARY
        order = sorted(hist, key=lambda k: -hist[k])
        g1 = order[0] if order else 15
        g2 = order[1] if len(order) > 1 else g1
        j = 0
        while j < 16:
            def guess(s):      # (first, second) guess for the count of sub-sample s
                if prev is not None and prev[s] in (g1, g2):
                    a = prev[s]
                    return a, (g2 if a == g1 else g1)
                return g1, g2
            a0, b0 = guess(j)
            a1, _ = guess(j + 1) if j + 1 < 16 else (g1, g2)
            # waves: frontier j; j+1 @ a0; j+1 @ b0; j+2 @ a0 + a1
            t = max(duration(row[j + io]) for io in (0, 1, 1, 2) if j + io < 16)
            total += t + T_ROUND_OVERHEAD
            rounds += 1
            m = 1
            if j + 1 < 16 and row[j] in (a0, b0):
                m = 2
                if j + 2 < 16 and row[j] == a0 and row[j + 1] == a1:
                    m = 3
            for q in range(m):
                hist[row[j + q]] = hist.get(row[j + q], 0) + 1
            j += m; commits += m
        prev = row
    return total / len(rows), rounds / len(rows), commits / rounds


sp, r2, c2 = lockstep_spatial(rows)
print(f"{name}: lock-step, guesses from the previous pixel's strata: {sp:8.0f} cycles/sample  rounds/sample {r2:.2f} "
      f"commits/round {c2:.2f}  speed-up {ls / sp:.2f}x")
