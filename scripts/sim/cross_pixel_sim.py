"""Round 6: what would tracing the NEXT pixel's camera ray + first hit in the idle waves of a pixel's last
round(s) buy the four-wave speculative kernel (traceSequentialSpec)?  Replays the oracle's real per-sub-sample
draw counts (scripts/sim/get_counts.py) through the shipped lock-step schedule - the same guesses (m1 / m2 from
the running histogram), the same commit rule - with the phase times measured on the device (DESIGN.md 3.1b:
ray 2.1 k cycles, scatter 0.6 k, primary phase 2.5 k, barrier + commit 0.95 k per round).

  A  a wave without a sub-sample (index >= 16) traces pixel i+1's primary at the stream offset its assignment
     implies (index 16 = "the primary of the next pixel": offset = its delta; beyond: the next-best guesses of the
     pixel's end); committed if the pixel ends exactly there -> the next pixel skips its primary phase and pays
     a fetch of the published hit instead.
  B  ... and goes on with the next pixel's sub-sample 0 (a chain of primary + sub-sample: a longer round).

usage: python scripts/sim/cross_pixel_sim.py [scene] [pixels]"""
import sys
from pathlib import Path

import numpy as np

name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
npx = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
c = np.load(Path(__file__).parent / f"counts_{name}.npy")
rows = c[c[:, 1] > 0][:npx].astype(int)     # [camera draws, 16 sub-sample draws]

T_RAY, T_SCATTER, T_PRIMARY, T_ROUND, T_FETCH = 2100, 600, 2500, 950, 250
ABANDON = False
NATURAL_ONLY = False   # only the waves whose assignment IS "sub-sample index 16" (no third-best guess of the pixel's end)


def duration(count):
    return T_SCATTER + min(count // 3, 4) * T_RAY


def guesses(hist):
    order = sorted(hist, key=lambda k: -hist[k])
    m1 = order[0] if order else 15
    m2 = order[1] if len(order) > 1 else m1
    m3 = order[2] if len(order) > 2 else None
    return m1, m2, m3


def run(variant):
    total = rounds = commits = 0
    hist = {}
    have_primary = False      # the current pixel's primary hit was committed by the previous pixel's last round
    skip0 = None              # variant B: ... and so was its sub-sample 0
    saved = tried = last_rounds = 0
    idle_hist = {}
    for pi, row in enumerate(rows):
        cam, sub = row[0], row[1:]
        nxt = rows[pi + 1] if pi + 1 < len(rows) else None
        total += T_FETCH if have_primary else T_PRIMARY
        m1, m2, m3 = guesses(hist)
        j = 0
        if skip0 is not None:
            hist[sub[0]] = hist.get(sub[0], 0) + 1
            j = 1
        have_primary, skip0 = False, None
        while j < 16:
            one = m2 == m1
            assign = [(0, 0), (1, m1), (3, 3 * m1) if one else (1, m2), (2, 2 * m1)]
            busy = [duration(sub[j + io]) for io, _ in assign if j + io < 16]
            if ABANDON:
                # (b) a speculating wave gives up as soon as the frontier's count contradicts it: it looks at a
                # flag once per ray, so it runs on for half a ray on average after wave 0 is done
                t0 = busy[0]
                c0 = sub[j]
                eff = [t0]
                for w in (1, 2, 3):
                    io, dl = assign[w]
                    if j + io >= 16:
                        continue
                    tw = duration(sub[j + io])
                    if w == 1:
                        valid = c0 == m1
                    elif w == 2:
                        valid = (c0 == m2) if not one else (c0 == m1)   # (oneMode: needs three right guesses; first one known at t0)
                    else:
                        valid = c0 in (m1, m2)
                    eff.append(tw if valid else min(tw, t0 + T_RAY // 2))
                busy = eff
            # the pixel's end, as this round sees it: the remaining r sub-samples consume `end` draws
            r = 16 - j
            end = int(sub[j:].sum())
            # idle waves -> candidates for the next pixel's primary (stream offset from this round's frontier)
            cand = []
            if variant and nxt is not None:
                idle = [w for w, (io, _) in enumerate(assign) if j + io >= 16]
                alts = [g for g in ((r - 1) * m1 + m1, (r - 1) * m1 + m2, (r - 1) * m1 + (m3 or 0)) if g]
                for w in idle:
                    io, dl = assign[w]
                    g = dl if j + io == 16 else (None if NATURAL_ONLY else next((a for a in alts if a not in cand), None))
                    if g is not None and g not in cand:
                        cand.append(g)
                idle_hist[len(idle)] = idle_hist.get(len(idle), 0) + 1
            t = max(busy)
            chain = 0
            if cand and variant == "B":
                chain = T_PRIMARY + 300 + duration(nxt[1])
                t = max(t, chain)
            elif cand:
                t = max(t, T_PRIMARY)
            total += t + T_ROUND
            rounds += 1
            D = sub[j]; m = 1
            ok1 = j + 1 < 16 and D == m1
            ok2a = j + 1 < 16 and not ok1 and not one and D == m2
            if ok1 or ok2a:
                D += sub[j + 1]; m = 2
                if j + 2 < 16 and D == 2 * m1:
                    D += sub[j + 2]; m = 3
                    if one and j + 3 < 16 and D == 3 * m1:
                        m = 4
            for q in range(m):
                hist[sub[j + q]] = hist.get(sub[j + q], 0) + 1
            if cand:
                tried += 1
            if j + m == 16:
                last_rounds += 1
                if cand and end in cand:       # (end == what this round committed, since it finished the pixel)
                    have_primary = True
                    saved += 1
                    if variant == "B":
                        skip0 = True
            j += m; commits += m
    n = len(rows)
    return dict(cycles_per_pixel=total / n, rounds=rounds / n, commits_per_round=commits / rounds,
                primary_committed=saved / n, rounds_with_candidate=tried / n,
                idle_waves_hist={k: round(v / n, 3) for k, v in sorted(idle_hist.items())})


base = run(None)
print("shipped  ", base)
ABANDON = True
r = run(None)
print("(b) abandon contradicted candidates", r, "speed-up %.4f" % (base["cycles_per_pixel"] / r["cycles_per_pixel"]))
NATURAL_ONLY = True
r = run("A")
print("(a)+(b), natural-only", r, "speed-up %.4f" % (base["cycles_per_pixel"] / r["cycles_per_pixel"]))
ABANDON = False
for nat in (False, True):
    NATURAL_ONLY = nat
    for v in ("A", "B"):
        r = run(v)
        print(v, "natural-only" if nat else "with alternatives", r, "speed-up %.4f" % (base["cycles_per_pixel"] / r["cycles_per_pixel"]))
