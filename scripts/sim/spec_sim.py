"""Speculation-schedule simulator for the SEQUENTIAL kernel (measurement helper).
Round model: at the frontier (sub-sample j of a pixel, known stream offset) N candidate nodes
(m, D) = "sub-sample j+m starting D units (3 draws) after the frontier" are traced in parallel;
the frontier then advances along the true chain while its nodes are in the candidate set.
Candidates never cross the pixel boundary (variant V1)."""
import sys, numpy as np
from pathlib import Path
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
c = np.load(Path(__file__).parent / f"counts_{name}.npy")
sub = (c[:, 1:] // 3).astype(int)          # levels reached per sub-sample, 1..5
hitmask = c[:, 1] > 0
sub = sub[hitmask]
# empirical iid distribution
p = np.bincount(sub.ravel(), minlength=6)[:6] / sub.size
def ranked(N, maxm=16):
    # P(node m at offset D is on the true path) = P(sum of m counts = D); m=0 -> D=0
    dist = {0: 1.0}
    cands = [(1.0, 0, 0)]
    for m in range(1, maxm):
        nd = {}
        for D, pr in dist.items():
            for k in range(1, 6):
                if p[k] > 0: nd[D + k] = nd.get(D + k, 0) + pr * p[k]
        dist = nd
        cands += [(pr, m, D) for D, pr in dist.items()]
    cands.sort(reverse=True)
    return cands[:N]
def simulate(N):
    cs = ranked(N)
    S = set((m, D) for _, m, D in cs)
    rounds = 0; commits = 0; maxlev = 0
    for row in sub:
        j = 0
        while j < 16:
            rounds += 1
            m = 0; D = 0
            while j + m < 16 and (m, D) in S:
                D += row[j + m]; m += 1
            j += m; commits += m
    return rounds / len(sub), commits / rounds
for N in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 496):
    r, a = simulate(N)
    print(f"N={N:4d} rounds/pixel={r:6.2f} commits/round={a:5.2f}")

# ---- what the ranked list looks like, and how much locality (per image row) would add ----
if len(sys.argv) > 2:
    N = int(sys.argv[2])
    cs = ranked(N)
    print("top", N, [(m, D) for _, m, D in cs])
    # per-row adaptive ranking: distribution from the previous row's counts
    W = int(np.sqrt(len(c)))
    rows = sub.reshape(-1, W, 16) if hitmask.all() else None
    if rows is not None:
        rounds = commits = 0
        for y in range(1, rows.shape[0]):
            p = np.bincount(rows[y - 1].ravel(), minlength=6)[:6] / rows[y - 1].size
            S = set((m, D) for _, m, D in ranked(N))
            for row in rows[y]:
                j = 0
                while j < 16:
                    rounds += 1
                    m = 0; D = 0
                    while j + m < 16 and (m, D) in S:
                        D += row[j + m]; m += 1
                    j += m; commits += m
        print(f"per-row adaptive N={N}: rounds/pixel={rounds/((rows.shape[0]-1)*W):.2f} commits/round={commits/rounds:.2f}")
