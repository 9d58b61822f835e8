"""Level-granular scheduler simulation (measurement helper): K slots, every step each running
candidate advances one level; after every step the frontier walks the finished candidates, dead
candidates (contradicted by the frontier) are killed, free slots take the best nodes of a ranked
list relative to the new frontier that are not present yet.  Counts steps per pixel on the
oracle's real draw-count sequences.  Candidates do not cross the pixel boundary."""
import sys, numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).parent))
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
c = np.load(Path(__file__).parent / f"counts_{name}.npy")
sub = (c[:, 1:] // 3).astype(int)
sub = sub[c[:, 1] > 0][:6000]
p = np.bincount(sub.ravel(), minlength=6)[:6] / sub.size

def greedy(N, maxm=16):
    S = {(0, 0): 1.0}; frontier = {}
    def expand(node, r):
        m, D = node
        if m + 1 >= maxm: return
        for k in range(1, 6):
            if p[k] > 0: frontier[(m + 1, D + k)] = frontier.get((m + 1, D + k), 0.0) + r * p[k]
    expand((0, 0), 1.0); order = [(0, 0)]
    while len(order) < N and frontier:
        best = max(frontier, key=frontier.get); r = frontier.pop(best)
        S[best] = r; order.append(best); expand(best, r)
    return order

def simulate(K, L, kill=True, maxlevels=4):
    ranked = greedy(L)
    steps = 0; started = 0; levels_run = 0
    for row in sub:
        # absolute nodes within the pixel: (j, S) with S in units of 3 draws from the pixel's first sub-sample
        truth = {}; S = 0
        for j in range(16): truth[j] = S; S += row[j]
        fj, fS = 0, 0
        running = {}   # node -> levels done
        done = {}      # node -> count
        while fj < 16:
            # assign
            free = K - len(running)
            if free > 0:
                for (m, D) in ranked:
                    node = (fj + m, fS + D)
                    if node[0] >= 16 or node in running or node in done: continue
                    running[node] = 0; started += 1; free -= 1
                    if free == 0: break
            # one level step for everybody
            steps += 1
            for node in list(running):
                running[node] += 1; levels_run += 1
                j, S0 = node
                # the candidate's own count: if it is the true node use the real count, otherwise a draw from p
                cnt = row[j] if truth[j] == S0 else int(rng.choice(6, p=p))
                if running[node] >= min(cnt, maxlevels):
                    done[node] = cnt; del running[node]
            # frontier walk
            while fj < 16 and (fj, fS) in done:
                fS += done[(fj, fS)]; fj += 1
            # kill what the frontier contradicts
            if kill:
                for node in list(running):
                    m = node[0] - fj; D = node[1] - fS
                    if m < 0 or (m == 0 and D != 0) or D < m or D > 5 * m: del running[node]
                for node in list(done):
                    if node[0] < fj: del done[node]
    n = len(sub)
    return steps / n, started / n, levels_run / n

rng = np.random.default_rng(1)
print("K slots, L ranked: steps/pixel, candidates started/pixel, candidate-levels/pixel")
for K, L in ((4, 8), (16, 32), (32, 64), (32, 128), (64, 96), (64, 128), (64, 256), (128, 256)):
    s, st, lv = simulate(K, L)
    print(f"K={K:3d} L={L:3d}: {s:6.2f} steps  {st:6.1f} started  {lv:6.1f} levels   (rounds-equivalent {s/4:.2f})")
