"""Candidate sets that maximise the expected number of committed sub-samples per round
(prefix-closed greedy on the reach probability), and the round counts they give on real data."""
import sys, numpy as np
from pathlib import Path
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
c = np.load(Path(__file__).parent / f"counts_{name}.npy")
sub = (c[:, 1:] // 3).astype(int)
sub = sub[c[:, 1] > 0]
p = np.bincount(sub.ravel(), minlength=6)[:6] / sub.size

def greedy(N, p, maxm=16):
    S = {(0, 0): 1.0}          # node -> reach probability
    frontier = {}              # candidate nodes not in S -> reach if added
    def expand(node, r):
        m, D = node
        if m + 1 >= maxm: return
        for k in range(1, 6):
            if p[k] > 0:
                n2 = (m + 1, D + k)
                if n2 in S: S[n2] += r * p[k]      # (cannot happen in this order, kept for safety)
                else: frontier[n2] = frontier.get(n2, 0.0) + r * p[k]
    expand((0, 0), 1.0)
    order = [(0, 0)]
    while len(S) < N and frontier:
        best = max(frontier, key=frontier.get)
        r = frontier.pop(best)
        S[best] = r
        order.append(best)
        expand(best, r)
    return order, sum(S.values())

def simulate(order, data):
    S = set(order)
    rounds = commits = 0
    for row in data:
        j = 0
        while j < 16:
            rounds += 1
            m = 0; D = 0
            while j + m < 16 and (m, D) in S:
                D += row[j + m]; m += 1
            j += m; commits += m
    return rounds / len(data), commits / rounds

for N in (4, 8, 16, 24, 32, 48, 64):
    order, exp = greedy(N, p)
    r, a = simulate(order, sub)
    print(f"N={N:3d} expected commits/round={exp:5.2f}  measured rounds/pixel={r:5.2f} commits/round={a:5.2f}")
if len(sys.argv) > 2:
    print(greedy(int(sys.argv[2]), p)[0])
