"""How lc3t_rs48_map (lc3_enc_pre.inc) was found: an assignment of the 128 outputs of a 48 kHz / 10 ms frame of the 12.8 kHz resampler to
(half-wave, pass) groups of 32 such that the 32 input indices ceil(15 n / 4) of a group fall into 32 different LDS banks, both outputs of a lane of
the same filter phase (n mod 4).  Local search from the plain assignment (n, n + 64); prints the two rows of the table."""
import math, collections, random
r = lambda n: math.ceil(15 * n / 4) % 32
random.seed(3)
grp = {n: n // 32 for n in range(128)}
def cost(a):
    tot = 0
    for g in range(4):
        c = collections.Counter(r(n) for n in range(128) if a[n] == g); tot += sum(v * v for v in c.values())
    return tot
cur = cost(grp); ns = list(range(128))
while cur > 128:
    a, b = random.sample(ns, 2)
    if grp[a] == grp[b] or a % 4 != b % 4: continue
    grp[a], grp[b] = grp[b], grp[a]
    nc = cost(grp)
    if nc <= cur: cur = nc
    else: grp[a], grp[b] = grp[b], grp[a]
G = [[n for n in range(128) if grp[n] == g] for g in range(4)]
n0, n1 = [0] * 64, [0] * 64
for half in range(2):
    a = sorted(G[half], key=lambda n: (n % 4, n)); b = sorted(G[2 + half], key=lambda n: (n % 4, n))
    for j in range(32): n0[half * 32 + j], n1[half * 32 + j] = a[j], b[j]
print(n0); print(n1)
