import heapq, os, sys
import numpy as np
t = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'gpurun_out', 'recon_trace.npy'))
ncx = int(os.environ.get("NCX", "60"))
have = t[:, 0] > 0
flags = t[:, 6] & 0xff; light = (flags & 1) != 0
Lw = (t[:, 2] - t[:, 1]) * 0.01; Cw = (t[:, 4] - t[:, 3]) * 0.01; Gw = (t[:, 5] - t[:, 1]) * 0.01
def deps(rs):
    ry, rx = divmod(rs, ncx)
    c = [rs - 1 if rx > 0 else -1, rs - ncx - 1 if rx > 0 and ry > 0 else -1, rs - ncx if ry > 0 else -1, rs - ncx + 1 if ry > 0 and rx + 1 < ncx else -1]
    return [d for d in c if d >= 0 and have[d]]
def ldeps(rs):
    ry, rx = divmod(rs, ncx); o = []
    if (flags[rs] & 2) and rx > 0 and have[rs - 1]: o.append(rs - 1)
    if (flags[rs] & 4) and ry > 0 and have[rs - ncx]: o.append(rs - ncx)
    return o
def sim(order, W, ov=3.0):
    free = [0.0] * W; heapq.heapify(free)
    lf, df = {}, {}
    for rs in order:
        T = heapq.heappop(free) + ov
        if light[rs]:
            s = max([T] + [lf[d] for d in ldeps(rs)]); e = s + Gw[rs]; lf[rs] = df[rs] = e
        else:
            ls = max([T] + [lf[d] for d in deps(rs) if not light[d]]); le = ls + Lw[rs]; lf[rs] = le
            cs = max([T] + [df[d] for d in deps(rs)]); ce = max(cs + Cw[rs], le); e = max(le, ce) + 2; df[rs] = e
        heapq.heappush(free, e)
    return max(df.values())
raster = [int(r) for r in np.nonzero(have)[0]]
# earliest-start estimate with infinitely many workgroups
lf, df, est = {}, {}, {}
for rs in raster:
    if light[rs]:
        s = max([0.0] + [lf[d] for d in ldeps(rs)]); lf[rs] = df[rs] = s + Gw[rs]; est[rs] = s
    else:
        ls = max([0.0] + [lf[d] for d in deps(rs) if not light[d]]); le = ls + Lw[rs]; lf[rs] = le
        cs = max([0.0] + [df[d] for d in deps(rs)]); df[rs] = max(cs + Cw[rs], le) + 2; est[rs] = ls
print("infinite workgroups (critical path):", max(df.values()))
def topo(key):
    indeg = {r: len(alldeps(r)) for r in raster}
    ready = [(key[r], r) for r in raster if indeg[r] == 0]; heapq.heapify(ready); out = []
    while ready:
        _, r = heapq.heappop(ready); out.append(r)
        for s_ in succ[r]:
            indeg[s_] -= 1
            if indeg[s_] == 0: heapq.heappush(ready, (key[s_], s_))
    return out
def alldeps(r): return ldeps(r) if light[r] else deps(r)
succ = {r: [] for r in raster}
for rs in raster:
    for d in alldeps(rs): succ[d].append(rs)
by_est = topo(est)
# depth-level order with unit costs (what a host can compute without knowing durations): level = longest chain of heavy deps
lvl = {}
for rs in raster:
    if light[rs]: lvl[rs] = max([0] + [lvl[d] + (0 if light[d] else 1) for d in ldeps(rs)])
    else: lvl[rs] = max([0] + [lvl[d] + (0 if light[d] else 1) for d in deps(rs)])
by_lvl = topo(lvl)
ncmd = t[:, 6] >> 8
# weighted by command count
cst = {}
for rs in raster:
    dd = ldeps(rs) if light[rs] else deps(rs)
    cst[rs] = max([0] + [cst[d] + (ncmd[d] if not light[d] else ncmd[d] // 4) for d in dd])
by_cmd = topo(cst)
# remaining-chain priority (critical path first), made topological by a ready-list
succ = {r: [] for r in raster}
for rs in raster:
    for d in (ldeps(rs) if light[rs] else deps(rs)): succ[d].append(rs)
tail = {}
for rs in reversed(raster):
    w = ncmd[rs] if not light[rs] else ncmd[rs] // 4
    tail[rs] = w + max([0] + [tail[s] for s in succ[rs]])
indeg = {r: len(ldeps(r) if light[r] else deps(r)) for r in raster}
ready = [(-tail[r], r) for r in raster if indeg[r] == 0]; heapq.heapify(ready); by_tail = []
while ready:
    _, r = heapq.heappop(ready); by_tail.append(r)
    for s in succ[r]:
        indeg[s] -= 1
        if indeg[s] == 0: heapq.heappush(ready, (-tail[s], s))
for W in (256, 512, 768, 1024):
    print(W, "raster", round(sim(raster, W)), "est", round(sim(by_est, W)), "level", round(sim(by_lvl, W)), "cmd-weighted", round(sim(by_cmd, W)), "tail-first", round(sim(by_tail, W)))
