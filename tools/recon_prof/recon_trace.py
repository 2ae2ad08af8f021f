import ctypes, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ffvvc_amd import abi
abi.LIB_PATH = os.path.join(ROOT, "tools", "recon_prof", "libvvc_mi355_prof.so")
import bench
bench.main(["--only", "intra_recon_wavefront", "--gop", "0", "--frames-in-flight", "1", "--steps", "3", "--warmup", "1", "--no-verify", "--no-cpu-baseline"] + sys.argv[1:])
lib = abi.load()
buf = (ctypes.c_ulonglong * (4096 * 8))()
lib.vvc355_recon_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.vvc355_recon_prof_read(buf, 2)
t = np.array(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
np.save(os.path.join(ROOT, "gpurun_out", "recon_trace.npy"), t)
ncx = int(os.environ.get("NCX", "60"))
have = t[:, 0] > 0
n = int(have.sum())
t0 = t[have, 0].min()
us = lambda v: (v - t0) * 0.01
flags = t[:, 6] & 0xff
ncmd = t[:, 6] >> 8
light = (flags & 1) != 0
print("ctus traced", n, "light", int((light & have).sum()), "span us", us(t[have, 5].max()))
# durations
hv = have & ~light
def stat(name, a):
    a = a * 0.01
    print(f"  {name:30s} mean {a.mean():8.2f} p50 {np.median(a):8.2f} p90 {np.percentile(a, 90):8.2f} max {a.max():8.2f} us")
print("heavy CTUs:", int(hv.sum()))
stat("luma wait", t[hv, 1] - t[hv, 0]); stat("luma walk (+store, flag)", t[hv, 2] - t[hv, 1])
stat("chroma wait", t[hv, 3] - t[hv, 0]); stat("chroma walk", t[hv, 4] - t[hv, 3]); stat("join->done", t[hv, 5] - np.maximum(t[hv, 2], t[hv, 4]))
stat("ticket->done", t[hv, 5] - t[hv, 0])
lv = have & light
if lv.any():
    print("light CTUs:", int(lv.sum()))
    stat("wait", t[lv, 1] - t[lv, 0]); stat("walk", t[lv, 5] - t[lv, 1])
# critical path: from the last done flag backwards.  A wave's start is max(ticket time, the flags it waited for).
def deps(rs, role):
    ry, rx = divmod(rs, ncx)
    out = []
    if light[rs]:
        if (flags[rs] & 2) and rx > 0: out.append((rs - 1, 0))
        if (flags[rs] & 4) and ry > 0: out.append((rs - ncx, 0))
        return [(d, r) for d, r in out if have[d]]
    cand = [rs - 1 if rx > 0 else -1, rs - ncx - 1 if rx > 0 and ry > 0 else -1, rs - ncx if ry > 0 else -1, rs - ncx + 1 if ry > 0 and rx + 1 < ncx else -1]
    for d in cand:
        if d < 0 or not have[d]: continue
        if role == 0:
            if light[d]: continue
            out.append((d, 0))
        else:
            out.append((d, 2))          # whole CTU done
    return out
def flag_time(rs, what):     # what: 0 luma flag, 2 done flag
    return t[rs, 2] if what == 0 else t[rs, 5]
end_rs = int(np.argmax(np.where(have, t[:, 5], 0)))
print("last CTU", divmod(end_rs, ncx), "done at", us(t[end_rs, 5]))
# walk back
cur, what = end_rs, 2
path = []
tot = {"luma walk": 0.0, "chroma walk": 0.0, "light walk": 0.0, "join/publish": 0.0, "flag latency": 0.0, "ticket late": 0.0}
for _ in range(400):
    if light[cur]:
        start_wait, end_wait, end = t[cur, 0], t[cur, 1], t[cur, 5]
        role = 0
        seg = ("light walk", (end - end_wait) * 0.01)
    elif what == 0:
        role = 0
        start_wait, end_wait, end = t[cur, 0], t[cur, 1], t[cur, 2]
        seg = ("luma walk", (end - end_wait) * 0.01)
    else:
        # done flag: after both waves; which one was later?
        if t[cur, 4] >= t[cur, 2]:
            role = 1
            start_wait, end_wait, end = t[cur, 0], t[cur, 3], t[cur, 4]
            seg = ("chroma walk", (end - end_wait) * 0.01)
        else:
            role = 0
            start_wait, end_wait, end = t[cur, 0], t[cur, 1], t[cur, 2]
            seg = ("luma walk", (end - end_wait) * 0.01)
        tot["join/publish"] += (t[cur, 5] - end) * 0.01
    tot[seg[0]] += seg[1]
    ds = deps(cur, role)
    path.append((divmod(cur, ncx), "light" if light[cur] else ("luma" if role == 0 else "chroma"), round(us(end_wait), 1), round(seg[1], 1), int(ncmd[cur])))
    if not ds:
        tot["ticket late"] += (end_wait - t0) * 0.01
        break
    d, w = max(ds, key=lambda q: flag_time(q[0], q[1]))
    ft = flag_time(d, w)
    if ft <= start_wait:      # was not waiting for a flag: the ticket came late (workgroup busy elsewhere)
        tot["ticket late"] += (end_wait - start_wait) * 0.01
        # what was this workgroup doing before?  follow the workgroup's previous CTU
        wg = t[cur, 7] >> 32
        prev = [r for r in np.nonzero(have)[0] if (t[r, 7] >> 32) == wg and t[r, 5] <= start_wait + 200]
        if not prev: break
        cur = max(prev, key=lambda r: t[r, 5]); what = 2
        path.append(("wg-prev",))
        continue
    tot["flag latency"] += (end_wait - ft) * 0.01
    cur, what = d, w
print("critical path segments:", len(path))
for k, v in tot.items(): print(f"  {k:14s} {v:9.1f} us")
print("sum", sum(tot.values()))
for p in path[:80]: print("   ", p)
