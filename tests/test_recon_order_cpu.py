"""vvc355_recon_order (host helper of the C ABI, runtime_api.cpp): the ticket order of the in-order pass.  No reference counterpart — the
reference decodes CTUs in raster order on CPU threads (vvc_thread.c) — so the checks are the properties the device pass relies on: a
permutation of the CTUs that have commands in which every CTU follows the CTUs it waits for (recon_one_ctu / recon_light_ctu, intra.hip),
equal to a plain restatement of the rule, and not worse than raster order under a list-scheduling model with few workgroups."""
import heapq

import numpy as np
import pytest

import recon_cases
from ffvvc_amd import abi


def waits_for(ctus, ncx, rs):
    ry, rx = divmod(rs, ncx)
    if ctus[rs]["flags"] & abi.RECON_CTU_LIGHT:
        cand = [rs - 1 if (ctus[rs]["flags"] & abi.RECON_CTU_LUMA_LEFT) and rx else -1, rs - ncx if (ctus[rs]["flags"] & abi.RECON_CTU_LUMA_UP) and ry else -1]
    else:
        cand = [rs - 1 if rx else -1, rs - ncx - 1 if rx and ry else -1, rs - ncx if ry else -1, rs - ncx + 1 if ry and rx + 1 < ncx else -1]
    return [d for d in cand if d >= 0 and ctus[d]["n_cmd"]]


def weight(c):
    return (int(c["n_cmd"]) + 3) // 4 if c["flags"] & abi.RECON_CTU_LIGHT else int(c["n_cmd"])


def restated(ctus, ncx, ncy):
    n = ncx * ncy
    work = [rs for rs in range(n) if ctus[rs]["n_cmd"]]
    below, tail = [0] * n, [0] * n
    for rs in reversed(work):
        tail[rs] = weight(ctus[rs]) + below[rs]
        for d in waits_for(ctus, ncx, rs):
            below[d] = max(below[d], tail[rs])
    left = {rs: len(waits_for(ctus, ncx, rs)) for rs in work}
    succ = {rs: [] for rs in work}
    for rs in work:
        for d in waits_for(ctus, ncx, rs):
            succ[d].append(rs)
    ready = [(-tail[rs], rs) for rs in work if not left[rs]]
    heapq.heapify(ready)
    out = []
    while ready:
        _, rs = heapq.heappop(ready)
        out.append(rs)
        for s in succ[rs]:
            left[s] -= 1
            if not left[s]:
                heapq.heappush(ready, (-tail[s], s))
    return out


def makespan(ctus, ncx, order, slots):
    """Workgroups take the CTUs in `order`, hold their slot while waiting, and finish weight() after the last CTU they wait for."""
    free = [0] * slots
    heapq.heapify(free)
    done = {}
    for rs in order:
        t = heapq.heappop(free)
        done[rs] = max([t] + [done[d] for d in waits_for(ctus, ncx, rs)]) + weight(ctus[rs])
        heapq.heappush(free, done[rs])
    return max(done.values())


def table(rng, ncx, ncy, p_heavy, p_light):
    ctus = np.zeros(ncx * ncy, np.dtype(abi.ReconCtu, align=True))
    kind = rng.choice(3, size=ncx * ncy, p=[1 - p_heavy - p_light, p_heavy, p_light])
    ctus["n_cmd"] = np.where(kind > 0, rng.integers(1, 300, size=ncx * ncy), 0)
    for rs in np.nonzero(kind == 2)[0]:
        ry, rx = divmod(int(rs), ncx)
        ctus[rs]["flags"] = (abi.RECON_CTU_LIGHT | (abi.RECON_CTU_LUMA_LEFT if rx and kind[rs - 1] == 1 else 0) |
                             (abi.RECON_CTU_LUMA_UP if ry and kind[rs - ncx] == 1 else 0))
    return ctus


@pytest.mark.parametrize("ncx,ncy,p_heavy,p_light", [(60, 34, 0.2, 0.3), (30, 17, 1.0, 0.0), (12, 7, 0.4, 0.0), (7, 1, 0.5, 0.5), (1, 9, 0.7, 0.3), (5, 5, 0.0, 0.0)])
def test_recon_order(ncx, ncy, p_heavy, p_light):
    lib = abi.load()
    rng = np.random.default_rng(0x5EED0EA0 + ncx)
    ctus = table(rng, ncx, ncy, p_heavy, p_light)
    order = recon_cases.critical_order(lib, ctus, ncx, ncy).tolist()
    work = np.nonzero(ctus["n_cmd"])[0].tolist()
    assert sorted(order) == work
    at = {rs: i for i, rs in enumerate(order)}
    assert all(at[d] < at[rs] for rs in order for d in waits_for(ctus, ncx, rs))
    assert order == restated(ctus, ncx, ncy)
    if len(work) > 64:
        assert makespan(ctus, ncx, order, 16) <= makespan(ctus, ncx, work, 16)
