#!/usr/bin/env python3
"""bench.py — throughput of the MI355X VVC pixel-kernel path on synthetic 8K 10-bit CTU batches.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched under
``python -m torch.distributed.run`` with one rank per GPU.  A *step* is one pass of every implemented stage of the
hot path over one synthetic 8K (7680x4320, 4:2:0, 10-bit) frame = 2040 CTUs of 128x128, all inputs resident in HBM.
Frames are independent, so ranks share nothing: weak scaling, no data-path collective (torch.distributed is used only
for the barrier and the max-over-ranks of the elapsed time).  Rank 0 prints ONE JSON line.

PyTorch is plumbing here (device memory, streams, events, the rendezvous); every timed kernel is a hand-written HIP
kernel reached through the C ABI (include/vvc_mi355.h).  The CPU oracle is timed separately, as ``cpu_baseline``.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ffvvc_amd import abi, batch, sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


DOMINANT = "alf_luma_fused"
# stages of the full 8K random-access chain (BASELINE.json configs[3]) that this round does not run yet
MISSING = ["alf_chroma", "alf_cc", "sao", "deblock", "lmcs", "inter_mc", "dmvr_bdof_prof", "itx_residual", "intra_pred"]
CTB = 128


class Stage:
    """One batched launch (or a few) of the hot path over the whole frame."""

    def __init__(self, name, kernel, launch, algorithmic_bytes):
        self.name, self.kernel, self.launch, self.algorithmic_bytes = name, kernel, launch, algorithmic_bytes


class SyntheticFrame:
    """A 4:2:0 frame of uniformly random samples in HBM (checkasm-style inputs), pitched planes."""

    def __init__(self, torch, width, height, bd, seed):
        self.torch, self.width, self.height, self.bd = torch, width, height, bd
        self.itemsize = 1 if bd == 8 else 2
        self.dtype = torch.uint8 if bd == 8 else torch.int16
        self.gen = torch.Generator(device="cuda")
        self.gen.manual_seed(seed)
        self.ncx, self.ncy = (width + CTB - 1) // CTB, (height + CTB - 1) // CTB
        self.n_ctus = self.ncx * self.ncy
        self.keep = []          # device allocations referenced by address from job descriptors

    def plane(self, w, h, fill_random=True):
        pitch_px = batch.plane_pitch(w, self.itemsize) // self.itemsize
        if fill_random:
            t = self.torch.randint(0, 1 << self.bd, (h, pitch_px), device="cuda", generator=self.gen, dtype=self.torch.int32).to(self.dtype)
        else:
            t = self.torch.zeros((h, pitch_px), device="cuda", dtype=self.dtype)
        self.keep.append(t)
        return t

    def upload(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr)).cuda()
        self.keep.append(t)
        return t


def alf_filter_sets(rng, n_sets):
    """APS-like luma filter sets: 25 filters x 12 int8-range coefficients, clip indices 0..3, identity class map."""
    return [(rng.integers(-128, 128, size=(25, 12)).astype(np.int16),
             rng.integers(0, 4, size=(25, 12)).astype(np.uint8),
             rng.permutation(25).astype(np.uint8)) for _ in range(n_sets)]


def build_chain(lib, torch, frame):
    rng = np.random.default_rng(0x5EED0001)
    bd, isz = frame.bd, frame.itemsize
    chain = []

    # ---- ALF luma: classify + coefficient gather + 7x7 diamond, one launch over every CTB of the frame
    src_y, dst_y = frame.plane(frame.width, frame.height), frame.plane(frame.width, frame.height, False)
    pitch = src_y.stride(0) * isz
    sets = alf_filter_sets(rng, 8)
    d_sets = [tuple(frame.upload(a) for a in s) for s in sets]

    def per_ctb(rx, ry):
        s = d_sets[(rx * 3 + ry) % len(d_sets)]
        return s[0].data_ptr(), s[1].data_ptr(), s[2].data_ptr()

    jobs = batch.alf_luma_jobs(dst_y.data_ptr(), src_y.data_ptr(), pitch, isz, frame.width, frame.height, CTB, per_ctb)
    d_jobs = frame.upload(np.frombuffer(bytes(jobs), dtype=np.uint8))
    n_jobs = len(jobs)
    luma_bytes = frame.width * frame.height * isz * 2          # read once + write once
    chain.append(Stage("alf_luma_fused", f"alf_luma_kernel<{bd}, 1>",
                       lambda st: lib.vvc355_alf_luma_batch(st, bd, 1, d_jobs.data_ptr(), n_jobs), luma_bytes))
    return chain


def time_stages(torch, chain, stream, reps):
    out = {}
    for st in chain:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.launch(stream)
        e0.record()
        for _ in range(reps):
            st.launch(stream)
        e1.record()
        torch.cuda.synchronize()
        out[st.name] = e0.elapsed_time(e1) / reps
    return out


def recorded_traffic(root, stage_name):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), or None."""
    path = os.path.join(root, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    return json.load(open(path)).get(stage_name, {}).get("hbm_bytes_per_launch")


def cpu_baseline(root, frame, budget_s):
    """The CPU oracle (oracle/liborc.so, a scalar C restatement: kind "port") timed on ONE host core over a bounded
    sample of the same per-CTU work, reported in the bench's unit (frames/s of the same stage chain)."""
    import subprocess
    so = os.path.join(root, "oracle", "liborc.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle")])
    orc = ctypes.CDLL(so)
    abi.bind(orc, "orc_", {k: v for k, v in abi.SLOT_SIGNATURES.items() if hasattr(orc, "orc_" + k)})
    bd = frame.bd
    rng = np.random.default_rng(7)
    dt = np.uint8 if bd == 8 else np.uint16
    padded = rng.integers(0, 1 << bd, size=(CTB + 16, CTB + 32)).astype(dt)
    dst = np.zeros((CTB, CTB), dt)
    coeff_set, clip_idx, c2f = alf_filter_sets(rng, 1)[0]
    n = (CTB // 4) ** 2
    cls, tr = np.zeros(n, np.int32), np.zeros(n, np.int32)
    grad = np.zeros(((CTB + 4) // 2) ** 2 * 4, np.int32)
    coeff, clip = np.zeros((n, 12), np.int16), np.zeros((n, 12), np.int16)
    off = 8 * padded.shape[1] + 8
    addr = lambda a, o=0: a.ctypes.data + o * a.itemsize  # noqa: E731

    def one_ctu():
        orc.orc_alf_classify(bd, addr(cls), addr(tr), addr(padded, off), padded.shape[1] * padded.itemsize, CTB, CTB, CTB - 4, addr(grad))
        orc.orc_alf_recon_coeff_and_clip(bd, addr(coeff), addr(clip), addr(cls), addr(tr), n, addr(coeff_set), addr(clip_idx), addr(c2f))
        orc.orc_alf_filter_luma(bd, addr(dst), CTB * dst.itemsize, addr(padded, off), padded.shape[1] * padded.itemsize,
                                CTB, CTB, addr(coeff), addr(clip), CTB - 4)

    one_ctu()
    t0 = time.perf_counter()
    one_ctu()
    per = time.perf_counter() - t0
    n_ctus = int(max(8, min(frame.n_ctus, budget_s / max(per, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(n_ctus):
        one_ctu()
    dt_s = time.perf_counter() - t0
    return {
        "value": (n_ctus / frame.n_ctus) / dt_s,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n_ctus} of {frame.n_ctus} CTUs (128x128 luma, {bd}-bit) through the same stage chain "
                  f"(alf classify + recon_coeff_and_clip + filter[LUMA]) in {dt_s:.2f} s on one host core",
    }


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--bd", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="rough budget of the CPU baseline leg")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    lib = abi.load()
    lib.vvc355_set_device(local_rank)

    frame = SyntheticFrame(torch, args.width, args.height, args.bd, seed=0x5EED0001 + rank)
    chain = build_chain(lib, torch, frame)
    stream = torch.cuda.current_stream().cuda_stream

    def run_step(events=None):
        for st in chain:
            if events is not None and st.name == DOMINANT:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                st.launch(stream)
                e1.record()
                events.append((e0, e1))
            else:
                st.launch(stream)

    def barrier():
        sharding.barrier(dist, world, torch.cuda.synchronize)

    for _ in range(args.warmup):
        run_step()
    barrier()
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step(events)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = sharding.max_over_ranks(dist, torch, world, elapsed, "cuda")

    if rank == 0:
        dom = next(st for st in chain if st.name == DOMINANT)
        dom_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
        achieved = dom.algorithmic_bytes / (dom_ms * 1e-3) / 1e9
        traffic = recorded_traffic(ROOT, dom.name)
        out = {
            "metric": "decoded frames/sec (4K/8K 10-bit VVC) per GPU; bit-exact vs FATE",
            "value": world * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic",
            "config": {
                "workload": f"{args.width}x{args.height} {args.bd}-bit 4:2:0 frame = {frame.n_ctus} CTUs of 128x128, one frame per GPU per step, "
                            f"HBM-resident; stages run per step: {', '.join(st.name for st in chain)}",
                "stages_not_yet_in_chain": MISSING,
                "parallelism": f"{world} independent frame stream(s), one per GPU, no collective",
            },
            "stage_ms": {st.name: None for st in chain},
            "roofline": {
                "kernel": dom.kernel,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "ms_per_launch": dom_ms,
                "algorithmic_bytes_per_launch": dom.algorithmic_bytes,
                "traffic": traffic,
            },
        }
        out["stage_ms"] = time_stages(torch, chain, stream, reps=5)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ROOT, frame, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
