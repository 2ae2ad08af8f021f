"""Empty inputs: every batched entry and stage driver with nothing to do returns without launching anything and without an error
(a frame without inter blocks, without intra coding units, without LFNST blocks ... is ordinary)."""
import ctypes

import pytest

from ffvvc_amd import abi

pytestmark = pytest.mark.gpu


def test_zero_jobs_are_no_ops(dev):
    dev.vvc355_clear_error()
    dev.vvc355_set_error_policy(1)
    try:
        for name, (ret, args) in abi.BATCH_SIGNATURES.items():
            if not name.endswith("_batch"):
                continue
            fn = getattr(dev, "vvc355_" + name)
            fn(*[None if a == "p" else (10 if i == 1 and a == "i" else 0) for i, a in enumerate(args)])      # stream, bd, jobs = NULL, n = 0, ...
        rf = abi.ReconFrame()            # n_work = 0
        dev.vvc355_recon_frame_pass(None, 10, None, ctypes.addressof(rf))
        inf = abi.InterFrame()           # n_pus = 0
        dev.vvc355_inter_frame_pass(None, 10, None, ctypes.addressof(inf))
        dev.vvc355_inter_frame_build(None, None, ctypes.addressof(inf))
        dev.vvc355_stream_sync(None)
        assert dev.vvc355_last_error() == 0, ctypes.string_at(dev.vvc355_last_error_string())
    finally:
        dev.vvc355_set_error_policy(0)
