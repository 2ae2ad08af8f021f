"""The hipGraph entries of the C ABI (vvc355_graph_begin / _end / _launch / _destroy): a stage-driver call recorded on a stream and
replayed gives what the direct call gives; replaying after the input tables changed picks up the new contents (the graph holds
addresses, not data)."""
import ctypes

import numpy as np
import pytest

import bs_cases
from ffvvc_amd import batch

pytestmark = pytest.mark.gpu


def test_graph_replay_of_a_stage_driver(dev, orc):
    rng = np.random.default_rng(0x5EED0D00)
    t = bs_cases.BsTables(rng, 264, 136, 6, n_slices=2, tiles=False)
    want = bs_cases.run_oracle(orc, t)
    for name in t.OUT:
        getattr(t, name)[:] = 0xEE
    bufs = {name: batch.DeviceBuffer.from_host(getattr(t, name)) for name in t.IN + t.OUT}
    f = t.frame(lambda name: bufs[name].ptr)
    d_f = batch.DeviceBuffer.from_host(np.frombuffer(bytes(f), np.uint8))
    s = dev.vvc355_stream_create()
    dev.vvc355_graph_begin(s)
    dev.vvc355_deblock_bs_pass(s, d_f.ptr, ctypes.addressof(f))        # recorded, not run
    g = dev.vvc355_graph_end(s)
    dev.vvc355_stream_sync(s)
    assert np.all(bufs["bs10"].to_host(np.uint8, want["bs10"].shape) == 0xEE), "capture must not execute the launch"
    dev.vvc355_graph_launch(g, s)
    dev.vvc355_stream_sync(s)
    for name in t.OUT:
        assert np.array_equal(bufs[name].to_host(np.uint8, want[name].shape), want[name]), name
    # new table contents at the same addresses, same graph
    t.cbf0[:] = 1 - t.cbf0
    dev.vvc355_upload(bufs["cbf0"].ptr, t.cbf0.ctypes.data, t.cbf0.nbytes)
    want2 = bs_cases.run_oracle(orc, t)
    dev.vvc355_graph_launch(g, s)
    dev.vvc355_stream_sync(s)
    assert any(not np.array_equal(want2[n], want[n]) for n in t.OUT)
    for name in t.OUT:
        assert np.array_equal(bufs[name].to_host(np.uint8, want2[name].shape), want2[name]), name
    dev.vvc355_graph_destroy(g)
    dev.vvc355_stream_destroy(s)
