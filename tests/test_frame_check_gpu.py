"""GPU: the bench's own frame (bench.build_chain: every stage of the hot path chained on device-resident planes and tables) checked
stage by stage against the oracle at small picture sizes — the same check bench.py runs at 7680x4320 after its timed region
(`verified`).  Sizes are not multiples of the 128x128 CTU, so the last CTU row / column are partial."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("width,height,bd,noise", [(832, 480, 10, False), (416, 240, 8, False), (832, 480, 10, True), (640, 368, 12, False)])
def test_bench_frame_against_oracle(width, height, bd, noise):
    # own process: bench.py brings PyTorch's HIP runtime up first (device memory, events) and the library's second; the other GPU
    # tests of this session have initialised the library's runtime already
    code = f"import json, bench; print('REPORT ' + json.dumps(bench.self_check({width}, {height}, {bd}, n_ctus=10, noise={noise})))"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    rep = json.loads(next(ln for ln in r.stdout.splitlines() if ln.startswith("REPORT "))[7:])
    for name, r in rep["stages"].items():
        assert r.get("mismatching", 0) == 0, (name, r)
    checked = [name for name, r in rep["stages"].items() if r["checked"]]
    assert {"inter_pred_luma_dmvr_bdof", "inter_pred_chroma", "dequant_itx_add_residual", "intra_tb_dequant_lfnst_itx", "intra_recon_wavefront", "lmcs_inverse_luma", "deblock_bs",
            "deblock_vertical", "deblock_horizontal", "sao", "alf"} <= set(checked)
    if not noise:
        # picture-like content must exercise the tools' decisions both ways and make deblocking actually filter
        st = rep["stats"]
        assert 0.05 < st["dmvr_searched_fraction"] <= 1.0 and 0.02 < st["bdof_applied_fraction"] < 0.98, st
        assert st["deblock_vertical_changed_luma_sample_fraction"] > 0.01, st


@pytest.mark.parametrize("gop,groups", [(4, 3), (8, 2)])
def test_gop_stream_matches_serial_decoding(gop, groups):
    """bench.py --gop: every picture on its own stream, waiting for its two reference pictures through events (three rotating picture
    sets, so a group's first pictures start while the previous group drains).  The decoded pictures must be those of decoding one
    picture at a time in decoding order — and decoding in the wrong order must give different pictures, or the check would say nothing
    about the waits."""
    r = subprocess.run([sys.executable, "bench.py", "--gop", str(gop), "--gop-check", str(groups), "--width", "832", "--height", "480", "--no-cpu-baseline", "--no-upload"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    rep = json.loads(next(ln for ln in r.stdout.splitlines() if ln.startswith('{"gop_check"')))["gop_check"]
    assert rep["concurrent_vs_in_order_mismatching_pictures"] == 0, rep
    assert rep["in_order_vs_reversed_order_differing_pictures"] > 0 and rep["pictures_changed_by_decoding"] == gop * min(groups, 3), rep
