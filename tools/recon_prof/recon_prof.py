import ctypes, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ffvvc_amd import abi
abi.LIB_PATH = os.path.join(ROOT, "tools", "recon_prof", "libvvc_mi355_prof.so")
import bench
bench.main(["--only", "intra_recon_wavefront", "--gop", "0", "--frames-in-flight", "1", "--steps", "5", "--warmup", "1", "--no-verify", "--no-cpu-baseline"] + sys.argv[1:])
lib = abi.load()
buf = (ctypes.c_ulonglong * 64)()
lib.vvc355_recon_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.vvc355_recon_prof_read(buf, 0)
v = list(buf)
print("wall clock kHz", v[63], "clock kHz", v[62], "cmd_loop clock64/wall ratio", v[7] / max(1, v[3]))
names = {26: "  body: MIP", 27: "  body: angular quads", 28: "  body: angular single", 14: "n_mip", 15: "n_ang_q", 30: "n_ang_1", 22: "  body: edge fetch+fill", 23: "  body: filter+side proj", 24: "  body: predictor", 25: "  body: pdpc", 0: "ctus", 1: "wait", 2: "tile_load", 3: "cmd_loop", 4: "join", 5: "store_publish", 6: "pred_avail(part of PRED)", 8: "skip_other", 9: "MARK", 10: "PRED", 11: "CCLM", 12: "RESID", 13: "CIIP",
         16: "n_skip", 17: "n_MARK", 18: "n_PRED", 19: "n_CCLM", 20: "n_RESID", 21: "n_CIIP"}
for role in (0, 1):
    n = max(1, v[32 * role])
    print("role", role, "ctus", v[32 * role])
    for k, nm in names.items():
        if k == 0: continue
        x = v[32 * role + k]
        if 16 <= k <= 21 or k in (14, 15, 30):
            print(f"  {nm:28s} {x / n:8.2f} per CTU")
        else:
            cnt = v[32 * role + k + 8] if 8 <= k <= 13 else 0
            print(f"  {nm:28s} {x * 0.01 / n:8.2f} us per CTU" + (f"   {x * 10.0 / cnt:8.1f} ns each" if cnt else ""))
