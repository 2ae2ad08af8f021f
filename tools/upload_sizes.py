"""What bench.py uploads per frame (every per-frame table / descriptor array of one frame object), largest first."""
import inspect, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ffvvc_amd import abi
lib = abi.load(); lib.vvc355_set_device(0)
fr = bench.Frame(torch, 7680, 4320, 10, seed=1)
names = {}
orig = fr.upload
def up(arr, per_frame=True):
    t = orig(arr, per_frame)
    fi = inspect.stack()[1]
    names[t.data_ptr()] = f"{fi.lineno}: {(fi.code_context or [''])[0].strip()[:100]}"
    return t
fr.upload = up
bench.build_chain(lib, torch, fr)
print("per-frame MB", sum(h.nbytes for h in fr.host.values()) / 1e6, " device-built MB", sum(h.nbytes for h in fr.derived.values()) / 1e6)
for p, h in sorted(fr.host.items(), key=lambda kv: -kv[1].nbytes)[:22]:
    print(f"{h.nbytes / 1e6:9.2f} MB  {names.get(p)}")
