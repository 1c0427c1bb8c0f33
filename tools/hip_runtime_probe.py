"""Probe: which load orders of torch's bundled HIP runtime and libvdbmi give one working runtime."""
import ctypes, importlib.util, os, sys
mode = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if mode == "torch_first":
    import torch
    torch.zeros(4, device="cuda")
elif mode == "preload":
    d = os.path.join(os.path.dirname(importlib.util.find_spec("torch").origin), "lib")
    ctypes.CDLL(os.path.join(d, "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
import vectordb_amd
v = vectordb_amd.api()
print(mode, "devices", v.fn.device_count())
a = v.score_scan(np.ones((2, 8), np.float32), np.ones((5, 8), np.float32), 2)
print(mode, "scan ok", a.ravel()[:3])
import torch
x = torch.arange(8, device="cuda", dtype=torch.float32)
print(mode, "torch ok", float(x.sum()))
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l or "libhsa-runtime" in l})
print(mode, libs)
