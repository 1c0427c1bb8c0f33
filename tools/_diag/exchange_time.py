"""Host and device time of one shard exchange (all-gather of the records and counts + merge) in a one-rank group."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vectordb_amd
from vectordb_amd.sharded import ShardedIndex
vdb = vectordb_amd.api()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 16, 64, 16, 32, seed=1); h.insert_batch(np.arange(1, 2001, dtype=np.uint64), rng.normal(size=(2000, 32)).astype(np.float32))
sh = ShardedIndex(vdb, h, device=dev, always_exchange=True)
print("rccl ranks", sh.rccl_ranks(), flush=True)
nq, k = 10000, 10
lo = torch.zeros(nq * k * 16, dtype=torch.uint8, device=dev); lc = torch.full((nq,), k, dtype=torch.int32, device=dev)
for mode in ("library", "torch"):
    if mode == "torch":
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1)
        sh2 = ShardedIndex.__new__(ShardedIndex); sh2.__dict__.update(sh.__dict__); sh2.vgroup = None; sh2.group = dist.group.WORLD
        call = lambda: sh2.merge_dev(lo, lc, nq, k)
    else:
        call = lambda: sh.merge_dev(lo, lc, nq, k)
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(50): call()
    e1.record(); host = (time.perf_counter() - t0) / 50; torch.cuda.synchronize()
    print(mode, "host us per call", round(host * 1e6, 1), "device us per call", round(e0.elapsed_time(e1) * 1e3 / 50, 1), flush=True)
