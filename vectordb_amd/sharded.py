"""Base-set sharding of the search path: one shard (vectors + its own graph) per rank.

The reference holds a single in-memory segment (src/segment/segment.rs:11-18) and has no
distributed layer; this module is the scaling axis the GPU build adds.  Every rank searches the
FULL query batch against its shard, the per-shard top-k lists are exchanged with ONE all-gather
and every rank merges them by (sort_key ascending, id ascending) — the order `HNSWIndex::search`
returns (src/vector/hnsw.rs:541-549) with ties broken by id.

On GPUs the exchange is the library's own: a `vdb_shard_group` (one RCCL communicator, created
through the C ABI; torch.distributed only carries the communicator's id from rank 0 to the other
ranks) and `vdb_shard_exchange_dev` / `vdb_shard_search_batch_dev` / `vdb_shard_exact_knn_batch_dev` /
`vdb_shard_segment_search_batch_dev`, which run the all-gather and the merge kernel on the caller's
stream.  A host-side process group ("gloo": CPU tests, rehearsals on a one-GPU box) gathers through
torch.distributed instead and merges with `vdb_topk_merge_dev`; a backend that exports a host-buffer
`topk_merge` (only the CPU checker does) is merged through that, which is how the world_size-2
gloo tests run without a GPU.
"""
from __future__ import annotations

import numpy as np


def shard_range(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous slot range [lo, hi) of `rank`: sizes differ by at most one, lower ranks get the extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_of(slot: int, n: int, world: int) -> int:
    """Rank that owns `slot` under shard_range."""
    q, r = divmod(n, world)
    edge = r * (q + 1)
    return slot // (q + 1) if slot < edge else r + (slot - edge) // max(q, 1)


class ShardedIndex:
    """A rank's shard plus the exchange step.

    `api` is the bound library namespace (`vectordb_amd.api()`), `index` the rank-local
    `api.HNSWIndex`.  `world == 1` (or no process group) makes every call a plain local search.
    """

    def __init__(self, api, index, group=None, device=None, always_exchange=False):
        self.api, self.index, self.group, self.device = api, index, group, device
        self.always_exchange = always_exchange      # run the exchange step even in a world of one (rehearsal)
        self.rank, self.world = 0, 1
        self.vgroup = None                          # vdb_shard_group handle (RCCL through the C ABI)
        backend = None
        if group is not None or _dist_ready():
            import torch.distributed as dist
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
            backend = dist.get_backend(group)
        fn = api.fn
        if hasattr(fn, "shard_group_new") and fn.device_count() > 0 and (backend == "nccl" or (backend is None and always_exchange)):
            self._make_group(backend)

    def _make_group(self, backend):
        """One RCCL communicator over the ranks of the process group, created by the library itself."""
        import ctypes as C
        import torch
        dev = self._torch_device()
        ident = torch.zeros(128, dtype=torch.uint8)
        import os
        if self.rank == 0 and os.environ.get("VDB_SHARD_EXCHANGE", "") != "torch":
            # an id of zeros tells every rank to stay with the process group's own all-gather (VDB_SHARD_EXCHANGE=torch,
            # or a library without a usable RCCL): the decision is rank 0's, so no rank waits for a communicator alone
            buf = (C.c_uint8 * 128)()
            if self.api.fn.shard_group_unique_id(C.cast(buf, C.c_void_p), 128) == 0:
                ident = torch.tensor(list(buf), dtype=torch.uint8)
        if backend is not None and self.world > 1:
            import torch.distributed as dist
            t = ident.to(dev)                        # the process group is RCCL: its tensors live on the GPU
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            ident = t.cpu()
        if not bool(ident.any()):
            return
        raw = bytes(ident.tolist())
        h = C.c_void_p()
        self.api.check(self.api.fn.shard_group_new(raw, 128, self.rank, self.world, dev.index or 0, C.byref(h)))
        self.vgroup = h.value

    def __del__(self):
        if getattr(self, "vgroup", None):
            self.api.fn.shard_group_free(self.vgroup); self.vgroup = None

    def rccl_ranks(self):
        """Size of the library's communicator as RCCL reports it (0: no communicator)."""
        return self.api.fn.shard_group_size(self.vgroup) if self.vgroup else 0

    def search_batch_dev(self, d_q, nq, k, ef_search=0, stream=None):
        """vdb_shard_search_batch_dev: the shard's search, all-gather and merge on one stream; device tensors in and out."""
        return self._dev_call(lambda o, n, s: self.api.fn.shard_search_batch_dev(self.vgroup, self.index._h, d_q.data_ptr(), nq, k, ef_search, o, n, s), d_q, nq, k, stream)

    def exact_knn_batch_dev(self, d_q, nq, k, stream=None):
        return self._dev_call(lambda o, n, s: self.api.fn.shard_exact_knn_batch_dev(self.vgroup, self.index._h, d_q.data_ptr(), nq, k, o, n, s), d_q, nq, k, stream)

    def segment_search_batch_dev(self, segment, d_q, nq, k, ef_search=0, filter=None, stream=None):
        """Segment::search / search_with_filter over every rank's segment (each owns its slice of payloads and tombstones)."""
        return self._dev_call(lambda o, n, s: self.api.fn.shard_segment_search_batch_dev(self.vgroup, segment._h, d_q.data_ptr(), nq, k, ef_search,
                                                                                         filter._h if filter else None, o, n, s), d_q, nq, k, stream)

    def _dev_call(self, call, d_q, nq, k, stream):
        import torch
        if not self.vgroup:
            raise RuntimeError("no vdb_shard_group: the device-buffer sharded calls need the RCCL process group (or always_exchange)")
        out = torch.zeros(nq * k * 16, dtype=torch.uint8, device=d_q.device); cnt = torch.zeros(nq, dtype=torch.int32, device=d_q.device)
        own = stream is None
        if own:                                      # a stream of this object's own: the null stream does not order against the library's
            if not hasattr(self, "_stream"): self._stream = torch.cuda.Stream(device=d_q.device)
            self._stream.wait_stream(torch.cuda.current_stream(d_q.device))
            stream = self._stream.cuda_stream
        self.api.check(call(out.data_ptr(), cnt.data_ptr(), stream))
        if own: torch.cuda.current_stream(d_q.device).wait_stream(self._stream)
        return out, cnt

    # ---- host-buffer path -------------------------------------------------------------------------
    def search_batch(self, queries, top_k, ef_search=0):
        out, cnt = self.index.search_batch(queries, top_k, ef_search=ef_search)
        return self.merge(out, cnt)

    def exact_knn_batch(self, queries, top_k):
        out, cnt = self.index.exact_knn_batch(queries, top_k)
        return self.merge(out, cnt)

    def merge(self, out, cnt):
        """All-gather the per-shard (nq x k) lists and merge them; returns (nq x k, counts) on every rank."""
        if self.world == 1:
            return out, cnt
        import torch
        import torch.distributed as dist
        nq, k = out.shape
        on_gpu = dist.get_backend(self.group) == "nccl"
        dev = self._torch_device() if on_gpu else torch.device("cpu")
        t_out = torch.from_numpy(np.ascontiguousarray(out).view(np.uint8).reshape(-1)).to(dev)
        t_cnt = torch.from_numpy(np.ascontiguousarray(cnt).view(np.int32)).to(dev)
        g_out = torch.empty(self.world * t_out.numel(), dtype=torch.uint8, device=dev)
        g_cnt = torch.empty(self.world * nq, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(g_out, t_out, group=self.group)
        dist.all_gather_into_tensor(g_cnt, t_cnt, group=self.group)
        fn = self.api.fn
        if hasattr(fn, "topk_merge"):                       # host-buffer merge: the CPU checker only
            lists = g_out.cpu().numpy().view(self.api.SCORED_DTYPE)
            counts = g_cnt.cpu().numpy().view(np.uint32)
            m_out = np.zeros((nq, k), dtype=self.api.SCORED_DTYPE); m_cnt = np.zeros(nq, dtype=np.uint32)
            self.api.check(fn.topk_merge(lists.ctypes.data, counts.ctypes.data, self.world, nq, k, m_out.ctypes.data, m_cnt.ctypes.data))
            return m_out, m_cnt
        gdev = self._torch_device()
        m_out, m_cnt = self.merge_dev(g_out.to(gdev), g_cnt.to(gdev), nq, k, gathered=True)
        torch.cuda.synchronize(gdev)
        return (m_out.cpu().numpy().view(self.api.SCORED_DTYPE).reshape(nq, k), m_cnt.cpu().numpy().view(np.uint32))

    # ---- device-resident path ---------------------------------------------------------------------
    def merge_dev(self, d_out, d_cnt, nq, k, gathered=False):
        """d_out: uint8 tensor of nq*k*16 B (or world x that when `gathered`), d_cnt: int32 tensor of nq
        (world x nq).  Runs on torch's current stream of the tensors' device; returns device tensors."""
        import torch
        if self.world == 1 and not gathered and not self.always_exchange:
            return d_out, d_cnt
        if not gathered and self.vgroup:            # RCCL all-gather + merge kernel inside the library, on torch's current stream
            m_out = torch.empty(nq * k * 16, dtype=torch.uint8, device=d_out.device)
            m_cnt = torch.empty(nq, dtype=torch.int32, device=d_out.device)
            cur = torch.cuda.current_stream(d_out.device)
            if cur.cuda_stream == 0:
                # the collective's copies on the null stream wait for every other stream of the device (measured: a search
                # queued on another stream no longer overlaps the exchange): run it on a stream of this object's own,
                # ordered after and before the null stream
                if not hasattr(self, "_stream"): self._stream = torch.cuda.Stream(device=d_out.device)
                self._stream.wait_stream(cur)
                self.api.check(self.api.fn.shard_exchange_dev(self.vgroup, d_out.data_ptr(), d_cnt.data_ptr(), nq, k, m_out.data_ptr(), m_cnt.data_ptr(), self._stream.cuda_stream))
                cur.wait_stream(self._stream)
            else:
                self.api.check(self.api.fn.shard_exchange_dev(self.vgroup, d_out.data_ptr(), d_cnt.data_ptr(), nq, k, m_out.data_ptr(), m_cnt.data_ptr(), cur.cuda_stream))
            return m_out, m_cnt
        if not gathered:
            import torch.distributed as dist
            if dist.get_backend(self.group) == "nccl":
                g_out = torch.empty(self.world * d_out.numel(), dtype=torch.uint8, device=d_out.device)
                g_cnt = torch.empty(self.world * nq, dtype=torch.int32, device=d_out.device)
                dist.all_gather_into_tensor(g_out, d_out, group=self.group)
                dist.all_gather_into_tensor(g_cnt, d_cnt, group=self.group)
            else:
                # a host-side backend (gloo rehearsals on a box with one GPU): the lists are staged through host memory
                h_out, h_cnt = d_out.cpu(), d_cnt.cpu()
                gh_out = torch.empty(self.world * h_out.numel(), dtype=torch.uint8)
                gh_cnt = torch.empty(self.world * nq, dtype=torch.int32)
                dist.all_gather_into_tensor(gh_out, h_out, group=self.group)
                dist.all_gather_into_tensor(gh_cnt, h_cnt, group=self.group)
                g_out, g_cnt = gh_out.to(d_out.device), gh_cnt.to(d_out.device)
            d_out, d_cnt = g_out, g_cnt
        m_out = torch.empty(nq * k * 16, dtype=torch.uint8, device=d_out.device)
        m_cnt = torch.empty(nq, dtype=torch.int32, device=d_out.device)
        cur = torch.cuda.current_stream(d_out.device).cuda_stream
        self.api.check(self.api.fn.topk_merge_dev(d_out.data_ptr(), d_cnt.data_ptr(), self.world, nq, k,
                                                  m_out.data_ptr(), m_cnt.data_ptr(), cur))
        return m_out, m_cnt

    def _torch_device(self):
        import torch
        return torch.device("cuda", torch.cuda.current_device()) if self.device is None else torch.device(self.device)


def _dist_ready():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:
        return False
