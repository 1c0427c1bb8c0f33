"""Reader and writer of the graph file that `vdb_index_save` / `vdb_index_load` use (layout: include/vdbmi.h).

Works on the dictionaries of `HNSWIndex.export_graph()` / `import_graph()`, so a graph built by any
backend that exposes those two calls can be shipped as one file and loaded by the GPU library.
The reference keeps nothing on disk (README.md:32); the format exists for fixtures and hand-over.
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"VDBGRAF1"
_HEADER = struct.Struct("<8sII" + "Q" * 12)          # 112 bytes


def _pad(f, nbytes):
    if nbytes % 8:
        f.write(b"\0" * (8 - nbytes % 8))


def _put(f, arr, dtype):
    a = np.ascontiguousarray(arr, dtype=dtype)
    f.write(a.tobytes()); _pad(f, a.nbytes)


def _get(f, count, dtype):
    nbytes = count * np.dtype(dtype).itemsize
    raw = f.read(nbytes)
    if len(raw) != nbytes:
        raise ValueError("graph file is truncated")
    if nbytes % 8:
        f.read(8 - nbytes % 8)
    return np.frombuffer(raw, dtype=dtype).copy()


def write_graph(path, g, *, metric, m, ef, max_level_cap, seed=0, level_draws=0, sample_draws=0):
    """g: dict(ids, levels, vectors, deleted, layers=[(owners, offsets, neighbors)], entry, current_max_level)."""
    vec = np.ascontiguousarray(g["vectors"], np.float32)
    n = len(g["ids"]); dim = vec.shape[1] if vec.ndim == 2 else 0
    entry = g["entry"]
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, 1, int(metric), m, ef, max_level_cap, dim, n, len(g["layers"]),
                             0 if entry is None else 1, 0 if entry is None else int(entry), int(g["current_max_level"]),
                             level_draws, sample_draws, seed))
        _put(f, g["ids"], np.uint64); _put(f, g["levels"], np.uint32); _put(f, g["deleted"], np.uint8); _put(f, vec, np.float32)
        for owners, offsets, neighbors in g["layers"]:
            f.write(struct.pack("<QQ", len(owners), len(neighbors)))
            _put(f, owners, np.uint32); _put(f, offsets, np.uint64); _put(f, neighbors, np.uint32)


def read_graph(path):
    """-> (config dict, graph dict as import_graph takes it)."""
    with open(path, "rb") as f:
        return _read_graph(f)


def _read_graph(f):
    raw = f.read(_HEADER.size)
    if len(raw) != _HEADER.size:
        raise ValueError("graph file is truncated")
    (magic, version, metric, m, ef, cap, dim, n, n_layers, has_entry, entry, cur_max, level_draws, sample_draws,
     seed) = _HEADER.unpack(raw)
    if magic != MAGIC or version != 1:
        raise ValueError("not a version-1 graph file")
    ids = _get(f, n, np.uint64); levels = _get(f, n, np.uint32); deleted = _get(f, n, np.uint8)
    vectors = _get(f, n * dim, np.float32).reshape(n, dim)
    layers = []
    for _ in range(n_layers):
        n_lists, n_edges = struct.unpack("<QQ", f.read(16))
        layers.append((_get(f, n_lists, np.uint32), _get(f, n_lists + 1, np.uint64), _get(f, n_edges, np.uint32)))
    cfg = dict(metric=metric, m=m, ef=ef, max_level_cap=cap, dim=dim, seed=seed, level_draws=level_draws, sample_draws=sample_draws)
    g = dict(ids=ids, levels=levels, vectors=vectors, deleted=deleted, layers=layers,
             entry=entry if has_entry else None, current_max_level=cur_max)
    return cfg, g


SEGMENT_MAGIC = b"VDBSEGM1"
_KINDS = ("Int", "Float", "Str", "Bool", "ListInt", "ListFloat", "ListStr", "ListBool")


def _u64(f):
    raw = f.read(8)
    if len(raw) != 8:
        raise ValueError("segment file is truncated")
    return struct.unpack("<Q", raw)[0]


def _string(f):
    n = _u64(f)
    raw = f.read(n)
    if len(raw) != n:
        raise ValueError("segment file is truncated")
    if n % 8:
        f.read(8 - n % 8)
    return raw.decode("utf-8")


def _value(f):
    kind = _u64(f)
    if kind == 0: return ("Int", struct.unpack("<q", f.read(8))[0])
    if kind == 1: return ("Float", struct.unpack("<d", f.read(8))[0])
    if kind == 2: return ("Str", _string(f))
    if kind == 3: return ("Bool", _u64(f) != 0)
    if kind == 4: return ("ListInt", _get(f, _u64(f), np.int64).tolist())
    if kind == 5: return ("ListFloat", _get(f, _u64(f), np.float64).tolist())
    if kind == 6: return ("ListStr", [_string(f) for _ in range(_u64(f))])
    if kind == 7: return ("ListBool", [bool(b) for b in _get(f, _u64(f), np.uint8)])
    raise ValueError(f"unknown payload value kind {kind}")


def read_segment(path):
    """A file written by `vdb_segment_save` -> (config, graph, dict(next_id, deleted=[ids], payloads={id: {key: (kind, value)}}))."""
    with open(path, "rb") as f:
        cfg, g = _read_graph(f)
        if f.read(8) != SEGMENT_MAGIC:
            raise ValueError("the file holds an index but no segment section")
        next_id, n_del, n_pay = _u64(f), _u64(f), _u64(f)
        deleted = [_u64(f) for _ in range(n_del)]
        payloads = {}
        for _ in range(n_pay):
            pid, nf = _u64(f), _u64(f)
            payloads[pid] = {}
            for _ in range(nf):
                key = _string(f)
                payloads[pid][key] = _value(f)
    return cfg, g, dict(next_id=next_id, deleted=deleted, payloads=payloads)
