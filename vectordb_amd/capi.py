"""ctypes binder for the C ABI declared in include/vdbmi.h.

`bind(cdll, prefix)` returns a namespace whose classes mirror the reference's public Rust API
(src/lib.rs:1-4): DistanceMetric, PayloadValue, Payload, Filter, PayloadIndex, HNSWIndex, Segment,
ScoredPoint, DBError and score().  The same binder serves any library exporting the ABI under a
symbol prefix; the product library uses "vdb_".
"""
import ctypes as C
import enum
from types import SimpleNamespace

import numpy as np


class DistanceMetric(enum.IntEnum):      # src/utils/types.rs:24-29
    Cosine = 0
    Dot = 1
    Euclidean = 2


class ScalarComparisonOp(enum.IntEnum):  # src/utils/payload.rs:131-139
    Eq = 0
    Neq = 1
    Lt = 2
    Lte = 3
    Gt = 4
    Gte = 5


class ListQueryKind(enum.IntEnum):       # src/utils/payload.rs:141-147
    Contains = 0
    Equals = 1
    Length = 2
    ElementCompare = 3


class Arith(enum.IntEnum):
    Fast = 0
    Strict = 1


class FilterMode(enum.IntEnum):
    Literal = 0
    Masked = 1


class DBError(Exception):                # src/utils/errors.rs:9-33
    code = -1

    def __init__(self, msg, code=None):
        super().__init__(msg)
        if code is not None:
            self.code = code


class NotFound(DBError):
    code = 1


class VectorLengthMismatch(DBError):
    code = 2

    def __init__(self, msg, code=None):
        super().__init__(msg, code)
        import re
        m = re.search(r"expected (\d+), got (\d+)", msg)
        self.expected = int(m.group(1)) if m else None
        self.actual = int(m.group(2)) if m else None


class InvalidPayload(DBError):
    code = 6


class SearchError(DBError):
    code = 7


class ReferencePanic(DBError):
    """The reference would have panicked here (e.g. score() on unequal lengths, metric.rs:5)."""
    code = 99


class DeviceError(DBError):
    code = 100


class ArgumentError(DBError):
    code = 101


_ERR = {c.code: c for c in (NotFound, VectorLengthMismatch, InvalidPayload, SearchError, ReferencePanic,
                            DeviceError, ArgumentError)}


class ScoredPointC(C.Structure):         # src/vector/hnsw.rs:11-16
    _fields_ = [("id", C.c_uint64), ("raw_score", C.c_float), ("sort_key", C.c_float)]


SCORED_DTYPE = np.dtype([("id", "<u8"), ("raw_score", "<f4"), ("sort_key", "<f4")])
STATS_DTYPE = np.dtype([("rows_scored", "<u4"), ("expanded", "<u4"), ("adj_entries", "<u4"), ("flags", "<u4")])


class OptionsC(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("device", C.c_int32), ("arith", C.c_int32),
                ("filter_mode", C.c_int32), ("beam_width", C.c_int32)]


class LayerCsrC(C.Structure):
    _fields_ = [("n_lists", C.c_size_t), ("owners", C.c_void_p), ("offsets", C.c_void_p), ("neighbors", C.c_void_p),
                ("n_edges", C.c_size_t)]


class ScoredPoint:
    """ScoredPoint (hnsw.rs:11-23): equality compares sort_key only, as the reference's PartialEq."""
    __slots__ = ("id", "raw_score", "sort_key")

    def __init__(self, id, raw_score, sort_key):
        self.id, self.raw_score, self.sort_key = int(id), float(raw_score), float(sort_key)

    def __eq__(self, o):
        return self.sort_key == o.sort_key

    def __repr__(self):
        return f"ScoredPoint(id={self.id}, raw_score={self.raw_score!r}, sort_key={self.sort_key!r})"


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# name -> (restype, argtypes); names without prefix.
_VP, _SZ, _I, _U64, _F, _D = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.c_float, C.c_double
_PI, _PSZ = C.POINTER(C.c_int), C.POINTER(C.c_size_t)
_SIGS = {
    "last_error_message": (C.c_char_p, []),
    "device_count": (_I, []),
    "score": (_I, [_VP, _SZ, _VP, _SZ, _I, C.POINTER(_F)]),
    "score_scan": (_I, [_VP, _SZ, _VP, _SZ, _SZ, _I, _I, _I, _VP]),
    "value_int": (_VP, [C.c_int64]), "value_float": (_VP, [_D]), "value_str": (_VP, [C.c_char_p, _SZ]),
    "value_bool": (_VP, [_I]), "value_list_int": (_VP, [_VP, _SZ]), "value_list_float": (_VP, [_VP, _SZ]),
    "value_list_str": (_VP, [_VP, _VP, _SZ]), "value_list_bool": (_VP, [_VP, _SZ]),
    "value_free": (None, [_VP]), "value_type": (_I, [_VP]), "value_equals": (_I, [_VP, _VP]),
    "value_compare_scalar": (_I, [_VP, _I, _VP, _PI, _PI]),
    "value_evaluate_list_query": (_I, [_VP, _I, _I, _SZ, _VP, _PI, _PI]),
    "payload_new": (_VP, []), "payload_free": (None, [_VP]), "payload_set": (_I, [_VP, C.c_char_p, _VP]),
    "payload_get": (_VP, [_VP, C.c_char_p]),
    "payload_compare_field": (_I, [_VP, C.c_char_p, _I, _VP, _PI]),
    "payload_evaluate_list_field": (_I, [_VP, C.c_char_p, _I, _I, _SZ, _VP, _PI]),
    "filter_match": (_VP, [C.c_char_p, _VP]), "filter_compare": (_VP, [C.c_char_p, _I, _VP]),
    "filter_and": (_VP, [_VP, _SZ]), "filter_or": (_VP, [_VP, _SZ]), "filter_not": (_VP, [_VP]),
    "filter_free": (None, [_VP]), "filter_evaluate": (_I, [_VP, _VP, _PI]),
    "payload_index_new": (_VP, []), "payload_index_free": (None, [_VP]),
    "payload_index_insert": (_I, [_VP, _U64, _VP]), "payload_index_remove": (_I, [_VP, _U64, _VP]),
    "payload_index_query_exact": (_I, [_VP, C.c_char_p, _VP, _VP, _SZ, _PSZ, _PI]),
    "payload_index_all_for_key": (_I, [_VP, C.c_char_p, _VP, _SZ, _PSZ, _PI]),
    "index_new": (_I, [_I, _SZ, _SZ, _SZ, _SZ, _VP, C.POINTER(_VP)]), "index_free": (None, [_VP]),
    "index_insert": (_I, [_VP, _U64, _VP, _SZ]), "index_insert_batch": (_I, [_VP, _VP, _VP, _SZ, _SZ]),
    "index_search": (_I, [_VP, _VP, _SZ, _SZ, _VP, _PSZ]),
    "index_search_batch": (_I, [_VP, _VP, _SZ, _SZ, _SZ, _SZ, _VP, _VP, _VP]),
    "index_search_batch_dev": (_I, [_VP, _VP, _SZ, _SZ, _SZ, _VP, _VP, _VP, _VP]),
    "index_exact_knn_batch": (_I, [_VP, _VP, _SZ, _SZ, _SZ, _VP, _VP]),
    "index_exact_knn_batch_dev": (_I, [_VP, _VP, _SZ, _SZ, _VP, _VP, _VP]),
    "index_exact_knn_batch_dev_noalloc": (_I, [_VP, _VP, _SZ, _SZ, _VP, _VP, _VP]),
    "index_mark_deleted": (_I, [_VP, _U64]), "index_contains": (_I, [_VP, _U64]), "index_len": (_SZ, [_VP]),
    "index_metric": (_I, [_VP]), "index_m": (_SZ, [_VP]), "index_ef": (_SZ, [_VP]),
    "index_max_level_cap": (_SZ, [_VP]), "index_dim": (_SZ, [_VP]), "index_current_max_level": (_SZ, [_VP]),
    "index_get_entry_point": (_I, [_VP, C.POINTER(_U64)]), "index_set_entry_point": (_I, [_VP, _U64]),
    "index_set_current_max_level": (_I, [_VP, _SZ]), "index_get_vector": (_I, [_VP, _U64, _VP, _SZ]),
    "index_layer_neighbors": (_I, [_VP, _SZ, _U64, _VP, _SZ, _PSZ, _PI]),
    "index_maybe_normalize": (_I, [_VP, _VP, _SZ, _VP]), "index_normalize_score": (_F, [_VP, _F]),
    "index_add_bidirectional_edge": (_I, [_VP, _SZ, _U64, _U64]),
    "index_greedy_search_layer_unfiltered": (_I, [_VP, _VP, _SZ, _U64, _SZ, C.POINTER(_U64)]),
    "index_import_graph": (_I, [_VP, _SZ, _VP, _VP, _VP, _VP, _SZ, _VP, _I, _U64, _SZ]),
    "index_export_sizes": (_I, [_VP, _PSZ, _PSZ, _VP, _VP, _SZ]),
    "index_export_points": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "index_export_layer": (_I, [_VP, _SZ, _VP, _VP, _VP]),
    "index_device_info": (_I, [_VP, _PSZ, _PSZ, _PSZ, _PSZ, _PSZ]),
    "index_sync_device": (_I, [_VP]),
    "index_save": (_I, [_VP, C.c_char_p]), "index_load": (_I, [_VP, C.c_char_p]),
    "segment_new": (_I, [_VP, C.POINTER(_VP)]), "segment_free": (None, [_VP]),
    "segment_insert": (_I, [_VP, _VP, _SZ, _VP, C.POINTER(_U64)]),
    "segment_insert_batch": (_I, [_VP, _VP, _SZ, _SZ, _VP, _VP]),
    "segment_get_vector": (_I, [_VP, _U64, _VP, _SZ]), "segment_delete": (_I, [_VP, _U64]),
    "segment_search": (_I, [_VP, _VP, _SZ, _SZ, _VP, _PSZ]),
    "segment_search_with_filter": (_I, [_VP, _VP, _SZ, _SZ, _VP, _VP, _PSZ]),
    "segment_search_unfiltered": (_I, [_VP, _VP, _SZ, _SZ, _VP, _PSZ]),
    "segment_post_filter": (_I, [_VP, _VP, _SZ, _SZ, _VP, _VP, _PSZ]),
    "segment_search_batch": (_I, [_VP, _VP, _SZ, _SZ, _SZ, _SZ, _VP, _VP, _VP]),
    "segment_get_payload": (_VP, [_VP, _U64]), "segment_is_deleted": (_I, [_VP, _U64]),
    "segment_purge": (_I, [_VP]), "segment_hnsw": (_VP, [_VP]), "segment_hnsw_mut": (_VP, [_VP]),
    "segment_payload_index": (_VP, [_VP]), "segment_payload_count": (_SZ, [_VP]),
    "segment_in_place_filtered_search": (_I, [_VP, _VP, _SZ, _SZ, _VP, _VP, _PSZ]),
    "segment_compile_filter": (_I, [_VP, _VP, _PSZ, _PSZ]),
    "segment_save": (_I, [_VP, C.c_char_p]), "segment_load": (_I, [_VP, C.c_char_p]),
    "segment_filter_bitmaps": (_I, [_VP, _VP, _VP, _VP, _SZ]),
    "segment_search_batch_dev": (_I, [_VP, _VP, _SZ, _SZ, _SZ, _VP, _VP, _VP, _VP, _VP]),
    "segment_insert_batch_columns": (_I, [_VP, _VP, _SZ, _SZ, _SZ, _VP, _VP, _VP, _VP]),
    "topk_merge_dev": (_I, [_VP, _VP, _SZ, _SZ, _SZ, _VP, _VP, _VP]),
    "score_scan_dev": (_I, [_VP, _SZ, _VP, _SZ, _SZ, _SZ, _I, _I, _VP, _VP]),
    "default_options": (None, [_VP]),
    "shard_group_unique_id": (_I, [_VP, _SZ]), "shard_group_new": (_I, [_VP, _SZ, _I, _I, _I, C.POINTER(_VP)]),
    "shard_group_free": (None, [_VP]), "shard_group_size": (_I, [_VP]), "shard_group_rank": (_I, [_VP]),
    "shard_exchange_dev": (_I, [_VP, _VP, _VP, _SZ, _SZ, _VP, _VP, _VP]),
    "shard_search_batch_dev": (_I, [_VP, _VP, _VP, _SZ, _SZ, _SZ, _VP, _VP, _VP]),
    "shard_exact_knn_batch_dev": (_I, [_VP, _VP, _VP, _SZ, _SZ, _VP, _VP, _VP]),
    "shard_segment_search_batch_dev": (_I, [_VP, _VP, _VP, _SZ, _SZ, _SZ, _VP, _VP, _VP, _VP]),
}
# entry points that only some libraries export (host-buffer helpers of the checker)
_OPTIONAL = {"set_threads": (None, [_I]), "segment_set_filter_mode": (None, [_VP, _I]),
             "topk_merge": (_I, [_VP, _VP, _SZ, _SZ, _SZ, _VP, _VP])}

ABI_SYMBOLS = sorted(_SIGS)


def bind(cdll, prefix="vdb_", required=None):
    """Bind every ABI symbol of `cdll` and return the API namespace."""
    fn = SimpleNamespace()
    missing = []
    for name, (res, args) in list(_SIGS.items()) + list(_OPTIONAL.items()):
        try:
            f = getattr(cdll, prefix + name)
        except AttributeError:
            if name in _SIGS and (required is None or name in required):
                missing.append(prefix + name)
            continue
        f.restype, f.argtypes = res, args
        setattr(fn, name, f)
    if missing and required is None and prefix == "vdb_":
        raise ImportError(f"library lacks ABI symbols: {missing}")

    def check(rc):
        if rc != 0:
            msg = fn.last_error_message().decode("utf-8", "replace")
            raise _ERR.get(rc, DBError)(msg, rc)

    def results(buf, n):
        return [ScoredPoint(buf[i].id, buf[i].raw_score, buf[i].sort_key) for i in range(n)]

    class PayloadValue:
        """PayloadValue (payload.rs:8-18).  Construct with PayloadValue.Int(3), .Str("x"), ..."""

        def __init__(self, handle, owned=True, keep=None):
            self._h, self._owned, self._keep = handle, owned, keep

        def __del__(self):
            if getattr(self, "_owned", False) and self._h:
                fn.value_free(self._h)
                self._h = None

        @classmethod
        def Int(cls, v): return cls(fn.value_int(int(v)))
        @classmethod
        def Float(cls, v): return cls(fn.value_float(float(v)))
        @classmethod
        def Str(cls, v):
            b = v.encode("utf-8"); return cls(fn.value_str(b, len(b)))
        @classmethod
        def Bool(cls, v): return cls(fn.value_bool(1 if v else 0))
        @classmethod
        def ListInt(cls, v):
            a = np.asarray(list(v), dtype=np.int64); return cls(fn.value_list_int(_ptr(a), len(a)))
        @classmethod
        def ListFloat(cls, v):
            a = np.asarray(list(v), dtype=np.float64); return cls(fn.value_list_float(_ptr(a), len(a)))
        @classmethod
        def ListStr(cls, v):
            bs = [s.encode("utf-8") for s in v]
            arr = (C.c_char_p * len(bs))(*bs)
            lens = (C.c_size_t * len(bs))(*[len(b) for b in bs])
            return cls(fn.value_list_str(C.cast(arr, _VP), C.cast(lens, _VP), len(bs)))
        @classmethod
        def ListBool(cls, v):
            a = np.asarray([1 if x else 0 for x in v], dtype=np.uint8); return cls(fn.value_list_bool(_ptr(a), len(a)))

        @property
        def type(self): return fn.value_type(self._h)
        def __eq__(self, o): return isinstance(o, PayloadValue) and bool(fn.value_equals(self._h, o._h))
        def __hash__(self): return id(self)

        def compare_scalar(self, op, other):          # payload.rs:150-186 -> Some(bool)/None
            some, out = C.c_int(0), C.c_int(0)
            check(fn.value_compare_scalar(self._h, int(op), other._h, C.byref(some), C.byref(out)))
            return bool(out.value) if some.value else None

        def evaluate_list_query(self, kind, *, value=None, op=ScalarComparisonOp.Eq, n=0):   # payload.rs:188-229
            some, out = C.c_int(0), C.c_int(0)
            check(fn.value_evaluate_list_query(self._h, int(kind), int(op), n, value._h if value else None,
                                               C.byref(some), C.byref(out)))
            return bool(out.value) if some.value else None

    class Payload:
        """Payload (payload.rs:21-129)."""

        def __init__(self, handle=None, owned=True, keep=None):
            self._h = handle if handle is not None else fn.payload_new()
            self._owned, self._keep = owned, keep

        def __del__(self):
            if getattr(self, "_owned", False) and self._h:
                fn.payload_free(self._h); self._h = None

        @classmethod
        def default(cls): return cls()
        def set(self, key, value): check(fn.payload_set(self._h, key.encode(), value._h))
        def get(self, key):
            h = fn.payload_get(self._h, key.encode())
            return PayloadValue(h, owned=False, keep=self) if h else None
        def compare_field(self, field, op, other):
            out = C.c_int(0)
            check(fn.payload_compare_field(self._h, field.encode(), int(op), other._h, C.byref(out)))
            return bool(out.value)
        def evaluate_list_field(self, field, kind, *, value=None, op=ScalarComparisonOp.Eq, n=0):
            out = C.c_int(0)
            check(fn.payload_evaluate_list_field(self._h, field.encode(), int(kind), int(op), n,
                                                 value._h if value else None, C.byref(out)))
            return bool(out.value)

    class Filter:
        """Filter (filters.rs:5-19)."""

        def __init__(self, handle): self._h = handle
        def __del__(self):
            if getattr(self, "_h", None): fn.filter_free(self._h); self._h = None
        @classmethod
        def Match(cls, key, value): return cls(fn.filter_match(key.encode(), value._h))
        @classmethod
        def Compare(cls, key, op, value): return cls(fn.filter_compare(key.encode(), int(op), value._h))
        @classmethod
        def _nary(cls, f, children):
            arr = (C.c_void_p * len(children))(*[c._h for c in children])
            return cls(f(C.cast(arr, _VP), len(children)))
        @classmethod
        def And(cls, children): return cls._nary(fn.filter_and, list(children))
        @classmethod
        def Or(cls, children): return cls._nary(fn.filter_or, list(children))
        @classmethod
        def Not(cls, child): return cls(fn.filter_not(child._h))

    def evaluate_filter(filter, payload):            # filters.rs:23-75
        out = C.c_int(0)
        check(fn.filter_evaluate(filter._h, payload._h, C.byref(out)))
        return bool(out.value)

    def _id_set(call):
        n, some = C.c_size_t(0), C.c_int(0)
        check(call(None, 0, C.byref(n), C.byref(some)))
        if not some.value:
            return None
        buf = np.zeros(max(n.value, 1), dtype=np.uint64)
        check(call(_ptr(buf), len(buf), C.byref(n), C.byref(some)))
        return set(int(x) for x in buf[: n.value])

    class PayloadIndex:
        """PayloadIndex (stores.rs:6-83)."""

        def __init__(self, handle=None, owned=True, keep=None):
            self._h = handle if handle is not None else fn.payload_index_new()
            self._owned, self._keep = owned, keep
        def __del__(self):
            if getattr(self, "_owned", False) and self._h: fn.payload_index_free(self._h); self._h = None
        def insert(self, id, payload): check(fn.payload_index_insert(self._h, id, payload._h))
        def remove(self, id, payload): check(fn.payload_index_remove(self._h, id, payload._h))
        def query_exact(self, key, value):
            return _id_set(lambda b, c, n, s: fn.payload_index_query_exact(self._h, key.encode(), value._h, b, c, n, s))
        def all_for_key(self, key):
            return _id_set(lambda b, c, n, s: fn.payload_index_all_for_key(self._h, key.encode(), b, c, n, s))

    def score(a, b, metric):                          # metric.rs:4-12
        a, b = _f32(a), _f32(b)
        out = C.c_float(0)
        check(fn.score(_ptr(a), len(a), _ptr(b), len(b), int(metric), C.byref(out)))
        return out.value

    def score_scan(queries, base, metric, arith=Arith.Strict, device=0):
        q, b = _f32(queries), _f32(base)
        out = np.zeros((q.shape[0], b.shape[0]), dtype=np.float32)
        check(fn.score_scan(_ptr(q), q.shape[0], _ptr(b), b.shape[0], q.shape[1], int(metric), int(arith), device, _ptr(out)))
        return out

    class HNSWIndex:
        """HNSWIndex (hnsw.rs:59-765)."""

        def __init__(self, metric, m, ef, max_level_cap, dim, *, seed=0, device=0, arith=Arith.Fast,
                     filter_mode=FilterMode.Literal, beam_width=0, _handle=None, _owned=True, _keep=None):
            if _handle is not None:
                self._h, self._owned, self._keep = _handle, _owned, _keep
                return
            opts = OptionsC(seed, device, int(arith), int(filter_mode), int(beam_width))
            h = C.c_void_p()
            check(fn.index_new(int(metric), m, ef, max_level_cap, dim, C.cast(C.byref(opts), _VP), C.byref(h)))
            self._h, self._owned, self._keep = h.value, True, None

        @classmethod
        def new(cls, metric, m, ef, max_level_cap, dim, **kw): return cls(metric, m, ef, max_level_cap, dim, **kw)
        def __del__(self):
            if getattr(self, "_owned", False) and self._h: fn.index_free(self._h); self._h = None
        def _release(self):
            h, self._owned = self._h, False
            return h

        def insert(self, id, vector):
            v = _f32(vector); check(fn.index_insert(self._h, id, _ptr(v), len(v)))
        def insert_batch(self, ids, vectors):
            v = _f32(vectors); i = np.ascontiguousarray(ids, dtype=np.uint64)
            check(fn.index_insert_batch(self._h, _ptr(i), _ptr(v), v.shape[0], v.shape[1]))
        def search(self, query, top_k):
            q = _f32(query); buf = (ScoredPointC * max(top_k, 1))(); n = C.c_size_t(0)
            check(fn.index_search(self._h, _ptr(q), len(q), top_k, C.cast(buf, _VP), C.byref(n)))
            return results(buf, n.value)
        def search_batch(self, queries, top_k, ef_search=0, with_stats=False):
            q = _f32(queries); nq = q.shape[0]
            out = np.zeros((nq, top_k), dtype=SCORED_DTYPE); cnt = np.zeros(nq, dtype=np.uint32)
            st = np.zeros(nq, dtype=STATS_DTYPE) if with_stats else None
            check(fn.index_search_batch(self._h, _ptr(q), nq, q.shape[1], top_k, ef_search, _ptr(out), _ptr(cnt), _ptr(st)))
            return (out, cnt, st) if with_stats else (out, cnt)
        def exact_knn_batch(self, queries, top_k):
            q = _f32(queries); nq = q.shape[0]
            out = np.zeros((nq, top_k), dtype=SCORED_DTYPE); cnt = np.zeros(nq, dtype=np.uint32)
            check(fn.index_exact_knn_batch(self._h, _ptr(q), nq, q.shape[1], top_k, _ptr(out), _ptr(cnt)))
            return out, cnt
        def mark_deleted(self, id): check(fn.index_mark_deleted(self._h, id))
        def contains(self, id): return bool(fn.index_contains(self._h, id))
        def len(self): return fn.index_len(self._h)
        def __len__(self): return fn.index_len(self._h)
        def metric(self): return DistanceMetric(fn.index_metric(self._h))
        def m(self): return fn.index_m(self._h)
        def ef(self): return fn.index_ef(self._h)
        def max_level_cap(self): return fn.index_max_level_cap(self._h)
        def dim(self): return fn.index_dim(self._h)
        def current_max_level(self): return fn.index_current_max_level(self._h)
        def get_entry_point(self):
            out = C.c_uint64(0)
            return out.value if fn.index_get_entry_point(self._h, C.byref(out)) else None
        def set_entry_point(self, id): check(fn.index_set_entry_point(self._h, id))
        def set_current_max_level(self, l): check(fn.index_set_current_max_level(self._h, l))
        def get_vector(self, id):
            out = np.zeros(self.dim(), dtype=np.float32)
            rc = fn.index_get_vector(self._h, id, _ptr(out), len(out))
            if rc == 1: return None
            check(rc); return out
        def layer_neighbors(self, level, id):
            n, some = C.c_size_t(0), C.c_int(0)
            check(fn.index_layer_neighbors(self._h, level, id, None, 0, C.byref(n), C.byref(some)))
            if not some.value: return None
            buf = np.zeros(max(n.value, 1), dtype=np.uint64)
            check(fn.index_layer_neighbors(self._h, level, id, _ptr(buf), len(buf), C.byref(n), C.byref(some)))
            return [int(x) for x in buf[: n.value]]
        def maybe_normalize(self, v):
            v = _f32(v); out = np.zeros_like(v)
            check(fn.index_maybe_normalize(self._h, _ptr(v), len(v), _ptr(out))); return out
        def normalize_score(self, raw): return fn.index_normalize_score(self._h, raw)
        def add_bidirectional_edge(self, level, a, b): check(fn.index_add_bidirectional_edge(self._h, level, a, b))
        def greedy_search_layer_unfiltered(self, query, entry, level):
            q = _f32(query); out = C.c_uint64(0)
            check(fn.index_greedy_search_layer_unfiltered(self._h, _ptr(q), len(q), entry, level, C.byref(out)))
            return out.value
        def sync_device(self): check(fn.index_sync_device(self._h))
        def device_info(self):
            v = [C.c_size_t(0) for _ in range(5)]
            check(fn.index_device_info(self._h, *[C.byref(x) for x in v]))
            return dict(zip(("n_slots", "row_stride_floats", "l0_edges", "upper_edges", "bytes_total"), (x.value for x in v)))

        def export_graph(self):
            """-> dict(ids, levels, vectors, deleted, layers=[(owners, offsets, neighbors)], entry, current_max_level)"""
            n, nl = C.c_size_t(0), C.c_size_t(0)
            lists = np.zeros(64, dtype=np.uint64); edges = np.zeros(64, dtype=np.uint64)
            check(fn.index_export_sizes(self._h, C.byref(n), C.byref(nl), _ptr(lists), _ptr(edges), 64))
            n, nl = n.value, nl.value
            ids = np.zeros(n, np.uint64); lv = np.zeros(n, np.uint32)
            vec = np.zeros((n, self.dim()), np.float32); dl = np.zeros(n, np.uint8)
            check(fn.index_export_points(self._h, _ptr(ids), _ptr(lv), _ptr(vec), _ptr(dl)))
            layers = []
            for l in range(nl):
                ow = np.zeros(int(lists[l]), np.uint32); of = np.zeros(int(lists[l]) + 1, np.uint64)
                nb = np.zeros(max(int(edges[l]), 1), np.uint32)
                check(fn.index_export_layer(self._h, l, _ptr(ow), _ptr(of), _ptr(nb)))
                layers.append((ow, of, nb[: int(edges[l])]))
            return dict(ids=ids, levels=lv, vectors=vec, deleted=dl, layers=layers,
                        entry=self.get_entry_point(), current_max_level=self.current_max_level())

        def save(self, path):
            """One graph file (layout: include/vdbmi.h); a backend without the entry point writes it from export_graph()."""
            if hasattr(fn, "index_save"):
                check(fn.index_save(self._h, str(path).encode())); return
            from . import graph_file
            graph_file.write_graph(path, self.export_graph(), metric=int(self.metric()), m=self.m(), ef=self.ef(),
                                   max_level_cap=self.max_level_cap())
        def load(self, path):
            if hasattr(fn, "index_load"):
                check(fn.index_load(self._h, str(path).encode())); return
            from . import graph_file
            cfg, g = graph_file.read_graph(path)
            if cfg["metric"] != int(self.metric()) or cfg["dim"] != self.dim():
                raise ArgumentError(f"{path} holds metric {cfg['metric']}, dimension {cfg['dim']}")
            self.import_graph(g)

        def import_graph(self, g):
            ids = np.ascontiguousarray(g["ids"], np.uint64); lv = np.ascontiguousarray(g["levels"], np.uint32)
            vec = _f32(g["vectors"]); dl = np.ascontiguousarray(g["deleted"], np.uint8)
            keep = []
            arr = (LayerCsrC * max(len(g["layers"]), 1))()
            for l, (ow, of, nb) in enumerate(g["layers"]):
                ow = np.ascontiguousarray(ow, np.uint32); of = np.ascontiguousarray(of, np.uint64)
                nb = np.ascontiguousarray(nb, np.uint32); keep += [ow, of, nb]
                arr[l] = LayerCsrC(len(ow), ow.ctypes.data, of.ctypes.data, nb.ctypes.data, len(nb))
            e = g["entry"]
            check(fn.index_import_graph(self._h, len(ids), _ptr(ids), _ptr(lv), _ptr(vec), _ptr(dl), len(g["layers"]),
                                        C.cast(arr, _VP), 0 if e is None else 1, 0 if e is None else e,
                                        g["current_max_level"]))

    class Segment:
        """Segment (segment.rs:11-264)."""

        def __init__(self, hnsw):
            h = C.c_void_p()
            check(fn.segment_new(hnsw._release(), C.byref(h)))
            self._h = h.value
        @classmethod
        def new(cls, hnsw): return cls(hnsw)
        def __del__(self):
            if getattr(self, "_h", None): fn.segment_free(self._h); self._h = None
        def insert(self, vector, payload=None):
            v = _f32(vector); out = C.c_uint64(0)
            check(fn.segment_insert(self._h, _ptr(v), len(v), payload._h if payload else None, C.byref(out)))
            return out.value
        def insert_batch(self, vectors, payloads=None):
            v = _f32(vectors); n = v.shape[0]; out = np.zeros(n, np.uint64)
            arr = None
            if payloads is not None:
                arr = (C.c_void_p * n)(*[p._h if p is not None else None for p in payloads])
            check(fn.segment_insert_batch(self._h, _ptr(v), n, v.shape[1], C.cast(arr, _VP) if arr is not None else None, _ptr(out)))
            return out
        def insert_batch_columns(self, vectors, columns):
            """columns: {key: int64 / float64 / bool array of one value per point}."""
            v = _f32(vectors); n = v.shape[0]; out = np.zeros(n, np.uint64)
            keys, kinds, cols = [], [], []
            for k, a in columns.items():
                a = np.asarray(a)
                if a.dtype == np.bool_: kinds.append(3); a = np.ascontiguousarray(a, np.uint8)
                elif np.issubdtype(a.dtype, np.integer): kinds.append(0); a = np.ascontiguousarray(a, np.int64)
                else: kinds.append(1); a = np.ascontiguousarray(a, np.float64)
                assert a.shape == (n,)
                keys.append(k.encode()); cols.append(a)
            ka = (C.c_char_p * max(len(keys), 1))(*keys); kk = np.asarray(kinds, np.int32)
            ca = (C.c_void_p * max(len(cols), 1))(*[a.ctypes.data for a in cols])
            check(fn.segment_insert_batch_columns(self._h, _ptr(v), n, v.shape[1], len(keys), C.cast(ka, _VP), _ptr(kk), C.cast(ca, _VP), _ptr(out)))
            return out
        def get_vector(self, id):
            out = np.zeros(self.hnsw().dim(), dtype=np.float32)
            rc = fn.segment_get_vector(self._h, id, _ptr(out), len(out))
            if rc == 1: return None
            check(rc); return out
        def delete(self, id): check(fn.segment_delete(self._h, id))
        def _search(self, f, query, top_k, *extra):
            q = _f32(query); buf = (ScoredPointC * max(top_k, 1))(); n = C.c_size_t(0)
            check(f(self._h, _ptr(q), len(q), top_k, *extra, C.cast(buf, _VP), C.byref(n)))
            return results(buf, n.value)
        def search(self, query, top_k): return self._search(fn.segment_search, query, top_k)
        def search_with_filter(self, query, top_k, filter=None):
            return self._search(fn.segment_search_with_filter, query, top_k, filter._h if filter else None)
        def search_unfiltered(self, query, top_k): return self._search(fn.segment_search_unfiltered, query, top_k)
        def post_filter(self, query, top_k, filter=None):
            return self._search(fn.segment_post_filter, query, top_k, filter._h if filter else None)
        def in_place_filtered_search(self, query, top_k, filter=None):
            return self._search(fn.segment_in_place_filtered_search, query, top_k, filter._h if filter else None)
        def search_batch(self, queries, top_k, ef_search=0, filter=None):
            q = _f32(queries); nq = q.shape[0]
            out = np.zeros((nq, top_k), dtype=SCORED_DTYPE); cnt = np.zeros(nq, dtype=np.uint32)
            check(fn.segment_search_batch(self._h, _ptr(q), nq, q.shape[1], top_k, ef_search,
                                          filter._h if filter else None, _ptr(out), _ptr(cnt)))
            return out, cnt
        def get_payload(self, id):
            h = fn.segment_get_payload(self._h, id)
            return Payload(h, owned=False, keep=self) if h else None
        def is_deleted(self, id): return bool(fn.segment_is_deleted(self._h, id))
        def purge(self): check(fn.segment_purge(self._h))
        def hnsw(self): return HNSWIndex(None, 0, 0, 0, 0, _handle=fn.segment_hnsw_mut(self._h), _owned=False, _keep=self)
        def payload_index(self): return PayloadIndex(fn.segment_payload_index(self._h), owned=False, keep=self)
        def payload_count(self): return fn.segment_payload_count(self._h)
        def compile_filter(self, filter):
            a, b = C.c_size_t(0), C.c_size_t(0)
            check(fn.segment_compile_filter(self._h, filter._h, C.byref(a), C.byref(b)))
            return a.value, b.value
        def save(self, path): check(fn.segment_save(self._h, str(path).encode()))
        def load(self, path): check(fn.segment_load(self._h, str(path).encode()))
        def filter_bitmaps(self, filter):
            """-> (pass, err) as bool arrays over the slots (insertion order)."""
            n = self.hnsw().len(); words = (n + 31) // 32
            p = np.zeros(max(words, 1), np.uint32); e = np.zeros(max(words, 1), np.uint32)
            check(fn.segment_filter_bitmaps(self._h, filter._h, _ptr(p), _ptr(e), len(p)))
            unpack = lambda w: np.unpackbits(w.view(np.uint8), bitorder="little")[:n].astype(bool)
            return unpack(p), unpack(e)
        def set_filter_mode(self, mode): fn.segment_set_filter_mode(self._h, int(mode))

    return SimpleNamespace(
        fn=fn, check=check, DistanceMetric=DistanceMetric, ScalarComparisonOp=ScalarComparisonOp,
        ListQueryKind=ListQueryKind, Arith=Arith, FilterMode=FilterMode,
        DBError=DBError, NotFound=NotFound, VectorLengthMismatch=VectorLengthMismatch,
        InvalidPayload=InvalidPayload, SearchError=SearchError, ReferencePanic=ReferencePanic,
        DeviceError=DeviceError, ArgumentError=ArgumentError,
        ScoredPoint=ScoredPoint, PayloadValue=PayloadValue, Payload=Payload, Filter=Filter,
        evaluate_filter=evaluate_filter, PayloadIndex=PayloadIndex, score=score, score_scan=score_scan,
        HNSWIndex=HNSWIndex, Segment=Segment, SCORED_DTYPE=SCORED_DTYPE, STATS_DTYPE=STATS_DTYPE,
    )
