// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's segment search path (rosharma719/vectordb).  Nothing in the
// shipped library may include, link or call this file; it exists so that tests, the smoke check
// and the benchmark's cpu_baseline leg can compare the HIP path with the reference's algorithm.
//
// Pinning: the reference is Rust and no Rust toolchain exists in this build environment, so the
// reference itself was never run.  The oracle is pinned by restating every RNG-independent
// known-answer test of the reference's own suite (tests/utilsvector.rs, tests/hnsw.rs,
// tests/segment.rs, tests/in_place.rs, tests/filters.rs, tests/payload.rs,
// tests/inverted_index.rs) against it.  Exact numeric outputs at large N are pinned by no
// reference test: for those, parity is oracle-vs-HIP only.
//
// Literal-mode rules followed here:
//   * fp32 sums are strict left-to-right, one rounding per operation, no FMA contraction
//     (compile with -ffp-contract=off, never -ffast-math).  Rust's `Iterator::sum::<f32>()`
//     folds from -0.0 (library/core/src/iter/traits/accum.rs, Rust >= 1.83; the reference is
//     edition 2024, i.e. rustc >= 1.85).
//   * `std::collections::BinaryHeap` (Rust std, library/alloc/src/collections/binary_heap) is
//     restated operation by operation (push = sift_up, pop = swap-with-last +
//     sift_down_to_bottom + sift_up, into_sorted_vec = in-place heap sort with
//     sift_down_range, into_iter = backing-array order) so that equal-key order matches what
//     the reference produces.
//   * `sort_by` is a stable sort.
//   * Where the reference depends on HashMap/HashSet iteration order or on an unseeded RNG the
//     oracle fixes a deterministic rule, listed next to each use as "DET-n".
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace ora {

using PointId = uint64_t;                       // src/utils/types.rs:6
enum Metric : int { Cosine = 0, Dot = 1, Euclidean = 2 };   // src/utils/types.rs:25-29

// src/utils/errors.rs:9-33 (declaration order = code)
enum Status : int {
    Ok = 0, NotFound = 1, VectorLengthMismatch = 2, IOError = 3, SerializationError = 4,
    WALCorrupt = 5, InvalidPayload = 6, SearchError = 7, Panic = 99
};
struct DBError {
    Status code; std::string msg;
};
inline DBError err_len_mismatch(size_t expected, size_t actual) {
    return {VectorLengthMismatch, "Vector length mismatch: expected " + std::to_string(expected) +
                                       ", got " + std::to_string(actual)};
}
inline DBError err_invalid_payload(const std::string& s) { return {InvalidPayload, "Invalid payload: " + s}; }
inline DBError err_search(const std::string& s) { return {SearchError, "Search failed: " + s}; }
struct PanicError : std::runtime_error { using std::runtime_error::runtime_error; };

// ---------------------------------------------------------------------------------------------
// Deterministic counter-based RNG replacing rand::rng() (src/vector/hnsw.rs:97, :232-237).
// ---------------------------------------------------------------------------------------------
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t rng_draw(uint64_t seed, uint64_t stream, uint64_t counter) {
    return splitmix64(splitmix64(seed ^ (stream * 0xD1B54A32D192ED03ull)) + counter);
}
enum : uint64_t { STREAM_LEVEL = 1, STREAM_SAMPLE = 2 };

// ---------------------------------------------------------------------------------------------
// Distance metrics — src/vector/metric.rs
// ---------------------------------------------------------------------------------------------
inline float dot_seq(const float* a, const float* b, size_t n) {      // metric.rs:24-26 / :16
    float s = -0.0f;
    for (size_t i = 0; i < n; ++i) { float p = a[i] * b[i]; s = s + p; }
    return s;
}
inline float sqnorm_seq(const float* a, size_t n) {                   // metric.rs:17-18, hnsw.rs:755
    float s = -0.0f;
    for (size_t i = 0; i < n; ++i) { float p = a[i] * a[i]; s = s + p; }
    return s;
}
inline float cosine_distance(const float* a, const float* b, size_t n) {   // metric.rs:15-21
    float dot = dot_seq(a, b, n);
    float na = std::sqrt(sqnorm_seq(a, n));
    float nb = std::sqrt(sqnorm_seq(b, n));
    float den = na * nb;
    den = den + 1e-10f;
    float q = dot / den;
    return 1.0f - q;
}
inline float euclidean_distance(const float* a, const float* b, size_t n) { // metric.rs:30-35
    float s = -0.0f;
    for (size_t i = 0; i < n; ++i) { float d = a[i] - b[i]; float p = d * d; s = s + p; }
    return std::sqrt(s);
}
// metric.rs:4-12.  The reference panics on a length mismatch; callers here check lengths first.
inline float score(const float* a, const float* b, size_t n, Metric m) {
    switch (m) {
        case Cosine: return cosine_distance(a, b, n);
        case Dot: return dot_seq(a, b, n);
        default: return euclidean_distance(a, b, n);
    }
}

// ---------------------------------------------------------------------------------------------
// Payload model — src/utils/payload.rs
// ---------------------------------------------------------------------------------------------
// OrderedFloat<f64> (ordered-float 4.6.0): total order, NaN == NaN, NaN greatest, -0.0 == 0.0.
inline int ordered_f64_cmp(double a, double b) {
    bool an = std::isnan(a), bn = std::isnan(b);
    if (an || bn) return an ? (bn ? 0 : 1) : -1;
    return a < b ? -1 : (a > b ? 1 : 0);
}
struct Value {                                                         // payload.rs:8-18
    enum Type : int { Int = 0, Float = 1, Str = 2, Bool = 3, ListInt = 4, ListFloat = 5, ListStr = 6, ListBool = 7 };
    Type t = Int;
    int64_t i = 0; double f = 0; std::string s; bool b = false;
    std::vector<int64_t> li; std::vector<double> lf; std::vector<std::string> ls; std::vector<uint8_t> lb;

    static Value mk_int(int64_t v) { Value x; x.t = Int; x.i = v; return x; }
    static Value mk_float(double v) { Value x; x.t = Float; x.f = v; return x; }
    static Value mk_str(std::string v) { Value x; x.t = Str; x.s = std::move(v); return x; }
    static Value mk_bool(bool v) { Value x; x.t = Bool; x.b = v; return x; }
    bool is_scalar() const { return t <= Bool; }

    // derived PartialEq/Eq (payload.rs:8)
    bool operator==(const Value& o) const {
        if (t != o.t) return false;
        switch (t) {
            case Int: return i == o.i;
            case Float: return ordered_f64_cmp(f, o.f) == 0;
            case Str: return s == o.s;
            case Bool: return b == o.b;
            case ListInt: return li == o.li;
            case ListFloat:
                if (lf.size() != o.lf.size()) return false;
                for (size_t k = 0; k < lf.size(); ++k) if (ordered_f64_cmp(lf[k], o.lf[k]) != 0) return false;
                return true;
            case ListStr: return ls == o.ls;
            default: return lb == o.lb;
        }
    }
    bool operator!=(const Value& o) const { return !(*this == o); }
};
// Strict weak order used only as a map key for the inverted index (stands in for Hash+Eq).
struct ValueLess {
    bool operator()(const Value& a, const Value& b) const {
        if (a.t != b.t) return a.t < b.t;
        switch (a.t) {
            case Value::Int: return a.i < b.i;
            case Value::Float: return ordered_f64_cmp(a.f, b.f) < 0;
            case Value::Str: return a.s < b.s;
            case Value::Bool: return a.b < b.b;
            default: return false;   // lists are never indexed (stores.rs:63-71)
        }
    }
};
enum CmpOp : int { Eq = 0, Neq = 1, Lt = 2, Lte = 3, Gt = 4, Gte = 5 };   // payload.rs:131-139

template <class T> inline bool apply_op(CmpOp op, const T& a, const T& b) {
    switch (op) {
        case Eq: return a == b; case Neq: return a != b; case Lt: return a < b;
        case Lte: return a <= b; case Gt: return a > b; default: return a >= b;
    }
}
inline bool apply_op_f64(CmpOp op, double a, double b) {
    int c = ordered_f64_cmp(a, b);
    switch (op) {
        case Eq: return c == 0; case Neq: return c != 0; case Lt: return c < 0;
        case Lte: return c <= 0; case Gt: return c > 0; default: return c >= 0;
    }
}
// payload.rs:150-186.  nullopt == None (type mismatch / unsupported op).
inline std::optional<bool> compare_scalar(const Value& a, CmpOp op, const Value& b) {
    if (a.t == Value::Int && b.t == Value::Int) return apply_op(op, a.i, b.i);
    if (a.t == Value::Float && b.t == Value::Float) return apply_op_f64(op, a.f, b.f);
    if (a.t == Value::Str && b.t == Value::Str) return apply_op(op, a.s, b.s);   // byte-wise like Rust String
    if (a.t == Value::Bool && b.t == Value::Bool) {
        if (op == Eq) return a.b == b.b;
        if (op == Neq) return a.b != b.b;
        return std::nullopt;
    }
    return std::nullopt;
}
// payload.rs:141-147
struct ListQuery {
    enum Kind : int { Contains = 0, Equals = 1, Length = 2, ElementCompare = 3 } kind;
    const Value* val = nullptr; CmpOp op = Eq; size_t n = 0;   // n = length or element index
};
inline bool compare_len(size_t actual, CmpOp op, size_t expected) { return apply_op(op, actual, expected); }
// payload.rs:188-229
inline std::optional<bool> evaluate_list_query(const Value& self, const ListQuery& q) {
    switch (q.kind) {
        case ListQuery::Contains: {
            const Value& v = *q.val;
            if (self.t == Value::ListInt && v.t == Value::Int)
                return std::find(self.li.begin(), self.li.end(), v.i) != self.li.end();
            if (self.t == Value::ListFloat && v.t == Value::Float) {
                for (double x : self.lf) if (ordered_f64_cmp(x, v.f) == 0) return true;
                return false;
            }
            if (self.t == Value::ListStr && v.t == Value::Str)
                return std::find(self.ls.begin(), self.ls.end(), v.s) != self.ls.end();
            if (self.t == Value::ListBool && v.t == Value::Bool) {
                for (uint8_t x : self.lb) if ((x != 0) == v.b) return true;
                return false;
            }
            return std::nullopt;
        }
        case ListQuery::Equals:
            if (self.t == q.val->t) return self == *q.val;
            return std::nullopt;
        case ListQuery::Length:
            switch (self.t) {
                case Value::ListInt: return compare_len(self.li.size(), q.op, q.n);
                case Value::ListFloat: return compare_len(self.lf.size(), q.op, q.n);
                case Value::ListStr: return compare_len(self.ls.size(), q.op, q.n);
                case Value::ListBool: return compare_len(self.lb.size(), q.op, q.n);
                default: return std::nullopt;
            }
        default: {   // ElementCompare
            const Value& v = *q.val;
            if (self.t == Value::ListInt && v.t == Value::Int) {
                if (q.n >= self.li.size()) return std::nullopt;
                return apply_op(q.op, self.li[q.n], v.i);
            }
            if (self.t == Value::ListFloat && v.t == Value::Float) {
                if (q.n >= self.lf.size()) return std::nullopt;
                return apply_op_f64(q.op, self.lf[q.n], v.f);
            }
            if (self.t == Value::ListStr && v.t == Value::Str) {
                if (q.n >= self.ls.size()) return std::nullopt;
                return apply_op(q.op, self.ls[q.n], v.s);
            }
            if (self.t == Value::ListBool && v.t == Value::Bool) {
                if (q.op != Eq && q.op != Neq) return std::nullopt;
                if (q.n >= self.lb.size()) return std::nullopt;
                // PartialOrd on bool is never reached for other ops (payload.rs:219-225)
                return apply_op(q.op, self.lb[q.n] != 0, v.b);
            }
            return std::nullopt;
        }
    }
}

template <class T> struct Result {
    std::optional<T> v; DBError e{Ok, ""};
    bool ok() const { return v.has_value(); }
    static Result good(T x) { Result r; r.v = std::move(x); return r; }
    static Result bad(DBError e) { Result r; r.e = std::move(e); return r; }
};

// payload.rs:21.  std::map => iteration in ascending byte order of the key, which is this
// oracle's deterministic stand-in for HashMap iteration order (DET-3).
struct Payload {
    std::map<std::string, Value> m;
    void set(const std::string& k, Value v) { m[k] = std::move(v); }            // payload.rs:27-29
    const Value* get(const std::string& k) const {                                // payload.rs:32-34
        auto it = m.find(k); return it == m.end() ? nullptr : &it->second;
    }
    // payload.rs:36-114
    Result<bool> compare_field(const std::string& field, CmpOp op, const Value& other) const {
        const Value* value = get(field);
        if (!value) return Result<bool>::bad(err_invalid_payload("Missing field: " + field));
        if (value->t == Value::ListStr && other.t == Value::ListStr) {
            if (op == Eq) return Result<bool>::good(value->ls == other.ls);
            if (op == Neq) return Result<bool>::good(value->ls != other.ls);
            return Result<bool>::bad(err_invalid_payload("Invalid operation for ListStr"));
        }
        if (value->t == Value::ListStr && other.t == Value::Str) {
            bool c = std::find(value->ls.begin(), value->ls.end(), other.s) != value->ls.end();
            if (op == Eq) return Result<bool>::good(c);
            if (op == Neq) return Result<bool>::good(!c);
            return Result<bool>::bad(err_invalid_payload("Invalid operation for ListStr and Str"));
        }
        auto r = compare_scalar(*value, op, other);
        if (r) return Result<bool>::good(*r);
        return Result<bool>::bad(err_invalid_payload("Type mismatch for field: " + field));
    }
    // payload.rs:117-128
    Result<bool> evaluate_list_field(const std::string& field, const ListQuery& q) const {
        const Value* value = get(field);
        if (!value) return Result<bool>::bad(err_invalid_payload("Missing field: " + field));
        auto r = evaluate_list_query(*value, q);
        if (r) return Result<bool>::good(*r);
        return Result<bool>::bad(err_invalid_payload("Invalid list operation on field: " + field));
    }
};

// ---------------------------------------------------------------------------------------------
// Filters — src/payload_storage/filters.rs
// ---------------------------------------------------------------------------------------------
struct Filter {                                                        // filters.rs:5-19
    enum Kind : int { Match = 0, Compare = 1, And = 2, Or = 3, Not = 4 } kind = Match;
    std::string key; CmpOp op = Eq; Value value;
    std::vector<Filter> children;   // And/Or: n children; Not: exactly one
};
// filters.rs:23-75
inline Result<bool> evaluate_filter(const Filter& f, const Payload& p) {
    switch (f.kind) {
        case Filter::Match: {
            const Value* actual = p.get(f.key);
            return Result<bool>::good(actual ? (*actual == f.value) : false);
        }
        case Filter::Compare: return p.compare_field(f.key, f.op, f.value);
        case Filter::And:
            for (const Filter& c : f.children) {
                auto r = evaluate_filter(c, p);
                if (!r.ok()) return r;
                if (!*r.v) return Result<bool>::good(false);
            }
            return Result<bool>::good(true);
        case Filter::Or:
            for (const Filter& c : f.children) {
                auto r = evaluate_filter(c, p);
                if (!r.ok()) return r;
                if (*r.v) return Result<bool>::good(true);
            }
            return Result<bool>::good(false);
        default: {
            auto r = evaluate_filter(f.children.at(0), p);
            if (!r.ok()) return r;
            return Result<bool>::good(!*r.v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Inverted payload index — src/payload_storage/stores.rs
// ---------------------------------------------------------------------------------------------
struct PayloadIndex {
    // std::set => ascending id iteration (DET-4, DET-2).
    std::map<std::string, std::map<Value, std::set<PointId>, ValueLess>> index;   // stores.rs:6-8
    static bool is_indexable(const Value& v) { return v.is_scalar(); }            // stores.rs:63-71
    void insert(PointId id, const Payload& p) {                                   // stores.rs:18-31
        for (auto& kv : p.m) {
            if (!is_indexable(kv.second)) continue;
            index[kv.first][kv.second].insert(id);
        }
    }
    void remove(PointId id, const Payload& p) {                                   // stores.rs:34-52
        for (auto& kv : p.m) {
            if (!is_indexable(kv.second)) continue;
            auto kit = index.find(kv.first);
            if (kit == index.end()) continue;
            auto vit = kit->second.find(kv.second);
            if (vit != kit->second.end()) {
                vit->second.erase(id);
                if (vit->second.empty()) kit->second.erase(vit);
            }
            if (kit->second.empty()) index.erase(kit);
        }
    }
    const std::set<PointId>* query_exact(const std::string& key, const Value& v) const {   // stores.rs:55-61
        if (!is_indexable(v)) return nullptr;
        auto kit = index.find(key);
        if (kit == index.end()) return nullptr;
        auto vit = kit->second.find(v);
        return vit == kit->second.end() ? nullptr : &vit->second;
    }
    std::optional<std::set<PointId>> all_for_key(const std::string& key) const {  // stores.rs:74-82
        auto kit = index.find(key);
        if (kit == index.end()) return std::nullopt;
        std::set<PointId> acc;
        for (auto& vv : kit->second) acc.insert(vv.second.begin(), vv.second.end());
        return acc;
    }
};

// ---------------------------------------------------------------------------------------------
// Rust std BinaryHeap, restated.  Ord3(a, b) returns the PartialOrd result of a vs b:
// -1 Less, 0 Equal, 1 Greater, 2 None (NaN).  `<=`, `>=`, `<` follow PartialOrd's defaults.
// ---------------------------------------------------------------------------------------------
template <class T, class Ord3> struct RustBinaryHeap {
    std::vector<T> d;
    static bool le(const T& a, const T& b) { int c = Ord3()(a, b); return c == -1 || c == 0; }
    static bool ge(const T& a, const T& b) { int c = Ord3()(a, b); return c == 1 || c == 0; }
    static bool lt(const T& a, const T& b) { return Ord3()(a, b) == -1; }
    size_t len() const { return d.size(); }
    bool empty() const { return d.empty(); }
    const T* peek() const { return d.empty() ? nullptr : &d[0]; }
    void push(const T& x) { size_t old = d.size(); d.push_back(x); sift_up(0, old); }
    bool pop(T& out) {
        if (d.empty()) return false;
        T item = d.back(); d.pop_back();
        if (!d.empty()) { std::swap(item, d[0]); sift_down_to_bottom(0); }
        out = item; return true;
    }
    size_t sift_up(size_t start, size_t pos) {
        T elt = d[pos];
        while (pos > start) {
            size_t parent = (pos - 1) / 2;
            if (le(elt, d[parent])) break;
            d[pos] = d[parent]; pos = parent;
        }
        d[pos] = elt; return pos;
    }
    void sift_down_range(size_t pos, size_t end) {
        T elt = d[pos];
        size_t child = 2 * pos + 1;
        size_t lim = end >= 2 ? end - 2 : 0;
        while (child <= lim) {
            child += le(d[child], d[child + 1]) ? 1 : 0;
            if (ge(elt, d[child])) { d[pos] = elt; return; }
            d[pos] = d[child]; pos = child; child = 2 * pos + 1;
        }
        if (child == end - 1 && lt(elt, d[child])) { d[pos] = d[child]; pos = child; }
        d[pos] = elt;
    }
    void sift_down_to_bottom(size_t pos) {
        size_t end = d.size(), start = pos;
        T elt = d[pos];
        size_t child = 2 * pos + 1;
        size_t lim = end >= 2 ? end - 2 : 0;
        while (child <= lim) {
            child += le(d[child], d[child + 1]) ? 1 : 0;
            d[pos] = d[child]; pos = child; child = 2 * pos + 1;
        }
        if (child == end - 1) { d[pos] = d[child]; pos = child; }
        d[pos] = elt;
        sift_up(start, pos);
    }
    std::vector<T> into_sorted_vec() {
        size_t end = d.size();
        while (end > 1) { end -= 1; std::swap(d[0], d[end]); sift_down_range(0, end); }
        return std::move(d);
    }
    std::vector<T> into_vec() { return std::move(d); }
};

// ---------------------------------------------------------------------------------------------
// HNSW — src/vector/hnsw.rs
// ---------------------------------------------------------------------------------------------
struct ScoredPoint { PointId id; float raw_score; float sort_key; };           // hnsw.rs:12-16
inline int f32_partial_cmp(float a, float b) {
    if (a < b) return -1;
    if (a > b) return 1;
    if (a == b) return 0;
    return 2;
}
// Internal records carry the dense slot; ids are translated at the API edge.
struct SP { uint32_t slot; float raw; float key; };
struct CandOrd { int operator()(const SP& a, const SP& b) const { return f32_partial_cmp(b.key, a.key); } };  // hnsw.rs:27-32 (inverted)
struct ResOrd { int operator()(const SP& a, const SP& b) const { return f32_partial_cmp(a.key, b.key); } };   // hnsw.rs:46-51

struct SearchStats {          // per-call work counters (not in the reference; used for algorithmic bytes)
    uint64_t rows_scored = 0;     // neighbour/entry rows whose distance was computed
    uint64_t expanded = 0;        // adjacency lists read
    uint64_t adj_entries = 0;     // total ids in those lists
    void add(const SearchStats& o) { rows_scored += o.rows_scored; expanded += o.expanded; adj_entries += o.adj_entries; }
};
struct Scratch {              // per-thread visited stamps (stands in for the per-call HashSet, hnsw.rs:435)
    std::vector<uint32_t> stamp; uint32_t epoch = 0;
    void begin(size_t n) {
        if (stamp.size() < n) stamp.resize(n, 0);
        if (++epoch == 0) { std::fill(stamp.begin(), stamp.end(), 0); epoch = 1; }
    }
    bool insert(uint32_t s) { if (stamp[s] == epoch) return false; stamp[s] = epoch; return true; }
};

inline void sort_by_key_stable(std::vector<SP>& v) {       // hnsw.rs:506 / :550 (panics on NaN)
    for (auto& x : v) if (std::isnan(x.key)) throw PanicError("called `Option::unwrap()` on a `None` value");
    std::stable_sort(v.begin(), v.end(), [](const SP& a, const SP& b) { return a.key < b.key; });
}

enum FilterMode : int { Literal = 0, Masked = 1 };

class Payloads;   // fwd

struct HNSWIndex {
    // hnsw.rs:59-73
    Metric metric; size_t m, ef, max_level_cap, dim; double level_scale;
    size_t current_max_level = 0; std::optional<PointId> entry_point;
    // dense storage replacing the HashMaps
    std::unordered_map<PointId, uint32_t> slot_of;
    std::vector<PointId> ids; std::vector<float> vecs; std::vector<size_t> levels;
    struct Layer { std::vector<int32_t> idx; std::vector<std::vector<uint32_t>> lists; std::vector<uint32_t> owner; };
    std::vector<Layer> layers;          // layers[l].idx[slot] = list index or -1 (absent)
    std::vector<uint8_t> deleted; size_t n_deleted = 0;
    std::set<PointId> live;             // DET-1: "first non-deleted key" == smallest live id
    std::set<PointId> pending_deleted;  // ids marked deleted before they were inserted (hnsw.rs:112 takes any id)
    uint64_t seed = 0, level_draws = 0, sample_draws = 0;
    static Scratch& tls_scratch() { static thread_local Scratch sc; return sc; }

    HNSWIndex(Metric metric_, size_t m_, size_t ef_, size_t cap_, size_t dim_, uint64_t seed_ = 0)   // hnsw.rs:77-94
        : metric(metric_), m(m_), ef(ef_), max_level_cap(cap_), dim(dim_),
          level_scale(1.0 / std::log((double)m_)), seed(seed_) {}

    size_t len() const { return ids.size(); }                                    // hnsw.rs:695
    bool contains(PointId id) const { return slot_of.count(id) != 0; }           // hnsw.rs:691
    const float* vec(uint32_t s) const { return &vecs[(size_t)s * dim]; }
    bool is_deleted_slot(uint32_t s) const { return deleted[s] != 0; }
    float normalize_score(float raw) const { return metric == Dot ? -raw : raw; } // hnsw.rs:103-108

    // hnsw.rs:96-101 with the thread RNG replaced by the counter RNG.
    size_t assign_random_level() {
        uint64_t h = rng_draw(seed, STREAM_LEVEL, level_draws++);
        double r = (double)(h >> 11) * (1.0 / 9007199254740992.0);
        double x = std::floor(-std::log(r) * level_scale);
        size_t l;
        if (!(x == x)) l = 0;                       // NaN as usize == 0
        else if (x >= 1.8446744073709552e19) l = SIZE_MAX;   // saturating float->usize cast
        else if (x <= 0) l = 0; else l = (size_t)x;
        return std::min(l, max_level_cap);
    }
    // hnsw.rs:752-764
    std::vector<float> maybe_normalize(const float* v) const {
        std::vector<float> out(v, v + dim);
        if (metric != Cosine) return out;
        float norm = std::sqrt(sqnorm_seq(v, dim));
        if (norm == 0.0f) return out;
        for (size_t i = 0; i < dim; ++i) out[i] = v[i] / norm;
        return out;
    }

    Layer& layer_mut(size_t l) { if (layers.size() <= l) layers.resize(l + 1); return layers[l]; }
    const std::vector<uint32_t>* list(size_t l, uint32_t s) const {              // layers.get(&l)?.get(&id)
        if (l >= layers.size()) return nullptr;
        const Layer& L = layers[l];
        if (s >= L.idx.size() || L.idx[s] < 0) return nullptr;
        return &L.lists[L.idx[s]];
    }
    std::vector<uint32_t>& list_or_default(size_t l, uint32_t s) {               // entry(..).or_default()
        Layer& L = layer_mut(l);
        if (L.idx.size() <= s) L.idx.resize(std::max<size_t>(s + 1, ids.size()), -1);
        if (L.idx[s] < 0) { L.idx[s] = (int32_t)L.lists.size(); L.lists.emplace_back(); L.owner.push_back(s); }
        return L.lists[L.idx[s]];
    }
    std::optional<uint32_t> first_live_slot() const {                            // DET-1
        if (live.empty()) return std::nullopt;
        return slot_of.at(*live.begin());
    }

    // hnsw.rs:111-117
    void mark_deleted(PointId id) {
        auto it = slot_of.find(id);
        // HashSet::insert accepts unknown ids too (hnsw.rs:112): a point inserted under such an id later is deleted from birth.
        if (it == slot_of.end()) pending_deleted.insert(id);
        if (it != slot_of.end() && !deleted[it->second]) { deleted[it->second] = 1; ++n_deleted; live.erase(id); }
        if (entry_point && *entry_point == id) {
            if (live.empty()) entry_point = std::nullopt; else entry_point = *live.begin();
        }
    }

    // hnsw.rs:333-368
    uint32_t greedy_search_layer_unfiltered(const float* q, uint32_t entry, size_t level, SearchStats* st = nullptr) const {
        uint32_t current = entry; bool changed = true; int steps = 0;
        while (changed && steps < 1000) {
            steps += 1; changed = false;
            const std::vector<uint32_t>* nb = list(level, current);
            if (!nb) continue;
            if (st) { st->expanded++; st->adj_entries += nb->size(); }
            for (uint32_t n : *nb) {
                if (deleted[n]) continue;
                float d_current = score(q, vec(current), dim, metric);
                float d_new = score(q, vec(n), dim, metric);
                if (st) st->rows_scored++;
                float s_current = normalize_score(d_current), s_new = normalize_score(d_new);
                if (s_new < s_current) { current = n; changed = true; break; }
            }
        }
        return current;
    }

    // hnsw.rs:420-511.  Returns records in slots; `stats` counts work.
    Result<std::vector<SP>> search_layer_unfiltered(const float* q, size_t qlen, uint32_t entry, size_t level,
                                                    size_t ef_, bool normalize, SearchStats* st = nullptr) const {
        if (qlen != dim) return Result<std::vector<SP>>::bad(err_len_mismatch(dim, qlen));
        Scratch& scratch = tls_scratch();
        scratch.begin(ids.size());
        RustBinaryHeap<SP, CandOrd> cand; RustBinaryHeap<SP, ResOrd> res;
        uint32_t start = entry;
        if (deleted[entry]) { auto f = first_live_slot(); if (f) start = *f; }      // hnsw.rs:440-444 (DET-1)
        float ed = score(q, vec(start), dim, metric);
        if (st) st->rows_scored++;
        float es = normalize ? normalize_score(ed) : ed;
        SP initial{start, ed, es};
        cand.push(initial); res.push(initial); scratch.insert(start);
        float worst = res.peek()->key;
        while (const SP* top = cand.peek()) {
            if (top->key > worst) break;
            SP cur; cand.pop(cur);
            const std::vector<uint32_t>* nb = list(level, cur.slot);
            if (!nb) continue;
            if (st) { st->expanded++; st->adj_entries += nb->size(); }
            for (uint32_t n : *nb) {
                if (deleted[n] || !scratch.insert(n)) continue;
                float raw = score(q, vec(n), dim, metric);
                if (st) st->rows_scored++;
                float sv = normalize ? normalize_score(raw) : raw;
                if (res.len() < ef_ || sv < worst) {
                    SP sp{n, raw, sv};
                    cand.push(sp); res.push(sp);
                    if (res.len() > ef_) { SP drop; res.pop(drop); }
                    if (const SP* rp = res.peek()) worst = rp->key;
                }
            }
        }
        std::vector<SP> out = res.into_vec();
        sort_by_key_stable(out);
        return Result<std::vector<SP>>::good(std::move(out));
    }

    // hnsw.rs:119-208
    Result<bool> insert(PointId id, const float* v, size_t vlen) {
        if (contains(id)) return Result<bool>::good(true);                       // :122-125
        if (vlen != dim) return Result<bool>::bad(err_len_mismatch(dim, vlen));  // :127-133
        size_t level = assign_random_level();
        std::vector<float> nv = maybe_normalize(v);
        uint32_t s = (uint32_t)ids.size();
        ids.push_back(id); slot_of[id] = s; vecs.insert(vecs.end(), nv.begin(), nv.end());
        levels.push_back(level);
        if (pending_deleted.erase(id)) { deleted.push_back(1); ++n_deleted; } else { deleted.push_back(0); live.insert(id); }
        for (size_t l = 0; l <= level; ++l) list_or_default(l, s).push_back(s);  // :143-148
        if (!entry_point) { entry_point = id; current_max_level = level; return Result<bool>::good(true); }
        uint32_t ep = slot_of.at(*entry_point);
        uint32_t current_entry = ep;
        if (deleted[ep]) { auto f = first_live_slot(); current_entry = f ? *f : ep; }   // :158-164 (DET-1)
        const float* pv = vec(s);
        for (size_t l = current_max_level; l >= level + 1 && l <= current_max_level; --l) {
            current_entry = greedy_search_layer_unfiltered(pv, current_entry, l);
            if (l == 0) break;
        }
        for (size_t l = level + 1; l-- > 0;) {
            bool use_norm = metric == Cosine || metric == Dot;
            auto r = search_layer_unfiltered(pv, dim, current_entry, l, ef, use_norm);
            if (!r.ok()) return Result<bool>::bad(r.e);
            std::vector<uint32_t> neighbors;
            for (size_t k = 0; k < r.v->size() && k < m; ++k) neighbors.push_back((*r.v)[k].slot);
            std::vector<uint32_t> linked = neighbors;
            if (std::find(linked.begin(), linked.end(), s) == linked.end()) linked.push_back(s);
            list_or_default(l, s) = linked;                                       // layer.insert(point_id, linked)
            for (uint32_t n : neighbors) {
                std::vector<uint32_t>& e = list_or_default(l, n);
                if (std::find(e.begin(), e.end(), s) == e.end()) e.push_back(s);
            }
            if (!neighbors.empty()) current_entry = neighbors[0];
        }
        if (level > current_max_level) { entry_point = id; current_max_level = level; }
        return Result<bool>::good(true);
    }

    void add_bidirectional_edge(size_t level, uint32_t a, uint32_t b) {           // hnsw.rs:328-331
        list_or_default(level, a).push_back(b);
        list_or_default(level, b).push_back(a);
    }
    const float* get_vector_slot(uint32_t s) const { return deleted[s] ? nullptr : vec(s); }   // hnsw.rs:727-734

    // hnsw.rs:210-325
    Result<bool> build_filter_aware_edges(PointId point_id, const float* vector, const Payload& payload,
                                          const PayloadIndex& pidx, const std::unordered_map<PointId, Payload>& payloads,
                                          const std::vector<std::string>& filter_keys) {
        std::vector<float> qv = metric == Cosine ? maybe_normalize(vector) : std::vector<float>(vector, vector + dim);
        std::set<PointId> extra;   // DET-5: iterate extra_neighbors in ascending id
        uint32_t self_slot = slot_of.at(point_id);
        auto sort_raw = [&](std::vector<SP>& v) {
            for (auto& x : v) if (std::isnan(x.raw)) throw PanicError("called `Option::unwrap()` on a `None` value");
            if (metric == Dot) std::stable_sort(v.begin(), v.end(), [](const SP& a, const SP& b) { return a.raw > b.raw; });
            else std::stable_sort(v.begin(), v.end(), [](const SP& a, const SP& b) { return a.raw < b.raw; });
        };
        for (const std::string& key : filter_keys) {
            const Value* value = payload.get(key);
            if (!value) continue;
            if (const std::set<PointId>* id_set = pidx.query_exact(key, *value)) {
                // choose_multiple(&mut rng, 100) == reservoir sampling (rand 0.9 IteratorRandom); DET-2: ascending id order.
                std::vector<uint32_t> reservoir; size_t i_after = 0; const size_t amount = 100;
                for (PointId pid : *id_set) {
                    if (pid == point_id) continue;
                    auto it = slot_of.find(pid);
                    if (it == slot_of.end() || deleted[it->second]) continue;      // get_vector(&id).is_some()
                    if (reservoir.size() < amount) { reservoir.push_back(it->second); continue; }
                    uint64_t k = rng_draw(seed, STREAM_SAMPLE, sample_draws++) % (i_after + 1 + amount);
                    if (k < amount) reservoir[k] = it->second;
                    ++i_after;
                }
                std::vector<SP> scored;
                for (uint32_t s : reservoir) {
                    float raw = score(qv.data(), vec(s), dim, metric);
                    scored.push_back({s, raw, normalize_score(raw)});
                }
                sort_raw(scored);
                for (size_t k = 0; k < scored.size() && k < m; ++k) extra.insert(ids[scored[k].slot]);
                if (extra.size() >= m) break;
            }
            std::vector<SP> candidates;
            if (current_max_level > 0) {
                uint32_t entry = slot_of.at(*entry_point);     // get_entry_point().unwrap()
                for (size_t l = current_max_level; l >= 1; --l) entry = greedy_search_layer_unfiltered(qv.data(), entry, l);
                auto r = search_layer_unfiltered(qv.data(), dim, entry, 0, ef, metric == Cosine || metric == Dot);
                if (!r.ok()) return Result<bool>::bad(r.e);
                candidates = std::move(*r.v);
            } else {
                // iter_vectors() order is HashMap order in the reference; DET-6: ascending id.
                for (PointId lid : live) {
                    if (lid == point_id) continue;
                    uint32_t s = slot_of.at(lid);
                    float raw = score(qv.data(), vec(s), dim, metric);
                    candidates.push_back({s, raw, normalize_score(raw)});
                }
            }
            sort_raw(candidates);
            size_t taken = 0;
            for (const SP& sp : candidates) {
                if (taken >= m) break;
                auto pit = payloads.find(ids[sp.slot]);
                if (pit == payloads.end()) continue;
                const Value* v = pit->second.get(key);
                if (!v || !(*v == *value)) continue;
                extra.insert(ids[sp.slot]); ++taken;
            }
            if (extra.size() >= m) break;
        }
        extra.insert(point_id);
        for (PointId nid : extra) add_bidirectional_edge(0, self_slot, slot_of.at(nid));
        return Result<bool>::good(true);
    }

    // hnsw.rs:513-554.  `ef_override` == 0 uses the index ef (the reference has no override).
    Result<std::vector<ScoredPoint>> search(const float* query, size_t qlen, size_t top_k, size_t ef_override = 0,
                                            SearchStats* st = nullptr) const {
        if (!entry_point) return Result<std::vector<ScoredPoint>>::good({});
        if (qlen != dim) return Result<std::vector<ScoredPoint>>::bad(err_len_mismatch(dim, qlen));
        bool normalize_query = metric == Cosine;
        bool normalize_flag = metric != Euclidean;
        std::vector<float> qn = normalize_query ? maybe_normalize(query) : std::vector<float>(query, query + dim);
        uint32_t current = slot_of.at(*entry_point);
        for (size_t l = current_max_level; l >= 1; --l) current = greedy_search_layer_unfiltered(qn.data(), current, l, st);
        auto r = search_layer_unfiltered(qn.data(), dim, current, 0, ef_override ? ef_override : ef, normalize_flag, st);
        if (!r.ok()) return Result<std::vector<ScoredPoint>>::bad(r.e);
        std::vector<SP>& res = *r.v;
        sort_by_key_stable(res);
        if (res.size() > top_k) res.resize(top_k);
        std::vector<ScoredPoint> out;
        for (auto& x : res) out.push_back({ids[x.slot], x.raw, x.key});
        return Result<std::vector<ScoredPoint>>::good(std::move(out));
    }
};

// hnsw.rs:558-583 / :370-418 / :585-689 need payloads; they live here as free functions.
using PayloadMap = std::unordered_map<PointId, Payload>;

inline std::optional<PointId> find_entry_point_matching_filter(const HNSWIndex& h, const Filter& f, const PayloadIndex& pidx) {
    switch (f.kind) {
        case Filter::Match: {
            const std::set<PointId>* s = pidx.query_exact(f.key, f.value);
            if (!s) return std::nullopt;
            for (PointId id : *s) {                           // DET-4: ascending id
                auto it = h.slot_of.find(id);
                if (it != h.slot_of.end() && !h.deleted[it->second]) return id;
            }
            return std::nullopt;
        }
        case Filter::And: case Filter::Or:
            for (const Filter& c : f.children) if (auto r = find_entry_point_matching_filter(h, c, pidx)) return r;
            return std::nullopt;
        case Filter::Not: return find_entry_point_matching_filter(h, f.children.at(0), pidx);
        default: return std::nullopt;
    }
}

// hnsw.rs:370-418
inline Result<uint32_t> greedy_search_layer_with_filter(const HNSWIndex& h, const float* q, uint32_t entry, size_t level,
                                                        const PayloadMap& payloads, const Filter* filter, SearchStats* st) {
    uint32_t current = entry; bool changed = true;
    while (changed) {
        changed = false;
        const std::vector<uint32_t>* nb = h.list(level, current);
        if (!nb) continue;
        if (st) { st->expanded++; st->adj_entries += nb->size(); }
        for (uint32_t n : *nb) {
            if (h.deleted[n]) continue;
            if (filter) {
                auto pit = payloads.find(h.ids[n]);
                if (pit == payloads.end()) continue;
                auto r = evaluate_filter(*filter, pit->second);
                if (!r.ok()) return Result<uint32_t>::bad(r.e);
                if (!*r.v) continue;
            }
            if (h.deleted[current]) throw PanicError("called `Option::unwrap()` on a `None` value");   // get_vector(current).unwrap()
            float d_current = score(q, h.vec(current), h.dim, h.metric);
            float d_new = score(q, h.vec(n), h.dim, h.metric);
            if (st) st->rows_scored++;
            float s_current = h.metric == Dot ? -d_current : d_current;
            float s_new = h.metric == Dot ? -d_new : d_new;
            if (s_new < s_current) { current = n; changed = true; }
        }
    }
    return Result<uint32_t>::good(current);
}

// hnsw.rs:585-689 (Literal) and the masked-beam extension (Masked): the unfiltered ef-beam drives
// the traversal exactly as search_layer_unfiltered does while a second bounded max-heap keeps the
// ef best filter-passing nodes among everything scored.
inline Result<std::vector<ScoredPoint>> in_place_filtered_search(const HNSWIndex& h, const float* query, size_t qlen, size_t top_k,
                                                                const PayloadMap& payloads, const PayloadIndex& pidx,
                                                                const Filter* filter, FilterMode mode = Literal,
                                                                size_t ef_override = 0, SearchStats* st = nullptr) {
    using R = Result<std::vector<ScoredPoint>>;
    if (qlen != h.dim) return R::bad(err_len_mismatch(h.dim, qlen));
    size_t ef = ef_override ? ef_override : h.ef;
    auto passes_lossy = [&](uint32_t s) -> bool {            // hnsw.rs:668-671
        if (!filter) return true;
        auto pit = payloads.find(h.ids[s]);
        if (pit == payloads.end()) return false;
        auto r = evaluate_filter(*filter, pit->second);
        return r.ok() ? *r.v : false;
    };
    if (mode == Masked) {
        if (!h.entry_point) return R::good({});
        std::vector<float> qn = h.metric == Cosine ? h.maybe_normalize(query) : std::vector<float>(query, query + h.dim);
        const float* q = qn.data();
        uint32_t current = h.slot_of.at(*h.entry_point);
        for (size_t l = h.current_max_level; l >= 1; --l) current = h.greedy_search_layer_unfiltered(q, current, l, st);
        Scratch& scratch = HNSWIndex::tls_scratch();
        scratch.begin(h.ids.size());
        RustBinaryHeap<SP, CandOrd> cand; RustBinaryHeap<SP, ResOrd> res, pass;
        uint32_t start = current;
        if (h.deleted[current]) { auto f = h.first_live_slot(); if (f) start = *f; }
        float ed = score(q, h.vec(start), h.dim, h.metric); if (st) st->rows_scored++;
        SP initial{start, ed, h.normalize_score(ed)};
        cand.push(initial); res.push(initial); scratch.insert(start);
        if (!h.deleted[start] && passes_lossy(start)) pass.push(initial);
        float worst = res.peek()->key;
        while (const SP* top = cand.peek()) {
            if (top->key > worst) break;
            SP cur; cand.pop(cur);
            const std::vector<uint32_t>* nb = h.list(0, cur.slot);
            if (!nb) continue;
            if (st) { st->expanded++; st->adj_entries += nb->size(); }
            for (uint32_t n : *nb) {
                if (h.deleted[n] || !scratch.insert(n)) continue;
                float raw = score(q, h.vec(n), h.dim, h.metric); if (st) st->rows_scored++;
                SP sp{n, raw, h.normalize_score(raw)};
                if (passes_lossy(n)) {
                    if (pass.len() < ef || sp.key < pass.peek()->key) {
                        pass.push(sp);
                        if (pass.len() > ef) { SP drop; pass.pop(drop); }
                    }
                }
                if (res.len() < ef || sp.key < worst) {
                    cand.push(sp); res.push(sp);
                    if (res.len() > ef) { SP drop; res.pop(drop); }
                    worst = res.peek()->key;
                }
            }
        }
        std::vector<SP> out = pass.into_vec();
        sort_by_key_stable(out);
        if (out.size() > top_k) out.resize(top_k);
        std::vector<ScoredPoint> o; for (auto& x : out) o.push_back({h.ids[x.slot], x.raw, x.key});
        return R::good(std::move(o));
    }
    // ---- Literal ----
    uint32_t entry;
    if (h.entry_point) {                                                            // :600-615
        PointId id = *h.entry_point;
        if (filter) {
            auto pit = payloads.find(id);
            bool use_id = false;
            if (pit != payloads.end()) {
                auto r = evaluate_filter(*filter, pit->second);
                if (!r.ok()) return R::bad(r.e);
                use_id = *r.v;
            }
            if (!use_id) { auto alt = find_entry_point_matching_filter(h, *filter, pidx); if (alt) id = *alt; }
        }
        entry = h.slot_of.at(id);
    } else {                                                                        // :616-625
        if (!filter) return R::good({});
        auto alt = find_entry_point_matching_filter(h, *filter, pidx);
        if (!alt) return R::good({});
        entry = h.slot_of.at(*alt);
    }
    for (size_t l = h.current_max_level; l >= 1; --l) {                             // :628-630
        auto r = greedy_search_layer_with_filter(h, query, entry, l, payloads, filter, st);
        if (!r.ok()) return R::bad(r.e);
        entry = *r.v;
    }
    Scratch& scratch = HNSWIndex::tls_scratch();
    scratch.begin(h.ids.size());
    RustBinaryHeap<SP, CandOrd> cand, res;        // both heaps use the inverted ScoredPoint order (:633-634)
    if (h.deleted[entry]) throw PanicError("called `Option::unwrap()` on a `None` value");   // :636 get_vector(&entry).unwrap()
    float dist = score(query, h.vec(entry), h.dim, h.metric); if (st) st->rows_scored++;
    float key0 = h.metric == Dot ? -dist : dist;
    SP first{entry, dist, key0};
    cand.push(first); res.push(first); scratch.insert(entry);
    float worst = key0;
    SP cur;
    while (cand.pop(cur)) {                                                         // :648
        if (cur.key > worst) break;
        const std::vector<uint32_t>* nb = h.list(0, cur.slot);
        if (!nb) continue;
        if (st) { st->expanded++; st->adj_entries += nb->size(); }
        for (uint32_t n : *nb) {
            if (h.deleted[n] || !scratch.insert(n)) continue;
            float d = score(query, h.vec(n), h.dim, h.metric); if (st) st->rows_scored++;
            SP sp{n, d, h.metric == Dot ? -d : d};
            cand.push(sp);
            if (passes_lossy(n)) {
                res.push(sp);
                if (res.len() > ef) { SP drop; res.pop(drop); }
                if (const SP* rp = res.peek()) worst = rp->key;
            }
        }
    }
    std::vector<SP> out = res.into_sorted_vec();                                    // :686 (ascending in the inverted order)
    if (out.size() > top_k) out.resize(top_k);
    std::vector<ScoredPoint> o; for (auto& x : out) o.push_back({h.ids[x.slot], x.raw, x.key});
    return R::good(std::move(o));
}

// ---------------------------------------------------------------------------------------------
// Segment — src/segment/segment.rs
// ---------------------------------------------------------------------------------------------
struct Segment {
    std::unique_ptr<HNSWIndex> hnsw; PayloadIndex payload_index; PayloadMap payloads;
    std::set<PointId> deleted; PointId next_id = 1;                                // segment.rs:21-29
    FilterMode filter_mode = Literal;
    explicit Segment(std::unique_ptr<HNSWIndex> h) : hnsw(std::move(h)) {}

    static std::vector<std::string> scalar_keys(const Payload& p) {                 // segment.rs:40-44 (DET-3)
        std::vector<std::string> k;
        for (auto& kv : p.m) if (kv.second.is_scalar()) k.push_back(kv.first);
        return k;
    }
    Result<PointId> insert(const float* v, size_t vlen, const Payload* payload) {   // segment.rs:32-61
        PointId pid = next_id;
        auto r = hnsw->insert(pid, v, vlen);
        if (!r.ok()) return Result<PointId>::bad(r.e);
        if (payload) {
            payload_index.insert(pid, *payload);
            payloads[pid] = *payload;
            auto keys = scalar_keys(*payload);
            if (!keys.empty()) {
                auto e = hnsw->build_filter_aware_edges(pid, v, *payload, payload_index, payloads, keys);
                if (!e.ok()) return Result<PointId>::bad(e.e);
            }
        }
        next_id += 1;
        return Result<PointId>::good(pid);
    }
    const float* get_vector(PointId id) const {                                    // segment.rs:64-69
        if (deleted.count(id)) return nullptr;
        auto it = hnsw->slot_of.find(id);
        if (it == hnsw->slot_of.end()) return nullptr;
        return hnsw->get_vector_slot(it->second);
    }
    Result<bool> del(PointId id) {                                                  // segment.rs:71-99
        if (deleted.count(id) || !hnsw->contains(id)) return Result<bool>::good(true);
        auto pit = payloads.find(id);
        if (pit != payloads.end()) payload_index.remove(id, pit->second);
        deleted.insert(id);
        hnsw->mark_deleted(id);
        size_t deleted_count = deleted.size(), total = hnsw->len();
        if (deleted_count >= 100 && ((float)deleted_count / (float)total) >= 0.25f) return purge();
        return Result<bool>::good(true);
    }
    size_t live_count() const { return hnsw->len() - deleted.size(); }
    template <class V> std::vector<ScoredPoint> drop_deleted_take(const V& c, size_t top_k) const {
        std::vector<ScoredPoint> out;
        for (auto& sp : c) { if (out.size() >= top_k) break; if (!deleted.count(sp.id)) out.push_back(sp); }
        return out;
    }
    Result<std::vector<ScoredPoint>> search(const float* q, size_t qlen, size_t top_k, size_t ef_override = 0, SearchStats* st = nullptr) const {   // segment.rs:103-119
        using R = Result<std::vector<ScoredPoint>>;
        if (live_count() == 0) return R::bad(err_search("No active points available to search."));
        auto c = hnsw->search(q, qlen, top_k * 2, ef_override, st);
        if (!c.ok()) return c;
        return R::good(drop_deleted_take(*c.v, top_k));
    }
    Result<std::vector<ScoredPoint>> search_with_filter(const float* q, size_t qlen, size_t top_k, const Filter* f,
                                                        size_t ef_override = 0, SearchStats* st = nullptr) const {   // segment.rs:121-147
        using R = Result<std::vector<ScoredPoint>>;
        if (live_count() == 0) return R::bad(err_search("No active points available to search."));
        auto c = in_place_filtered_search(*hnsw, q, qlen, top_k * 2, payloads, payload_index, f, filter_mode, ef_override, st);
        if (!c.ok()) return c;
        return R::good(drop_deleted_take(*c.v, top_k));
    }
    Result<std::vector<ScoredPoint>> search_unfiltered(const float* q, size_t qlen, size_t top_k) const {            // segment.rs:152-154
        return hnsw->search(q, qlen, top_k);
    }
    Result<std::vector<ScoredPoint>> post_filter(const float* q, size_t qlen, size_t top_k, const Filter* f) const { // segment.rs:221-249
        using R = Result<std::vector<ScoredPoint>>;
        if (live_count() == 0) return R::bad(err_search("No active points available to search."));
        auto c = hnsw->search(q, qlen, top_k * 4);
        if (!c.ok()) return c;
        std::vector<ScoredPoint> out;
        for (auto& sp : *c.v) {
            if (out.size() >= top_k) break;
            if (deleted.count(sp.id)) continue;
            if (f) {
                auto pit = payloads.find(sp.id);
                if (pit == payloads.end()) continue;
                auto r = evaluate_filter(*f, pit->second);
                if (!r.ok() || !*r.v) continue;
            }
            out.push_back(sp);
        }
        return R::good(std::move(out));
    }
    Result<bool> purge() {                                                          // segment.rs:166-217
        auto nh = std::make_unique<HNSWIndex>(hnsw->metric, hnsw->m, hnsw->ef, hnsw->max_level_cap, hnsw->dim, hnsw->seed);
        nh->level_draws = hnsw->level_draws; nh->sample_draws = hnsw->sample_draws;   // the RNG stream continues
        PayloadIndex npi; PayloadMap npl;
        std::map<PointId, uint32_t> ordered(hnsw->slot_of.begin(), hnsw->slot_of.end());   // DET-6: ascending id
        for (auto& kv : ordered) {
            PointId id = kv.first;
            if (deleted.count(id)) continue;
            const float* v = hnsw->vec(kv.second);
            auto r = nh->insert(id, v, hnsw->dim);
            if (!r.ok()) return r;
            auto pit = payloads.find(id);
            if (pit != payloads.end()) {
                npi.insert(id, pit->second);
                npl[id] = pit->second;
                auto keys = scalar_keys(pit->second);
                auto e = nh->build_filter_aware_edges(id, v, pit->second, npi, npl, keys);
                if (!e.ok()) return e;
            }
        }
        hnsw = std::move(nh); payload_index = std::move(npi); payloads = std::move(npl);
        deleted.clear();
        return Result<bool>::good(true);
    }
};

}  // namespace ora
