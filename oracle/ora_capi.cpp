// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points over ora_core.hpp, named ora_* with the same
// signatures as the vdb_* entry points of include/vdbmi.h so one ctypes binder serves both.
// Only tests/, the smoke check and the benchmark's cpu_baseline leg may load this library.
#include "ora_core.hpp"

#include <atomic>
#include <thread>

using namespace ora;

namespace {
thread_local std::string g_err;
int fail(const DBError& e) { g_err = e.msg; return (int)e.code; }
int fail_arg(const char* m) { g_err = m; return 101; }
#define ORA_TRY try {
#define ORA_CATCH                                                                        \
    }                                                                                    \
    catch (const PanicError& e) { g_err = e.what(); return 99; }                         \
    catch (const std::exception& e) { g_err = std::string("oracle exception: ") + e.what(); return 99; }

struct scored_point_c { uint64_t id; float raw_score; float sort_key; };
struct query_stats_c { uint32_t rows_scored, expanded, adj_entries, flags; };
struct options_c { uint64_t seed; int32_t device, arith, filter_mode, reserved; };
struct layer_csr_c { size_t n_lists; const uint32_t* owners; const uint64_t* offsets; const uint32_t* neighbors; size_t n_edges; };

int emit(const std::vector<ScoredPoint>& v, scored_point_c* out, size_t* out_n) {
    for (size_t i = 0; i < v.size(); ++i) out[i] = {v[i].id, v[i].raw_score, v[i].sort_key};
    if (out_n) *out_n = v.size();
    return 0;
}
int emit_ids(const std::set<PointId>& s, uint64_t* out, size_t cap, size_t* out_n) {
    size_t k = 0;
    for (PointId id : s) { if (k < cap && out) out[k] = id; ++k; }
    if (out_n) *out_n = k;
    return 0;
}
// Oracle-only knob: number of host threads for the batched calls (1 = the reference's model).
static std::atomic<int> g_threads{1};

template <class F> int run_batch(size_t nq, F&& one) {
    int nt = g_threads.load();
    std::atomic<size_t> next{0}; std::atomic<int> rc{0}; std::string err;
    auto work = [&]() {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nq || rc.load()) return;
            int r;
            try { r = one(i); } catch (const PanicError& e) { g_err = e.what(); r = 99; }
            if (r) { int z = 0; if (rc.compare_exchange_strong(z, r)) err = g_err; }
        }
    };
    if (nt <= 1) work();
    else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(work); for (auto& t : th) t.join(); }
    if (rc.load()) g_err = err;
    return rc.load();
}
}  // namespace

extern "C" {

void ora_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

const char* ora_last_error_message(void) { return g_err.c_str(); }
int ora_device_count(void) { return 0; }
void ora_default_options(options_c* o) { *o = {0, -1, 1, 0, 0}; }

int ora_score(const float* a, size_t na, const float* b, size_t nb, int metric, float* out) {
    if (na != nb) { g_err = "Vectors must be the same length"; return 99; }   // metric.rs:5
    *out = score(a, b, na, (Metric)metric);
    return 0;
}
int ora_score_scan(const float* q, size_t nq, const float* base, size_t n, size_t dim, int metric, int, int, float* out) {
    for (size_t i = 0; i < nq; ++i)
        for (size_t j = 0; j < n; ++j) out[i * n + j] = score(q + i * dim, base + j * dim, dim, (Metric)metric);
    return 0;
}

// ---- values ----
Value* ora_value_int(int64_t v) { return new Value(Value::mk_int(v)); }
Value* ora_value_float(double v) { return new Value(Value::mk_float(v)); }
Value* ora_value_str(const char* s, size_t len) { return new Value(Value::mk_str(std::string(s, len))); }
Value* ora_value_bool(int v) { return new Value(Value::mk_bool(v != 0)); }
Value* ora_value_list_int(const int64_t* v, size_t n) { auto* x = new Value; x->t = Value::ListInt; x->li.assign(v, v + n); return x; }
Value* ora_value_list_float(const double* v, size_t n) { auto* x = new Value; x->t = Value::ListFloat; x->lf.assign(v, v + n); return x; }
Value* ora_value_list_str(const char* const* s, const size_t* lens, size_t n) {
    auto* x = new Value; x->t = Value::ListStr;
    for (size_t i = 0; i < n; ++i) x->ls.emplace_back(s[i], lens[i]);
    return x;
}
Value* ora_value_list_bool(const uint8_t* v, size_t n) { auto* x = new Value; x->t = Value::ListBool; x->lb.assign(v, v + n); return x; }
void ora_value_free(Value* v) { delete v; }
int ora_value_type(const Value* v) { return (int)v->t; }
int ora_value_equals(const Value* a, const Value* b) { return *a == *b; }
int ora_value_compare_scalar(const Value* a, int op, const Value* b, int* is_some, int* out) {
    auto r = compare_scalar(*a, (CmpOp)op, *b);
    *is_some = r.has_value(); *out = r.value_or(false);
    return 0;
}
int ora_value_evaluate_list_query(const Value* self, int kind, int op, size_t n, const Value* v, int* is_some, int* out) {
    ListQuery q{(ListQuery::Kind)kind, v, (CmpOp)op, n};
    if (kind != ListQuery::Length && !v) return fail_arg("value required");
    auto r = evaluate_list_query(*self, q);
    *is_some = r.has_value(); *out = r.value_or(false);
    return 0;
}

// ---- payload ----
Payload* ora_payload_new(void) { return new Payload; }
void ora_payload_free(Payload* p) { delete p; }
int ora_payload_set(Payload* p, const char* key, const Value* v) { p->set(key, *v); return 0; }
const Value* ora_payload_get(const Payload* p, const char* key) { return p->get(key); }
int ora_payload_compare_field(const Payload* p, const char* field, int op, const Value* other, int* out) {
    auto r = p->compare_field(field, (CmpOp)op, *other);
    if (!r.ok()) return fail(r.e);
    *out = *r.v; return 0;
}
int ora_payload_evaluate_list_field(const Payload* p, const char* field, int kind, int op, size_t n, const Value* v, int* out) {
    ListQuery q{(ListQuery::Kind)kind, v, (CmpOp)op, n};
    auto r = p->evaluate_list_field(field, q);
    if (!r.ok()) return fail(r.e);
    *out = *r.v; return 0;
}

// ---- filter ----
Filter* ora_filter_match(const char* key, const Value* v) { auto* f = new Filter; f->kind = Filter::Match; f->key = key; f->value = *v; return f; }
Filter* ora_filter_compare(const char* key, int op, const Value* v) {
    auto* f = new Filter; f->kind = Filter::Compare; f->key = key; f->op = (CmpOp)op; f->value = *v; return f;
}
Filter* ora_filter_and(const Filter* const* c, size_t n) { auto* f = new Filter; f->kind = Filter::And; for (size_t i = 0; i < n; ++i) f->children.push_back(*c[i]); return f; }
Filter* ora_filter_or(const Filter* const* c, size_t n) { auto* f = new Filter; f->kind = Filter::Or; for (size_t i = 0; i < n; ++i) f->children.push_back(*c[i]); return f; }
Filter* ora_filter_not(const Filter* c) { auto* f = new Filter; f->kind = Filter::Not; f->children.push_back(*c); return f; }
void ora_filter_free(Filter* f) { delete f; }
int ora_filter_evaluate(const Filter* f, const Payload* p, int* out) {
    auto r = evaluate_filter(*f, *p);
    if (!r.ok()) return fail(r.e);
    *out = *r.v; return 0;
}

// ---- payload index ----
PayloadIndex* ora_payload_index_new(void) { return new PayloadIndex; }
void ora_payload_index_free(PayloadIndex* x) { delete x; }
int ora_payload_index_insert(PayloadIndex* x, uint64_t id, const Payload* p) { x->insert(id, *p); return 0; }
int ora_payload_index_remove(PayloadIndex* x, uint64_t id, const Payload* p) { x->remove(id, *p); return 0; }
int ora_payload_index_query_exact(const PayloadIndex* x, const char* key, const Value* v, uint64_t* out, size_t cap, size_t* out_n, int* is_some) {
    const std::set<PointId>* s = x->query_exact(key, *v);
    *is_some = s != nullptr; if (out_n) *out_n = 0;
    return s ? emit_ids(*s, out, cap, out_n) : 0;
}
int ora_payload_index_all_for_key(const PayloadIndex* x, const char* key, uint64_t* out, size_t cap, size_t* out_n, int* is_some) {
    auto s = x->all_for_key(key);
    *is_some = s.has_value(); if (out_n) *out_n = 0;
    return s ? emit_ids(*s, out, cap, out_n) : 0;
}

// ---- HNSWIndex ----
int ora_index_new(int metric, size_t m, size_t ef, size_t cap, size_t dim, const options_c* o, HNSWIndex** out) {
    *out = new HNSWIndex((Metric)metric, m, ef, cap, dim, o ? o->seed : 0);
    return 0;
}
void ora_index_free(HNSWIndex* x) { delete x; }
int ora_index_insert(HNSWIndex* x, uint64_t id, const float* v, size_t len) {
    ORA_TRY
    auto r = x->insert(id, v, len);
    return r.ok() ? 0 : fail(r.e);
    ORA_CATCH
}
int ora_index_insert_batch(HNSWIndex* x, const uint64_t* ids, const float* v, size_t n, size_t len) {
    ORA_TRY
    for (size_t i = 0; i < n; ++i) { auto r = x->insert(ids[i], v + i * len, len); if (!r.ok()) return fail(r.e); }
    return 0;
    ORA_CATCH
}
int ora_index_search(const HNSWIndex* x, const float* q, size_t len, size_t top_k, scored_point_c* out, size_t* out_n) {
    ORA_TRY
    auto r = x->search(q, len, top_k);
    if (!r.ok()) return fail(r.e);
    return emit(*r.v, out, out_n);
    ORA_CATCH
}
int ora_index_search_batch(const HNSWIndex* x, const float* q, size_t nq, size_t len, size_t top_k, size_t ef_search,
                           scored_point_c* out, uint32_t* out_n, query_stats_c* stats) {
    return run_batch(nq, [&](size_t i) {
        SearchStats st;
        auto r = x->search(q + i * len, len, top_k, ef_search, &st);
        if (!r.ok()) return fail(r.e);
        size_t n; emit(*r.v, out + i * top_k, &n); out_n[i] = (uint32_t)n;
        if (stats) stats[i] = {(uint32_t)st.rows_scored, (uint32_t)st.expanded, (uint32_t)st.adj_entries, 0};
        return 0;
    });
}
// Brute force over live points, ordered by (sort_key asc, id asc).
int ora_index_exact_knn_batch(const HNSWIndex* x, const float* q, size_t nq, size_t len, size_t top_k, scored_point_c* out, uint32_t* out_n) {
    if (len != x->dim) return fail(err_len_mismatch(x->dim, len));
    return run_batch(nq, [&](size_t i) {
        std::vector<float> qn = x->metric == Cosine ? x->maybe_normalize(q + i * len) : std::vector<float>(q + i * len, q + (i + 1) * len);
        std::vector<ScoredPoint> all;
        for (uint32_t s = 0; s < x->ids.size(); ++s) {
            if (x->deleted[s]) continue;
            float raw = score(qn.data(), x->vec(s), x->dim, x->metric);
            all.push_back({x->ids[s], raw, x->normalize_score(raw)});
        }
        size_t k = std::min(top_k, all.size());
        std::partial_sort(all.begin(), all.begin() + k, all.end(), [](const ScoredPoint& a, const ScoredPoint& b) {
            return a.sort_key < b.sort_key || (a.sort_key == b.sort_key && a.id < b.id);
        });
        all.resize(k);
        size_t n; emit(all, out + i * top_k, &n); out_n[i] = (uint32_t)n;
        return 0;
    });
}
int ora_index_mark_deleted(HNSWIndex* x, uint64_t id) { x->mark_deleted(id); return 0; }
int ora_index_contains(const HNSWIndex* x, uint64_t id) { return x->contains(id); }
size_t ora_index_len(const HNSWIndex* x) { return x->len(); }
int ora_index_metric(const HNSWIndex* x) { return x->metric; }
size_t ora_index_m(const HNSWIndex* x) { return x->m; }
size_t ora_index_ef(const HNSWIndex* x) { return x->ef; }
size_t ora_index_max_level_cap(const HNSWIndex* x) { return x->max_level_cap; }
size_t ora_index_dim(const HNSWIndex* x) { return x->dim; }
size_t ora_index_current_max_level(const HNSWIndex* x) { return x->current_max_level; }
int ora_index_get_entry_point(const HNSWIndex* x, uint64_t* out) { if (!x->entry_point) return 0; *out = *x->entry_point; return 1; }
int ora_index_set_entry_point(HNSWIndex* x, uint64_t id) { x->entry_point = id; return 0; }
int ora_index_set_current_max_level(HNSWIndex* x, size_t l) { x->current_max_level = l; return 0; }
int ora_index_get_vector(const HNSWIndex* x, uint64_t id, float* out, size_t cap) {
    auto it = x->slot_of.find(id);
    if (it == x->slot_of.end() || x->deleted[it->second]) { g_err = "Point with ID " + std::to_string(id) + " not found"; return 1; }
    if (cap < x->dim) return fail_arg("capacity too small");
    std::memcpy(out, x->vec(it->second), x->dim * sizeof(float));
    return 0;
}
int ora_index_layer_neighbors(const HNSWIndex* x, size_t level, uint64_t id, uint64_t* out, size_t cap, size_t* out_n, int* is_some) {
    *is_some = 0; if (out_n) *out_n = 0;
    auto it = x->slot_of.find(id);
    if (it == x->slot_of.end()) return 0;
    const std::vector<uint32_t>* l = x->list(level, it->second);
    if (!l) return 0;
    *is_some = 1;
    for (size_t i = 0; i < l->size(); ++i) if (i < cap && out) out[i] = x->ids[(*l)[i]];
    if (out_n) *out_n = l->size();
    return 0;
}
int ora_index_maybe_normalize(const HNSWIndex* x, const float* v, size_t len, float* out) {
    if (len != x->dim) return fail(err_len_mismatch(x->dim, len));
    auto r = x->maybe_normalize(v); std::memcpy(out, r.data(), len * sizeof(float)); return 0;
}
float ora_index_normalize_score(const HNSWIndex* x, float raw) { return x->normalize_score(raw); }
int ora_index_add_bidirectional_edge(HNSWIndex* x, size_t level, uint64_t a, uint64_t b) {
    auto ia = x->slot_of.find(a), ib = x->slot_of.find(b);
    if (ia == x->slot_of.end() || ib == x->slot_of.end()) return fail_arg("unknown id");
    x->add_bidirectional_edge(level, ia->second, ib->second); return 0;
}
int ora_index_greedy_search_layer_unfiltered(const HNSWIndex* x, const float* q, size_t len, uint64_t entry, size_t level, uint64_t* out_id) {
    if (len != x->dim) { g_err = "Vectors must be the same length"; return 99; }
    auto it = x->slot_of.find(entry);
    if (it == x->slot_of.end()) { g_err = "index out of bounds (unknown entry id)"; return 99; }
    *out_id = x->ids[x->greedy_search_layer_unfiltered(q, it->second, level)];
    return 0;
}
int ora_index_import_graph(HNSWIndex* x, size_t n, const uint64_t* ids, const uint32_t* levels, const float* vectors, const uint8_t* deleted,
                           size_t n_layers, const layer_csr_c* layers, int has_entry, uint64_t entry_id, size_t current_max_level) {
    x->slot_of.clear(); x->ids.assign(ids, ids + n); x->vecs.assign(vectors, vectors + n * x->dim);
    x->levels.assign(n, 0); x->deleted.assign(n, 0); x->n_deleted = 0; x->live.clear(); x->layers.clear();
    for (size_t i = 0; i < n; ++i) {
        x->slot_of[ids[i]] = (uint32_t)i; x->levels[i] = levels ? levels[i] : 0;
        if (deleted && deleted[i]) { x->deleted[i] = 1; x->n_deleted++; } else x->live.insert(ids[i]);
    }
    for (size_t l = 0; l < n_layers; ++l) {
        x->layer_mut(l);
        for (size_t k = 0; k < layers[l].n_lists; ++k) {
            auto& lst = x->list_or_default(l, layers[l].owners[k]);
            lst.assign(layers[l].neighbors + layers[l].offsets[k], layers[l].neighbors + layers[l].offsets[k + 1]);
        }
    }
    if (has_entry) x->entry_point = entry_id; else x->entry_point = std::nullopt;
    x->current_max_level = current_max_level;
    return 0;
}
int ora_index_export_sizes(const HNSWIndex* x, size_t* n, size_t* n_layers, size_t* n_lists, size_t* n_edges, size_t cap_layers) {
    *n = x->ids.size(); *n_layers = x->layers.size();
    for (size_t l = 0; l < x->layers.size() && l < cap_layers; ++l) {
        n_lists[l] = x->layers[l].lists.size(); size_t e = 0;
        for (auto& v : x->layers[l].lists) e += v.size();
        n_edges[l] = e;
    }
    return 0;
}
int ora_index_export_points(const HNSWIndex* x, uint64_t* ids, uint32_t* levels, float* vectors, uint8_t* deleted) {
    size_t n = x->ids.size();
    if (n == 0) return 0;                                  // empty vectors may hand out a null data()
    if (ids) std::memcpy(ids, x->ids.data(), n * 8);
    if (levels) for (size_t i = 0; i < n; ++i) levels[i] = (uint32_t)x->levels[i];
    if (vectors) std::memcpy(vectors, x->vecs.data(), n * x->dim * 4);
    if (deleted) std::memcpy(deleted, x->deleted.data(), n);
    return 0;
}
int ora_index_export_layer(const HNSWIndex* x, size_t level, uint32_t* owners, uint64_t* offsets, uint32_t* neighbors) {
    if (level >= x->layers.size()) return fail_arg("no such layer");
    const auto& L = x->layers[level];
    // lists in ascending owner slot order
    std::vector<uint32_t> order(L.lists.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return L.owner[a] < L.owner[b]; });
    uint64_t off = 0;
    for (size_t k = 0; k < order.size(); ++k) {
        owners[k] = L.owner[order[k]]; offsets[k] = off;
        for (uint32_t s : L.lists[order[k]]) neighbors[off++] = s;
    }
    offsets[order.size()] = off;
    return 0;
}
int ora_index_sync_device(HNSWIndex*) { return 0; }

// ---- Segment ----
int ora_segment_new(HNSWIndex* h, Segment** out) { *out = new Segment(std::unique_ptr<HNSWIndex>(h)); return 0; }
void ora_segment_free(Segment* s) { delete s; }
void ora_segment_set_filter_mode(Segment* s, int mode) { s->filter_mode = (FilterMode)mode; }
int ora_segment_insert(Segment* s, const float* v, size_t len, const Payload* p, uint64_t* out_id) {
    ORA_TRY
    auto r = s->insert(v, len, p);
    if (!r.ok()) return fail(r.e);
    if (out_id) *out_id = *r.v;
    return 0;
    ORA_CATCH
}
int ora_segment_insert_batch(Segment* s, const float* v, size_t n, size_t len, const Payload* const* payloads, uint64_t* out_ids) {
    ORA_TRY
    for (size_t i = 0; i < n; ++i) {
        auto r = s->insert(v + i * len, len, payloads ? payloads[i] : nullptr);
        if (!r.ok()) return fail(r.e);
        if (out_ids) out_ids[i] = *r.v;
    }
    return 0;
    ORA_CATCH
}
int ora_segment_get_vector(const Segment* s, uint64_t id, float* out, size_t cap) {
    const float* v = s->get_vector(id);
    if (!v) { g_err = "Point with ID " + std::to_string(id) + " not found"; return 1; }
    if (cap < s->hnsw->dim) return fail_arg("capacity too small");
    std::memcpy(out, v, s->hnsw->dim * 4); return 0;
}
int ora_segment_delete(Segment* s, uint64_t id) {
    ORA_TRY
    auto r = s->del(id); return r.ok() ? 0 : fail(r.e);
    ORA_CATCH
}
int ora_segment_search(const Segment* s, const float* q, size_t len, size_t top_k, scored_point_c* out, size_t* out_n) {
    ORA_TRY
    auto r = s->search(q, len, top_k); if (!r.ok()) return fail(r.e); return emit(*r.v, out, out_n);
    ORA_CATCH
}
int ora_segment_search_with_filter(const Segment* s, const float* q, size_t len, size_t top_k, const Filter* f, scored_point_c* out, size_t* out_n) {
    ORA_TRY
    auto r = s->search_with_filter(q, len, top_k, f); if (!r.ok()) return fail(r.e); return emit(*r.v, out, out_n);
    ORA_CATCH
}
int ora_segment_search_unfiltered(const Segment* s, const float* q, size_t len, size_t top_k, scored_point_c* out, size_t* out_n) {
    ORA_TRY
    auto r = s->search_unfiltered(q, len, top_k); if (!r.ok()) return fail(r.e); return emit(*r.v, out, out_n);
    ORA_CATCH
}
int ora_segment_post_filter(const Segment* s, const float* q, size_t len, size_t top_k, const Filter* f, scored_point_c* out, size_t* out_n) {
    ORA_TRY
    auto r = s->post_filter(q, len, top_k, f); if (!r.ok()) return fail(r.e); return emit(*r.v, out, out_n);
    ORA_CATCH
}
int ora_segment_search_batch(const Segment* s, const float* q, size_t nq, size_t len, size_t top_k, size_t ef_search, const Filter* f,
                             scored_point_c* out, uint32_t* out_n) {
    return run_batch(nq, [&](size_t i) {
        auto r = f ? s->search_with_filter(q + i * len, len, top_k, f, ef_search) : s->search(q + i * len, len, top_k, ef_search);
        if (!r.ok()) return fail(r.e);
        size_t n; emit(*r.v, out + i * top_k, &n); out_n[i] = (uint32_t)n;
        return 0;
    });
}
const Payload* ora_segment_get_payload(const Segment* s, uint64_t id) { auto it = s->payloads.find(id); return it == s->payloads.end() ? nullptr : &it->second; }
int ora_segment_is_deleted(const Segment* s, uint64_t id) { return s->deleted.count(id) != 0; }
int ora_segment_purge(Segment* s) {
    ORA_TRY
    auto r = s->purge(); return r.ok() ? 0 : fail(r.e);
    ORA_CATCH
}
const HNSWIndex* ora_segment_hnsw(const Segment* s) { return s->hnsw.get(); }
HNSWIndex* ora_segment_hnsw_mut(Segment* s) { return s->hnsw.get(); }
const PayloadIndex* ora_segment_payload_index(const Segment* s) { return &s->payload_index; }
size_t ora_segment_payload_count(const Segment* s) { return s->payloads.size(); }
int ora_segment_in_place_filtered_search(const Segment* s, const float* q, size_t len, size_t top_k, const Filter* f, scored_point_c* out, size_t* out_n) {
    ORA_TRY
    auto r = in_place_filtered_search(*s->hnsw, q, len, top_k, s->payloads, s->payload_index, f, s->filter_mode);
    if (!r.ok()) return fail(r.e);
    return emit(*r.v, out, out_n);
    ORA_CATCH
}
int ora_segment_compile_filter(const Segment* s, const Filter* f, size_t* n_pass, size_t* n_err) {
    size_t np = 0, ne = 0;
    for (uint32_t i = 0; i < s->hnsw->ids.size(); ++i) {
        if (s->hnsw->deleted[i]) continue;
        auto it = s->payloads.find(s->hnsw->ids[i]);
        if (it == s->payloads.end()) continue;
        auto r = evaluate_filter(*f, it->second);
        if (!r.ok()) ++ne; else if (*r.v) ++np;
    }
    *n_pass = np; *n_err = ne; return 0;
}
// k-way merge of per-shard lists by (sort_key asc, id asc), host buffers.
int ora_topk_merge(const scored_point_c* lists, const uint32_t* counts, size_t n_shards, size_t nq, size_t k, scored_point_c* out, uint32_t* out_n) {
    for (size_t q = 0; q < nq; ++q) {
        std::vector<scored_point_c> all;
        for (size_t s = 0; s < n_shards; ++s)
            for (uint32_t i = 0; i < counts[s * nq + q]; ++i) all.push_back(lists[(s * nq + q) * k + i]);
        std::stable_sort(all.begin(), all.end(), [](const scored_point_c& a, const scored_point_c& b) {
            return a.sort_key < b.sort_key || (a.sort_key == b.sort_key && a.id < b.id);
        });
        size_t n = std::min(k, all.size());
        for (size_t i = 0; i < n; ++i) out[q * k + i] = all[i];
        out_n[q] = (uint32_t)n;
    }
    return 0;
}

}  // extern "C"
