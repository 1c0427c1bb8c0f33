// extern "C" surface of libvdbmi (declarations and reference citations: include/vdbmi.h).
#include <cstdio>
#include <cstring>

#include "index.hpp"

using namespace vdb;

namespace {
template <class F> vdb_status guard(F&& f) {
    try { f(); return VDB_OK; }
    catch (const Failure& e) { set_last_error(e.e.msg); return e.e.code; }
    catch (const std::bad_alloc&) { set_last_error("out of host memory"); return VDB_ERR_DEVICE; }
    catch (const std::exception& e) { set_last_error(std::string("internal error: ") + e.what()); return VDB_ERR_PANIC; }
}
void need(const void* p, const char* what) { if (!p) throw Failure(err_arg(std::string(what) + " is NULL")); }
void emit(const std::vector<vdb_scored_point>& v, vdb_scored_point* out, size_t* out_n) {
    if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(vdb_scored_point));
    if (out_n) *out_n = v.size();
}
vdb_status emit_ids(const std::set<uint64_t>* s, uint64_t* out, size_t cap, size_t* out_n, int* is_some) {
    if (is_some) *is_some = s != nullptr;
    size_t k = 0;
    if (s) for (uint64_t id : *s) { if (out && k < cap) out[k] = id; ++k; }
    if (out_n) *out_n = k;
    return VDB_OK;
}
vdb_status tri_out(const Tri& t, int* out) {
    if (t.s == Tri::Err) { Error e = err_invalid_payload(t.msg); set_last_error(e.msg); return e.code; }
    *out = t.s == Tri::True; return VDB_OK;
}
vdb_value* mk(ValueData d) { auto* v = new vdb_value; v->d = std::move(d); return v; }
template <class T> T* null_arg(const char* what) { set_last_error(std::string(what) + " is NULL"); return nullptr; }
}  // namespace

extern "C" {

int vdb_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
void vdb_default_options(vdb_options* o) { if (o) *o = vdb_options{0, 0, VDB_ARITH_FAST, VDB_FILTER_LITERAL, 0}; }

// ---- metric.rs ----
vdb_status vdb_score(const float* a, size_t na, const float* b, size_t nb, vdb_metric metric, float* out) {
    return guard([&] {
        need(a, "a"); need(b, "b"); need(out, "out");
        if (na != nb) throw Failure(err_panic("Vectors must be the same length"));       // metric.rs:5
        if (na == 0) {   // empty sums fold from -0.0 (metric.rs:16-33)
            *out = metric == VDB_COSINE ? 1.0f - (-0.0f / (0.0f * 0.0f + 1e-10f)) : -0.0f;
            if (metric == VDB_EUCLIDEAN) *out = -0.0f;
            return;
        }
        float r = 0;
        vdb_status st = vdb_score_scan(a, 1, b, 1, na, metric, VDB_ARITH_STRICT, 0, &r);
        if (st != VDB_OK) throw Failure(Error{st, vdb_last_error_message()});
        *out = r;
    });
}
vdb_status vdb_score_scan(const float* queries, size_t nq, const float* base, size_t n, size_t dim, vdb_metric metric, vdb_arith arith,
                          int device, float* out) {
    return guard([&] {
        need(queries, "queries"); need(base, "base"); need(out, "out");
        if (dim == 0) throw Failure(err_arg("dim must be positive"));
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw Failure(err_device("no HIP device available; libvdbmi has no CPU path"));
        VDB_HIP(hipSetDevice(device));
        const uint32_t stride = (uint32_t)((dim + 3) & ~size_t(3));
        DevBuf<float> dq, db, dout;
        hipStream_t s = nullptr;
        dq.reserve(nq * dim, s, false); db.reserve(n * stride, s, false); dout.reserve(nq * n, s, false);
        VDB_HIP(hipMemcpy(dq.p, queries, nq * dim * 4, hipMemcpyHostToDevice));
        if (stride == dim) VDB_HIP(hipMemcpy(db.p, base, n * dim * 4, hipMemcpyHostToDevice));
        else {
            std::vector<float> padded(n * stride, 0.f);
            for (size_t i = 0; i < n; ++i) std::memcpy(&padded[i * stride], base + i * dim, dim * 4);
            VDB_HIP(hipMemcpy(db.p, padded.data(), padded.size() * 4, hipMemcpyHostToDevice));
        }
        launch_score_scan(dq.p, (uint32_t)nq, db.p, (uint32_t)n, (uint32_t)dim, stride, (int)metric, (int)arith, dout.p, s);
        VDB_HIP(hipMemcpy(out, dout.p, nq * n * 4, hipMemcpyDeviceToHost));
    });
}
vdb_status vdb_score_scan_dev(const float* d_queries, size_t nq, const float* d_base, size_t n, size_t dim, size_t stride, vdb_metric metric,
                              vdb_arith arith, float* d_out, void* stream) {
    return guard([&] {
        need(d_queries, "d_queries"); need(d_base, "d_base"); need(d_out, "d_out");
        if (dim == 0) throw Failure(err_arg("dim must be positive"));
        if (stride < dim) throw Failure(err_arg("the row stride is shorter than the rows"));
        if (n > 0xFFFFFFFFull || nq > 0xFFFFFFFFull || stride > 0xFFFFFFFFull) throw Failure(err_arg("more than 2^32 - 1 rows, queries or floats per row in one scan"));
        launch_score_scan(d_queries, (uint32_t)nq, d_base, (uint32_t)n, (uint32_t)dim, (uint32_t)stride, (int)metric, (int)arith, d_out, (hipStream_t)stream);
    });
}

// ---- payload.rs ----
vdb_value* vdb_value_int(int64_t v) { return mk(v); }
vdb_value* vdb_value_float(double v) { return mk(OF64{v}); }
vdb_value* vdb_value_str(const char* s, size_t len) { return mk(std::string(s ? s : "", s ? len : 0)); }
vdb_value* vdb_value_bool(int v) { return mk(Bool8{(uint8_t)(v != 0)}); }
// constructors return NULL (with the reason in vdb_last_error_message) when handed a NULL array of n > 0 elements
vdb_value* vdb_value_list_int(const int64_t* v, size_t n) { if (n && !v) return null_arg<vdb_value>("v"); return mk(std::vector<int64_t>(v, v + n)); }
vdb_value* vdb_value_list_float(const double* v, size_t n) {
    if (n && !v) return null_arg<vdb_value>("v");
    std::vector<OF64> l(n); for (size_t i = 0; i < n; ++i) l[i].v = v[i]; return mk(std::move(l));
}
vdb_value* vdb_value_list_str(const char* const* s, const size_t* lens, size_t n) {
    if (n && (!s || !lens)) return null_arg<vdb_value>("s / lens");
    for (size_t i = 0; i < n; ++i) if (!s[i] && lens[i]) return null_arg<vdb_value>("s[i]");
    std::vector<std::string> l; for (size_t i = 0; i < n; ++i) l.emplace_back(s[i] ? s[i] : "", lens[i]); return mk(std::move(l));
}
vdb_value* vdb_value_list_bool(const uint8_t* v, size_t n) {
    if (n && !v) return null_arg<vdb_value>("v");
    std::vector<Bool8> l(n); for (size_t i = 0; i < n; ++i) l[i].v = v[i] != 0; return mk(std::move(l));
}
void vdb_value_free(vdb_value* v) { delete v; }
int vdb_value_type(const vdb_value* v) { if (!v) { set_last_error("value is NULL"); return -1; } return v->kind(); }
int vdb_value_equals(const vdb_value* a, const vdb_value* b) { if (!a || !b) { set_last_error("value is NULL"); return 0; } return *a == *b; }
vdb_status vdb_value_compare_scalar(const vdb_value* a, vdb_cmp_op op, const vdb_value* b, int* is_some, int* out) {
    return guard([&] { need(a, "a"); need(b, "b"); need(is_some, "is_some"); need(out, "out"); auto r = compare_scalar(*a, op, *b); *is_some = r.has_value(); *out = r.value_or(false); });
}
vdb_status vdb_value_evaluate_list_query(const vdb_value* self, vdb_list_query kind, vdb_cmp_op op, size_t n, const vdb_value* v, int* is_some, int* out) {
    return guard([&] { need(self, "self"); need(is_some, "is_some"); need(out, "out"); auto r = evaluate_list_query(*self, kind, op, n, v); *is_some = r.has_value(); *out = r.value_or(false); });
}
vdb_payload* vdb_payload_new(void) { return new vdb_payload; }
void vdb_payload_free(vdb_payload* p) { delete p; }
vdb_status vdb_payload_set(vdb_payload* p, const char* key, const vdb_value* v) { return guard([&] { need(p, "payload"); need(key, "key"); need(v, "value"); p->set(key, *v); }); }
const vdb_value* vdb_payload_get(const vdb_payload* p, const char* key) { return p && key ? p->get(key) : nullptr; }
vdb_status vdb_payload_compare_field(const vdb_payload* p, const char* field, vdb_cmp_op op, const vdb_value* other, int* out) {
    if (!p || !field || !other || !out) { set_last_error("NULL argument"); return VDB_ERR_ARGUMENT; }
    return tri_out(p->compare_field(field, op, *other), out);
}
vdb_status vdb_payload_evaluate_list_field(const vdb_payload* p, const char* field, vdb_list_query kind, vdb_cmp_op op, size_t n, const vdb_value* v, int* out) {
    if (!p || !field || !out) { set_last_error("NULL argument"); return VDB_ERR_ARGUMENT; }
    return tri_out(p->evaluate_list_field(field, kind, op, n, v), out);
}

// ---- filters.rs ----
vdb_filter* vdb_filter_match(const char* key, const vdb_value* v) {
    if (!key || !v) return null_arg<vdb_filter>("key / value");
    auto* f = new vdb_filter; f->kind = vdb_filter::Match; f->key = key; f->value = *v; return f;
}
vdb_filter* vdb_filter_compare(const char* key, vdb_cmp_op op, const vdb_value* v) {
    if (!key || !v) return null_arg<vdb_filter>("key / value");
    auto* f = new vdb_filter; f->kind = vdb_filter::Compare; f->key = key; f->op = op; f->value = *v; return f;
}
static vdb_filter* nary(vdb_filter::Kind kind, const vdb_filter* const* c, size_t n) {
    if (n && !c) return null_arg<vdb_filter>("children");
    for (size_t i = 0; i < n; ++i) if (!c[i]) return null_arg<vdb_filter>("children[i]");
    auto* f = new vdb_filter; f->kind = kind; for (size_t i = 0; i < n; ++i) f->kids.push_back(*c[i]); return f;
}
vdb_filter* vdb_filter_and(const vdb_filter* const* c, size_t n) { return nary(vdb_filter::And, c, n); }
vdb_filter* vdb_filter_or(const vdb_filter* const* c, size_t n) { return nary(vdb_filter::Or, c, n); }
vdb_filter* vdb_filter_not(const vdb_filter* c) { if (!c) return null_arg<vdb_filter>("child"); auto* f = new vdb_filter; f->kind = vdb_filter::Not; f->kids.push_back(*c); return f; }
void vdb_filter_free(vdb_filter* f) { delete f; }
vdb_status vdb_filter_evaluate(const vdb_filter* f, const vdb_payload* p, int* out) {
    if (!f || !p || !out) { set_last_error("NULL argument"); return VDB_ERR_ARGUMENT; }
    return tri_out(f->eval(*p), out);
}

// ---- stores.rs ----
vdb_payload_index* vdb_payload_index_new(void) { return new vdb_payload_index; }
void vdb_payload_index_free(vdb_payload_index* x) { delete x; }
vdb_status vdb_payload_index_insert(vdb_payload_index* x, uint64_t id, const vdb_payload* p) { return guard([&] { need(x, "index"); need(p, "payload"); x->insert(id, *p); }); }
vdb_status vdb_payload_index_remove(vdb_payload_index* x, uint64_t id, const vdb_payload* p) { return guard([&] { need(x, "index"); need(p, "payload"); x->remove(id, *p); }); }
vdb_status vdb_payload_index_query_exact(const vdb_payload_index* x, const char* key, const vdb_value* v, uint64_t* out, size_t cap, size_t* out_n, int* is_some) {
    if (!x || !key || !v) { set_last_error("NULL argument"); return VDB_ERR_ARGUMENT; }
    return emit_ids(x->query_exact(key, *v), out, cap, out_n, is_some);
}
vdb_status vdb_payload_index_all_for_key(const vdb_payload_index* x, const char* key, uint64_t* out, size_t cap, size_t* out_n, int* is_some) {
    if (!x || !key) { set_last_error("NULL argument"); return VDB_ERR_ARGUMENT; }
    auto s = x->all_for_key(key);
    return emit_ids(s ? &*s : nullptr, out, cap, out_n, is_some);
}

// ---- hnsw.rs ----
vdb_status vdb_index_new(vdb_metric metric, size_t m, size_t ef, size_t cap, size_t dim, const vdb_options* opts, vdb_index** out) {
    return guard([&] {
        need(out, "out");
        vdb_options o; vdb_default_options(&o);
        if (opts) o = *opts;
        *out = new vdb_index(metric, m, ef, cap, dim, o);
    });
}
void vdb_index_free(vdb_index* x) { delete x; }
vdb_status vdb_index_insert(vdb_index* x, uint64_t id, const float* v, size_t len) { return guard([&] { need(x, "index"); need(v, "vector"); x->insert(id, v, len); }); }
vdb_status vdb_index_insert_batch(vdb_index* x, const uint64_t* ids, const float* v, size_t n, size_t len) {
    return guard([&] { need(x, "index"); if (n) { need(ids, "ids"); need(v, "vectors"); } x->insert_batch(ids, v, n, len); });
}
vdb_status vdb_index_search(const vdb_index* x, const float* q, size_t len, size_t top_k, vdb_scored_point* out, size_t* out_n) {
    return guard([&] {
        need(x, "index"); need(q, "query");
        auto* h = const_cast<vdb_index*>(x);
        if (out_n) *out_n = 0;
        if (!h->entry_point) return;                                                    // hnsw.rs:515-518
        if (len != h->dim) throw Failure(err_length_mismatch(h->dim, len));            // hnsw.rs:519-524
        if (top_k == 0) return;
        need(out, "out");
        std::vector<vdb_scored_point> buf(top_k); uint32_t n = 0;
        h->search_batch(q, 1, top_k, 0, {}, buf.data(), &n, nullptr);
        buf.resize(n); emit(buf, out, out_n);
    });
}
vdb_status vdb_index_search_batch(const vdb_index* x, const float* q, size_t nq, size_t len, size_t top_k, size_t ef_search,
                                  vdb_scored_point* out, uint32_t* out_n, vdb_query_stats* stats) {
    return guard([&] {
        need(x, "index"); auto* h = const_cast<vdb_index*>(x);
        if (nq == 0) return;
        need(q, "queries"); need(out_n, "out_n");
        if (h->entry_point && len != h->dim) throw Failure(err_length_mismatch(h->dim, len));
        if (top_k) need(out, "out");
        h->search_batch(q, nq, top_k, ef_search, {}, out, out_n, stats);
    });
}
vdb_status vdb_index_search_batch_dev(const vdb_index* x, const float* d_q, size_t nq, size_t top_k, size_t ef_search, vdb_scored_point* d_out,
                                      uint32_t* d_out_n, vdb_query_stats* d_stats, void* stream) {
    return guard([&] {
        need(x, "index"); need(d_q, "d_queries"); need(d_out, "d_out"); need(d_out_n, "d_out_n");
        const_cast<vdb_index*>(x)->search_batch_dev(d_q, nq, top_k, ef_search, d_out, d_out_n, d_stats, (hipStream_t)stream);
    });
}
vdb_status vdb_index_exact_knn_batch(const vdb_index* x, const float* q, size_t nq, size_t len, size_t top_k, vdb_scored_point* out, uint32_t* out_n) {
    return guard([&] {
        need(x, "index"); need(q, "queries"); need(out, "out"); need(out_n, "out_n");
        auto* h = const_cast<vdb_index*>(x);
        if (len != h->dim) throw Failure(err_length_mismatch(h->dim, len));
        if (h->len() == 0) { std::memset(out_n, 0, nq * 4); return; }
        h->exact_knn(q, nq, top_k, false, out, out_n, nullptr);
    });
}
vdb_status vdb_index_exact_knn_batch_dev(const vdb_index* x, const float* d_q, size_t nq, size_t top_k, vdb_scored_point* d_out, uint32_t* d_out_n, void* stream) {
    return guard([&] {
        need(x, "index"); need(d_q, "d_queries"); need(d_out, "d_out"); need(d_out_n, "d_out_n");
        const_cast<vdb_index*>(x)->exact_knn(d_q, nq, top_k, true, d_out, d_out_n, (hipStream_t)stream);
    });
}
vdb_status vdb_index_exact_knn_batch_dev_noalloc(const vdb_index* x, const float* d_q, size_t nq, size_t top_k, vdb_scored_point* d_out, uint32_t* d_out_n, void* stream) {
    return guard([&] {
        need(x, "index"); need(d_q, "d_queries"); need(d_out, "d_out"); need(d_out_n, "d_out_n");
        const_cast<vdb_index*>(x)->exact_knn_static(d_q, nq, top_k, d_out, d_out_n, (hipStream_t)stream);
    });
}
vdb_status vdb_index_mark_deleted(vdb_index* x, uint64_t id) { return guard([&] { need(x, "index"); x->mark_deleted(id); }); }
int vdb_index_contains(const vdb_index* x, uint64_t id) { return x->slot_of.count(id) != 0; }
size_t vdb_index_len(const vdb_index* x) { return x->len(); }
int vdb_index_metric(const vdb_index* x) { return (int)x->metric; }
size_t vdb_index_m(const vdb_index* x) { return x->m; }
size_t vdb_index_ef(const vdb_index* x) { return x->ef; }
size_t vdb_index_max_level_cap(const vdb_index* x) { return x->max_level_cap; }
size_t vdb_index_dim(const vdb_index* x) { return x->dim; }
size_t vdb_index_current_max_level(const vdb_index* x) { return x->current_max_level; }
int vdb_index_get_entry_point(const vdb_index* x, uint64_t* out) { if (!x->entry_point) return 0; if (out) *out = *x->entry_point; return 1; }
vdb_status vdb_index_set_entry_point(vdb_index* x, uint64_t id) { return guard([&] { need(x, "index"); x->entry_point = id; }); }
vdb_status vdb_index_set_current_max_level(vdb_index* x, size_t level) { return guard([&] { need(x, "index"); x->current_max_level = level; }); }
vdb_status vdb_index_get_vector(const vdb_index* x, uint64_t id, float* out, size_t cap) {
    return guard([&] {
        need(x, "index"); need(out, "out");
        auto it = x->slot_of.find(id);
        if (it == x->slot_of.end() || x->deleted[it->second]) throw Failure(err_not_found(id));
        if (cap < x->dim) throw Failure(err_arg("capacity too small"));
        std::memcpy(out, &x->hvecs[(size_t)it->second * x->dim], x->dim * 4);
    });
}
vdb_status vdb_index_layer_neighbors(const vdb_index* x, size_t level, uint64_t id, uint64_t* out, size_t cap, size_t* out_n, int* is_some) {
    return guard([&] {
        need(x, "index");
        if (is_some) *is_some = 0;
        if (out_n) *out_n = 0;
        auto it = x->slot_of.find(id);
        if (it == x->slot_of.end()) return;
        const std::vector<uint32_t>* l = x->list(level, it->second);
        if (!l) return;
        if (is_some) *is_some = 1;
        for (size_t i = 0; i < l->size(); ++i) if (out && i < cap) out[i] = x->ids[(*l)[i]];
        if (out_n) *out_n = l->size();
    });
}
vdb_status vdb_index_maybe_normalize(const vdb_index* x, const float* v, size_t len, float* out) {
    return guard([&] {
        need(x, "index"); need(v, "v"); need(out, "out");
        if (len != x->dim) throw Failure(err_length_mismatch(x->dim, len));
        auto r = const_cast<vdb_index*>(x)->maybe_normalize(v);
        std::memcpy(out, r.data(), len * 4);
    });
}
float vdb_index_normalize_score(const vdb_index* x, float raw) { return x->normalize_score(raw); }
vdb_status vdb_index_add_bidirectional_edge(vdb_index* x, size_t level, uint64_t a, uint64_t b) {
    return guard([&] {
        need(x, "index");
        auto ia = x->slot_of.find(a), ib = x->slot_of.find(b);
        if (ia == x->slot_of.end() || ib == x->slot_of.end()) throw Failure(err_arg("unknown id"));
        x->add_bidirectional_edge(level, ia->second, ib->second);
    });
}
vdb_status vdb_index_greedy_search_layer_unfiltered(const vdb_index* x, const float* q, size_t len, uint64_t entry, size_t level, uint64_t* out_id) {
    return guard([&] {
        need(x, "index"); need(q, "query"); need(out_id, "out_id");
        if (len != x->dim) throw Failure(err_panic("Vectors must be the same length"));
        auto it = x->slot_of.find(entry);
        if (it == x->slot_of.end()) throw Failure(err_panic("unknown entry id"));
        *out_id = x->ids[const_cast<vdb_index*>(x)->greedy_unfiltered(q, it->second, level)];
    });
}
vdb_status vdb_index_import_graph(vdb_index* x, size_t n, const uint64_t* ids, const uint32_t* levels, const float* vectors, const uint8_t* deleted,
                                  size_t n_layers, const vdb_layer_csr* layers, int has_entry, uint64_t entry_id, size_t current_max_level) {
    return guard([&] {
        need(x, "index");
        if (n) { need(ids, "ids"); need(vectors, "vectors"); }
        if (n_layers) need(layers, "layers");
        x->import_graph(n, ids, levels, vectors, deleted, n_layers, layers, has_entry != 0, entry_id, current_max_level);
    });
}
vdb_status vdb_index_export_sizes(const vdb_index* x, size_t* n, size_t* n_layers, size_t* n_lists, size_t* n_edges, size_t cap_layers) {
    return guard([&] {
        need(x, "index");
        if (n) *n = x->len();
        if (n_layers) *n_layers = x->layers.size();
        for (size_t l = 0; l < x->layers.size() && l < cap_layers; ++l) {
            size_t e = 0; for (auto& v : x->layers[l].lists) e += v.size();
            if (n_lists) n_lists[l] = x->layers[l].lists.size();
            if (n_edges) n_edges[l] = e;
        }
    });
}
vdb_status vdb_index_export_points(const vdb_index* x, uint64_t* ids, uint32_t* levels, float* vectors, uint8_t* deleted) {
    return guard([&] {
        need(x, "index");
        const size_t n = x->len();
        if (ids && n) std::memcpy(ids, x->ids.data(), n * 8);
        if (levels && n) std::memcpy(levels, x->levels.data(), n * 4);
        if (vectors && n) std::memcpy(vectors, x->hvecs.data(), n * x->dim * 4);
        if (deleted && n) std::memcpy(deleted, x->deleted.data(), n);
    });
}
vdb_status vdb_index_export_layer(const vdb_index* x, size_t level, uint32_t* owners, uint64_t* offsets, uint32_t* neighbors) {
    return guard([&] {
        need(x, "index");
        if (level >= x->layers.size()) throw Failure(err_arg("no such layer"));
        const HostLayer& L = x->layers[level];
        std::vector<uint32_t> order(L.lists.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return L.owner[a] < L.owner[b]; });
        uint64_t off = 0;
        for (size_t k = 0; k < order.size(); ++k) {
            owners[k] = L.owner[order[k]]; offsets[k] = off;
            for (uint32_t s : L.lists[order[k]]) neighbors[off++] = s;
        }
        offsets[order.size()] = off;
    });
}
// ---- graph file (layout in vdbmi.h) ----
namespace {
constexpr char GRAPH_MAGIC[8] = {'V', 'D', 'B', 'G', 'R', 'A', 'F', '1'};
struct GraphFileHeader {
    char magic[8]; uint32_t version, metric; uint64_t m, ef, max_level_cap, dim, n, n_layers;
    uint64_t has_entry, entry_id, current_max_level, level_draws, sample_draws, seed;
};
static_assert(sizeof(GraphFileHeader) == 112, "graph file header is 112 bytes");
struct File {
    FILE* f; std::string path;
    File(const char* p, const char* mode) : f(std::fopen(p, mode)), path(p) { if (!f) throw Failure(err_io("cannot open " + path)); }
    ~File() { if (f) std::fclose(f); }
    void put(const void* d, size_t bytes) {
        if (bytes && std::fwrite(d, 1, bytes, f) != bytes) throw Failure(err_io("short write to " + path));
        static const char zero[8] = {0};
        if (bytes % 8 && std::fwrite(zero, 1, 8 - bytes % 8, f) != 8 - bytes % 8) throw Failure(err_io("short write to " + path));
    }
    void get(void* d, size_t bytes) {
        if (bytes && std::fread(d, 1, bytes, f) != bytes) throw Failure(err_io(path + " is truncated"));
        char pad[8];
        if (bytes % 8 && std::fread(pad, 1, 8 - bytes % 8, f) != 8 - bytes % 8) throw Failure(err_io(path + " is truncated"));
    }
    void close() {                                  // buffered data reaches the file (or the failure is reported) here
        FILE* g = f; f = nullptr;
        if (g && std::fclose(g) != 0) throw Failure(err_io("cannot finish writing " + path));
    }
};
}  // namespace
extern "C++" {
namespace {
void write_index(File& out, const vdb_index* x) {
    GraphFileHeader h{};
    std::memcpy(h.magic, GRAPH_MAGIC, 8); h.version = 1; h.metric = (uint32_t)x->metric;
    h.m = x->m; h.ef = x->ef; h.max_level_cap = x->max_level_cap; h.dim = x->dim; h.n = x->len(); h.n_layers = x->layers.size();
    h.has_entry = x->entry_point ? 1 : 0; h.entry_id = x->entry_point ? *x->entry_point : 0; h.current_max_level = x->current_max_level;
    h.level_draws = x->level_draws; h.sample_draws = x->sample_draws; h.seed = x->opts.seed;
    out.put(&h, sizeof h);
    out.put(x->ids.data(), h.n * 8); out.put(x->levels.data(), h.n * 4); out.put(x->deleted.data(), h.n);
    out.put(x->hvecs.data(), h.n * x->dim * 4);
    for (size_t l = 0; l < x->layers.size(); ++l) {
        size_t n_lists = 0, n_edges = 0;
        n_lists = x->layers[l].lists.size(); for (auto& v : x->layers[l].lists) n_edges += v.size();
        std::vector<uint32_t> owners(n_lists), nb(std::max<size_t>(n_edges, 1)); std::vector<uint64_t> off(n_lists + 1);
        if (vdb_index_export_layer(x, l, owners.data(), off.data(), nb.data()) != VDB_OK) throw Failure(err_arg(vdb_last_error_message()));
        const uint64_t cnt[2] = {n_lists, n_edges};
        out.put(cnt, 16); out.put(owners.data(), n_lists * 4); out.put(off.data(), (n_lists + 1) * 8); out.put(nb.data(), n_edges * 4);
    }
}
// The index part of a file, read and checked against the handle it is meant for; nothing is applied yet.
struct IndexImage {
    GraphFileHeader h{};
    std::vector<uint64_t> ids; std::vector<uint32_t> levels; std::vector<uint8_t> del; std::vector<float> vecs;
    std::vector<std::vector<uint32_t>> owners, nbs; std::vector<std::vector<uint64_t>> offs; std::vector<vdb_layer_csr> csr;
    uint64_t file_bytes = 0;
    void read(File& in, const vdb_index* x) {
        in.get(&h, sizeof h);
        if (std::memcmp(h.magic, GRAPH_MAGIC, 8) != 0 || h.version != 1) throw Failure(err_arg(in.path + " is not a version-1 graph file"));
        if (h.metric != (uint32_t)x->metric || h.dim != x->dim)
            throw Failure(err_arg(in.path + " holds metric " + std::to_string(h.metric) + ", dimension " + std::to_string(h.dim) + "; the index was created with metric " +
                                  std::to_string((int)x->metric) + ", dimension " + std::to_string(x->dim)));
        // nothing is allocated from a count the file cannot back with bytes
        std::fseek(in.f, 0, SEEK_END); file_bytes = (uint64_t)std::ftell(in.f); std::fseek(in.f, (long)sizeof h, SEEK_SET);
        if (h.n > 0x7FFFFFF0ull || h.n_layers > 64 || h.n * (13 + 4 * h.dim) > file_bytes) throw Failure(err_arg(in.path + " has an implausible header"));
        ids.resize(h.n); levels.resize(h.n); del.resize(h.n); vecs.resize(h.n * h.dim);
        in.get(ids.data(), h.n * 8); in.get(levels.data(), h.n * 4); in.get(del.data(), h.n); in.get(vecs.data(), h.n * h.dim * 4);
        owners.resize(h.n_layers); nbs.resize(h.n_layers); offs.resize(h.n_layers); csr.resize(h.n_layers);
        for (size_t l = 0; l < h.n_layers; ++l) {
            uint64_t cnt[2]; in.get(cnt, 16);
            if (cnt[0] > h.n || cnt[1] > file_bytes / 4) throw Failure(err_arg(in.path + ": layer " + std::to_string(l) + " has implausible sizes"));
            owners[l].resize(cnt[0]); offs[l].resize(cnt[0] + 1); nbs[l].resize(cnt[1]);
            in.get(owners[l].data(), cnt[0] * 4); in.get(offs[l].data(), (cnt[0] + 1) * 8); in.get(nbs[l].data(), cnt[1] * 4);
            if (offs[l].back() != cnt[1]) throw Failure(err_arg(in.path + ": layer " + std::to_string(l) + " offsets do not cover its neighbours"));
            csr[l] = {(size_t)cnt[0], owners[l].data(), offs[l].data(), nbs[l].data(), (size_t)cnt[1]};
        }
    }
    void apply(vdb_index* x) {            // import_graph validates the lists before it touches the index
        x->import_graph(h.n, ids.data(), levels.data(), vecs.data(), del.data(), h.n_layers, csr.data(), h.has_entry != 0, h.entry_id, h.current_max_level);
        x->level_draws = h.level_draws; x->sample_draws = h.sample_draws; x->opts.seed = h.seed;     // the saved build's random streams continue
    }
};
constexpr char SEGMENT_MAGIC[8] = {'V', 'D', 'B', 'S', 'E', 'G', 'M', '1'};
void put_u64(File& f, uint64_t v) { f.put(&v, 8); }
uint64_t get_u64(File& f) { uint64_t v; f.get(&v, 8); return v; }
void put_str(File& f, const std::string& s) { put_u64(f, s.size()); f.put(s.data(), s.size()); }
std::string get_str(File& f, uint64_t limit) {
    const uint64_t n = get_u64(f);
    if (n > limit) throw Failure(err_arg(f.path + ": a string is longer than the file"));
    std::string s(n, '\0'); f.get(s.data(), n); return s;
}
void put_value(File& f, const vdb_value& v) {
    put_u64(f, (uint64_t)v.kind());
    std::visit([&](const auto& x) {
        using T = std::decay_t<decltype(x)>;
        if constexpr (std::is_same_v<T, int64_t>) f.put(&x, 8);
        else if constexpr (std::is_same_v<T, OF64>) f.put(&x.v, 8);
        else if constexpr (std::is_same_v<T, std::string>) put_str(f, x);
        else if constexpr (std::is_same_v<T, Bool8>) put_u64(f, x.v != 0);
        else if constexpr (std::is_same_v<T, std::vector<int64_t>>) { put_u64(f, x.size()); f.put(x.data(), x.size() * 8); }
        else if constexpr (std::is_same_v<T, std::vector<OF64>>) { put_u64(f, x.size()); for (auto& e : x) f.put(&e.v, 8); }
        else if constexpr (std::is_same_v<T, std::vector<std::string>>) { put_u64(f, x.size()); for (auto& e : x) put_str(f, e); }
        else { put_u64(f, x.size()); std::vector<uint8_t> b(x.size()); for (size_t i = 0; i < x.size(); ++i) b[i] = x[i].v != 0; f.put(b.data(), b.size()); }
    }, v.d);
}
vdb_value get_value(File& f, uint64_t limit) {
    vdb_value v;
    const uint64_t kind = get_u64(f);
    auto count = [&](uint64_t each) { const uint64_t n = get_u64(f); if (n > limit / each) throw Failure(err_arg(f.path + ": a list is longer than the file")); return n; };
    switch (kind) {
        case VDB_T_INT: { int64_t x; f.get(&x, 8); v.d = x; break; }
        case VDB_T_FLOAT: { double x; f.get(&x, 8); v.d = OF64{x}; break; }
        case VDB_T_STR: v.d = get_str(f, limit); break;
        case VDB_T_BOOL: v.d = Bool8{(uint8_t)(get_u64(f) != 0)}; break;
        case VDB_T_LIST_INT: { std::vector<int64_t> l(count(8)); f.get(l.data(), l.size() * 8); v.d = std::move(l); break; }
        case VDB_T_LIST_FLOAT: { std::vector<OF64> l(count(8)); for (auto& e : l) f.get(&e.v, 8); v.d = std::move(l); break; }
        case VDB_T_LIST_STR: { std::vector<std::string> l(count(8)); for (auto& e : l) e = get_str(f, limit); v.d = std::move(l); break; }
        case VDB_T_LIST_BOOL: { std::vector<uint8_t> b(count(1)); f.get(b.data(), b.size()); std::vector<Bool8> l(b.size()); for (size_t i = 0; i < b.size(); ++i) l[i].v = b[i] != 0; v.d = std::move(l); break; }
        default: throw Failure(err_arg(f.path + ": unknown payload value kind " + std::to_string(kind)));
    }
    return v;
}
}  // namespace
}  // extern "C++"
vdb_status vdb_index_save(const vdb_index* x, const char* path) {
    return guard([&] {
        need(x, "index"); need(path, "path");
        File out(path, "wb");
        write_index(out, x);
        out.close();
    });
}
vdb_status vdb_index_load(vdb_index* x, const char* path) {
    return guard([&] {
        need(x, "index"); need(path, "path");
        File in(path, "rb");
        IndexImage img; img.read(in, x);
        img.apply(x);
    });
}
vdb_status vdb_segment_save(const vdb_segment* s, const char* path) {
    return guard([&] {
        need(s, "segment"); need(path, "path");
        File out(path, "wb");
        write_index(out, s->hnsw.get());
        out.put(SEGMENT_MAGIC, 8);
        put_u64(out, s->next_id); put_u64(out, s->deleted.size()); put_u64(out, s->payloads.size());
        for (uint64_t id : s->deleted) put_u64(out, id);
        std::vector<uint64_t> order; order.reserve(s->payloads.size());
        for (auto& kv : s->payloads) order.push_back(kv.first);
        std::sort(order.begin(), order.end());
        for (uint64_t id : order) {
            const vdb_payload& p = s->payloads.at(id);
            put_u64(out, id); put_u64(out, p.fields.size());
            for (auto& f : p.fields) { put_str(out, f.first); put_value(out, f.second); }
        }
        out.close();
    });
}
vdb_status vdb_segment_load(vdb_segment* s, const char* path) {
    return guard([&] {
        need(s, "segment"); need(path, "path");
        File in(path, "rb");
        IndexImage img; img.read(in, s->hnsw.get());
        char magic[8]; in.get(magic, 8);
        if (std::memcmp(magic, SEGMENT_MAGIC, 8) != 0) throw Failure(err_arg(in.path + " holds an index but no segment section"));
        const uint64_t next_id = get_u64(in), n_del = get_u64(in), n_pay = get_u64(in);
        if (n_del > img.file_bytes / 8 || n_pay > img.file_bytes / 16) throw Failure(err_arg(in.path + ": implausible segment counts"));
        std::unordered_map<uint64_t, uint32_t> known; known.reserve(img.ids.size() * 2);
        for (size_t i = 0; i < img.ids.size(); ++i) known.emplace(img.ids[i], (uint32_t)i);
        std::set<uint64_t> del;
        for (uint64_t i = 0; i < n_del; ++i) {
            const uint64_t id = get_u64(in);
            auto k = known.find(id);
            if (k == known.end() || !img.del[k->second]) throw Failure(err_arg(in.path + ": deleted id " + std::to_string(id) + " is not a tombstone of the index"));
            del.insert(id);
        }
        std::unordered_map<uint64_t, vdb_payload> pays; pays.reserve(n_pay * 2);
        for (uint64_t i = 0; i < n_pay; ++i) {
            const uint64_t id = get_u64(in), nf = get_u64(in);
            if (nf > img.file_bytes / 16) throw Failure(err_arg(in.path + ": implausible field count"));
            vdb_payload p;
            for (uint64_t k = 0; k < nf; ++k) { std::string key = get_str(in, img.file_bytes); p.set(key, get_value(in, img.file_bytes)); }
            if (!pays.emplace(id, std::move(p)).second) throw Failure(err_arg(in.path + ": payload of id " + std::to_string(id) + " occurs twice"));
        }
        for (uint64_t id : img.ids) if (id >= next_id) throw Failure(err_arg(in.path + ": next_id does not lie above the ids in use"));
        // everything is read and checked: take over (a bad list still leaves index and segment as they were)
        img.apply(s->hnsw.get());
        s->payloads.swap(pays); s->deleted.swap(del); s->next_id = next_id;
        s->payload_index = vdb_payload_index();
        for (auto& kv : s->payloads) if (!s->deleted.count(kv.first) && known.count(kv.first)) s->payload_index.insert(kv.first, kv.second);   // stores.rs:18-52: live points only
        { std::lock_guard<std::mutex> lk(s->filter_mu); s->filter_cache.clear(); }
        ++s->payload_version;
        s->rebuild_columns();
    });
}
vdb_status vdb_index_device_info(const vdb_index* x, size_t* n_slots, size_t* row_stride_floats, size_t* l0_edges, size_t* upper_edges, size_t* bytes_total) {
    return guard([&] {
        need(x, "index");
        auto* h = const_cast<vdb_index*>(x);
        { std::lock_guard<std::mutex> lk(h->mu); h->sync_device(); }
        if (n_slots) *n_slots = h->len();
        if (row_stride_floats) *row_stride_floats = h->stride;
        if (l0_edges) *l0_edges = h->l0_edges;
        if (upper_edges) *upper_edges = h->upper_edges;
        if (bytes_total) *bytes_total = h->device_bytes;
    });
}
vdb_status vdb_index_sync_device(vdb_index* x) { return guard([&] { need(x, "index"); std::lock_guard<std::mutex> lk(x->mu); x->sync_device(); }); }

// ---- segment.rs ----
vdb_status vdb_segment_new(vdb_index* hnsw, vdb_segment** out) { return guard([&] { need(hnsw, "hnsw"); need(out, "out"); *out = new vdb_segment(hnsw); }); }
void vdb_segment_free(vdb_segment* s) { delete s; }
vdb_status vdb_segment_insert(vdb_segment* s, const float* v, size_t len, const vdb_payload* p, uint64_t* out_id) {
    return guard([&] { need(s, "segment"); need(v, "vector"); uint64_t id = s->insert(v, len, p); if (out_id) *out_id = id; });
}
vdb_status vdb_segment_insert_batch(vdb_segment* s, const float* v, size_t n, size_t len, const vdb_payload* const* p, uint64_t* out_ids) {
    return guard([&] { need(s, "segment"); if (n) need(v, "vectors"); s->insert_batch(v, n, len, p, out_ids); });
}
vdb_status vdb_segment_insert_batch_columns(vdb_segment* s, const float* v, size_t n, size_t len, size_t n_keys, const char* const* keys,
                                            const int32_t* kinds, const void* const* cols, uint64_t* out_ids) {
    return guard([&] {
        need(s, "segment"); if (n) need(v, "vectors");
        if (n_keys) { need(keys, "keys"); need(kinds, "kinds"); need(cols, "cols"); }
        for (size_t j = 0; j < n_keys; ++j) {
            need(keys[j], "keys[j]"); need(cols[j], "cols[j]");
            if (kinds[j] != VDB_T_INT && kinds[j] != VDB_T_FLOAT && kinds[j] != VDB_T_BOOL) throw Failure(err_arg("payload columns hold Int, Float or Bool values"));
        }
        std::vector<vdb_payload> pays(n_keys ? n : 0);
        for (size_t j = 0; j < n_keys; ++j) {
            const std::string key(keys[j]);
            for (size_t i = 0; i < n; ++i) {
                vdb_value val;
                if (kinds[j] == VDB_T_INT) val.d = static_cast<const int64_t*>(cols[j])[i];
                else if (kinds[j] == VDB_T_FLOAT) val.d = OF64{static_cast<const double*>(cols[j])[i]};
                else val.d = Bool8{(uint8_t)(static_cast<const uint8_t*>(cols[j])[i] != 0)};
                pays[i].set(key, val);
            }
        }
        std::vector<const vdb_payload*> ptrs(pays.size());
        for (size_t i = 0; i < pays.size(); ++i) ptrs[i] = &pays[i];
        s->insert_batch(v, n, len, n_keys ? ptrs.data() : nullptr, out_ids);
    });
}
vdb_status vdb_segment_get_vector(const vdb_segment* s, uint64_t id, float* out, size_t cap) {
    return guard([&] {
        need(s, "segment"); need(out, "out");
        const float* v = s->get_vector(id);
        if (!v) throw Failure(err_not_found(id));
        if (cap < s->hnsw->dim) throw Failure(err_arg("capacity too small"));
        std::memcpy(out, v, s->hnsw->dim * 4);
    });
}
vdb_status vdb_segment_delete(vdb_segment* s, uint64_t id) { return guard([&] { need(s, "segment"); s->del(id); }); }
vdb_status vdb_segment_search(const vdb_segment* s, const float* q, size_t len, size_t top_k, vdb_scored_point* out, size_t* out_n) {
    return guard([&] { need(s, "segment"); need(q, "query"); emit(s->search(q, len, top_k), out, out_n); });
}
vdb_status vdb_segment_search_with_filter(const vdb_segment* s, const float* q, size_t len, size_t top_k, const vdb_filter* f, vdb_scored_point* out, size_t* out_n) {
    return guard([&] { need(s, "segment"); need(q, "query"); emit(s->search_with_filter(q, len, top_k, f), out, out_n); });
}
vdb_status vdb_segment_search_unfiltered(const vdb_segment* s, const float* q, size_t len, size_t top_k, vdb_scored_point* out, size_t* out_n) {
    if (!s) { set_last_error("segment is NULL"); return VDB_ERR_ARGUMENT; }
    return vdb_index_search(s->hnsw.get(), q, len, top_k, out, out_n);
}
vdb_status vdb_segment_post_filter(const vdb_segment* s, const float* q, size_t len, size_t top_k, const vdb_filter* f, vdb_scored_point* out, size_t* out_n) {
    return guard([&] { need(s, "segment"); need(q, "query"); emit(s->post_filter(q, len, top_k, f), out, out_n); });
}
vdb_status vdb_segment_search_batch(const vdb_segment* s, const float* q, size_t nq, size_t len, size_t top_k, size_t ef_search, const vdb_filter* f,
                                    vdb_scored_point* out, uint32_t* out_n) {
    return guard([&] { need(s, "segment"); if (nq) { need(q, "queries"); need(out_n, "out_n"); } s->search_batch(q, nq, len, top_k, ef_search, f, out, out_n); });
}
const vdb_payload* vdb_segment_get_payload(const vdb_segment* s, uint64_t id) { auto it = s->payloads.find(id); return it == s->payloads.end() ? nullptr : &it->second; }
int vdb_segment_is_deleted(const vdb_segment* s, uint64_t id) { return s->deleted.count(id) != 0; }
vdb_status vdb_segment_purge(vdb_segment* s) { return guard([&] { need(s, "segment"); s->purge(); }); }
const vdb_index* vdb_segment_hnsw(const vdb_segment* s) { return s->hnsw.get(); }
vdb_index* vdb_segment_hnsw_mut(vdb_segment* s) { return s->hnsw.get(); }
const vdb_payload_index* vdb_segment_payload_index(const vdb_segment* s) { return &s->payload_index; }
size_t vdb_segment_payload_count(const vdb_segment* s) { return s->payloads.size(); }
vdb_status vdb_segment_in_place_filtered_search(const vdb_segment* s, const float* q, size_t len, size_t top_k, const vdb_filter* f, vdb_scored_point* out, size_t* out_n) {
    return guard([&] { need(s, "segment"); need(q, "query"); emit(s->in_place_filtered_search(q, len, top_k, f), out, out_n); });
}
vdb_status vdb_segment_compile_filter(const vdb_segment* s, const vdb_filter* f, size_t* n_pass, size_t* n_err) {
    return guard([&] { need(s, "segment"); need(f, "filter"); const auto b = s->compile(*f); if (n_pass) *n_pass = b->n_pass; if (n_err) *n_err = b->n_err; });
}
vdb_status vdb_segment_filter_bitmaps(const vdb_segment* s, const vdb_filter* f, uint32_t* pass, uint32_t* err, size_t cap_words) {
    return guard([&] {
        need(s, "segment"); need(f, "filter"); need(pass, "pass"); need(err, "err");
        if (cap_words < (s->hnsw->len() + 31) / 32) throw Failure(err_arg("capacity too small"));
        s->filter_bitmaps(*f, pass, err);
    });
}
vdb_status vdb_segment_search_batch_dev(const vdb_segment* s, const float* d_q, size_t nq, size_t top_k, size_t ef_search, const vdb_filter* f,
                                        vdb_scored_point* d_out, uint32_t* d_out_n, vdb_query_stats* d_stats, void* stream) {
    return guard([&] {
        need(s, "segment");
        if (nq == 0) return;
        need(d_q, "d_queries"); need(d_out, "d_out"); need(d_out_n, "d_out_n");
        s->search_batch_dev(d_q, nq, top_k, ef_search, f, d_out, d_out_n, d_stats, (hipStream_t)stream);
    });
}

// ---- scale-out ----
vdb_status vdb_topk_merge_dev(const vdb_scored_point* d_lists, const uint32_t* d_counts, size_t n_shards, size_t nq, size_t k,
                              vdb_scored_point* d_out, uint32_t* d_out_n, void* stream) {
    return guard([&] {
        need(d_lists, "d_lists"); need(d_counts, "d_counts"); need(d_out, "d_out"); need(d_out_n, "d_out_n");
        launch_topk_merge(d_lists, d_counts, (uint32_t)n_shards, (uint32_t)nq, (uint32_t)k, d_out, d_out_n, (hipStream_t)stream);
    });
}

}  // extern "C"
