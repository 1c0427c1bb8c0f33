/*
 * vdbmi.h — C ABI of libvdbmi, the MI355X (gfx950) search engine that stands in for the search
 * path of rosharma719/vectordb.
 *
 * The reference has no FFI of its own: its boundary is the public Rust API of the crate
 * (src/lib.rs:1-4).  Every entry point below names the Rust item it replaces (file:line in the
 * reference tree); a Rust wrapper crate re-creates the original signatures on top of these calls.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller allocates every output buffer;
 *   - every call returns a vdb_status; the Display text of the matching DBError
 *     (src/utils/errors.rs:9-33) is available from vdb_last_error_message() on the same thread;
 *   - where the reference panics (score() on unequal lengths, metric.rs:5; NaN in a heap compare,
 *     hnsw.rs:36,55) the ABI returns VDB_ERR_PANIC and the wrapper re-raises;
 *   - handles are not synchronised for mutation; const searches on one handle may run from
 *     several host threads;
 *   - all distance arithmetic runs on the GPU; there is no CPU implementation behind this ABI
 *     and the library fails with VDB_ERR_DEVICE when no gfx950 device is usable.
 *
 * Determinism contract (the reference is non-deterministic where it uses rand::rng() and
 * HashMap/HashSet iteration order; this library fixes those choices):
 *   DET-1  "first non-deleted key" (hnsw.rs:115,161,441)            -> smallest live id
 *   DET-2  choose_multiple source order (hnsw.rs:233-237)           -> ascending id, reservoir
 *          sampling of rand 0.9 `IteratorRandom::choose_multiple` driven by the counter RNG
 *   DET-3  payload key order for filter_keys (segment.rs:40-44)     -> ascending byte order
 *   DET-4  posting-set order for entry hints (hnsw.rs:566-570)      -> ascending id
 *   DET-5  extra_neighbors iteration (hnsw.rs:320)                  -> ascending id
 *   DET-6  iter_vectors order (hnsw.rs:275, segment.rs:178)         -> ascending id
 *   RNG    draw(seed, stream, n) = splitmix64(splitmix64(seed ^ stream*0xD1B54A32D192ED03) + n);
 *          level = min(floor(-ln((draw>>11) * 2^-53) / ln(M)), max_level_cap)   (stream 1, hnsw.rs:96-101)
 *          reservoir index k = draw % (i + 1 + 100)                           (stream 2)
 *   TIES   equal sort keys are ordered by arrival (stable); the reference's order among equal
 *          keys is whatever std::collections::BinaryHeap yields.
 */
#ifndef VDBMI_H
#define VDBMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- types ------------------- */

/* DistanceMetric, src/utils/types.rs:24-29 (declaration order) */
typedef enum { VDB_COSINE = 0, VDB_DOT = 1, VDB_EUCLIDEAN = 2 } vdb_metric;

/* DBError, src/utils/errors.rs:9-33 (declaration order, shifted by one for OK) */
typedef enum {
    VDB_OK = 0,
    VDB_ERR_NOT_FOUND = 1,
    VDB_ERR_VECTOR_LENGTH_MISMATCH = 2,
    VDB_ERR_IO = 3,
    VDB_ERR_SERIALIZATION = 4,
    VDB_ERR_WAL_CORRUPT = 5,
    VDB_ERR_INVALID_PAYLOAD = 6,
    VDB_ERR_SEARCH = 7,
    VDB_ERR_PANIC = 99,      /* the reference would have panicked */
    VDB_ERR_DEVICE = 100,    /* HIP runtime / no usable GPU */
    VDB_ERR_ARGUMENT = 101   /* NULL pointer, capacity too small, unsupported size */
} vdb_status;

/* ScalarComparisonOp, src/utils/payload.rs:131-139 */
typedef enum { VDB_OP_EQ = 0, VDB_OP_NEQ = 1, VDB_OP_LT = 2, VDB_OP_LTE = 3, VDB_OP_GT = 4, VDB_OP_GTE = 5 } vdb_cmp_op;

/* PayloadValue discriminants, src/utils/payload.rs:8-18 */
typedef enum {
    VDB_T_INT = 0, VDB_T_FLOAT = 1, VDB_T_STR = 2, VDB_T_BOOL = 3,
    VDB_T_LIST_INT = 4, VDB_T_LIST_FLOAT = 5, VDB_T_LIST_STR = 6, VDB_T_LIST_BOOL = 7
} vdb_value_kind;

/* ListQueryOp, src/utils/payload.rs:141-147 */
typedef enum { VDB_LQ_CONTAINS = 0, VDB_LQ_EQUALS = 1, VDB_LQ_LENGTH = 2, VDB_LQ_ELEMENT_COMPARE = 3 } vdb_list_query;

/* ScoredPoint, src/vector/hnsw.rs:11-16 */
typedef struct { uint64_t id; float raw_score; float sort_key; } vdb_scored_point;

/* Arithmetic of the distance kernels.
 *   STRICT: one lane sums one row left to right with separate multiply and add, sqrt and divide
 *           correctly rounded — bit-identical to metric.rs:15-35.
 *   FAST:   lanes of a wavefront split a row and combine partial sums with a DPP butterfly
 *           (fused multiply-add allowed); distances agree with STRICT to <= 1e-5 relative. */
typedef enum { VDB_ARITH_FAST = 0, VDB_ARITH_STRICT = 1 } vdb_arith;

/* Filtered-search semantics.
 *   LITERAL: hnsw.rs:585-689 as written (inverted result heap, entry admitted unfiltered,
 *            results worst-first).
 *   MASKED:  extension — the unfiltered ef-beam drives the traversal, a device bitmap masks
 *            which scored nodes enter the result list; results best-first. */
typedef enum { VDB_FILTER_LITERAL = 0, VDB_FILTER_MASKED = 1 } vdb_filter_mode;

typedef struct {
    uint64_t seed;        /* counter RNG seed (see RNG above) */
    int32_t  device;      /* HIP device ordinal */
    int32_t  arith;       /* vdb_arith */
    int32_t  filter_mode; /* vdb_filter_mode */
    int32_t  beam_width;  /* FAST arithmetic only: nodes expanded per step of the layer-0 beam, 1..4;
                           * 0 = library default (4).  1 expands strictly best-first like
                           * hnsw.rs:448-505; wider steps expand the best few unexpanded entries
                           * together (same ef, same termination; recall equal or higher, ids may
                           * differ where the order of expansion matters).  STRICT always uses 1. */
} vdb_options;

/* Per-query work counters written by the search kernels (extension; the reference has none). */
typedef struct { uint32_t rows_scored; uint32_t expanded; uint32_t adj_entries; uint32_t flags; } vdb_query_stats;

typedef struct vdb_value vdb_value;                 /* PayloadValue        payload.rs:8-18   */
typedef struct vdb_payload vdb_payload;             /* Payload             payload.rs:21     */
typedef struct vdb_filter vdb_filter;               /* Filter              filters.rs:5-19   */
typedef struct vdb_payload_index vdb_payload_index; /* PayloadIndex        stores.rs:6-8     */
typedef struct vdb_index vdb_index;                 /* HNSWIndex           hnsw.rs:59-73     */
typedef struct vdb_segment vdb_segment;             /* Segment             segment.rs:11-18  */

/* ---------------------------------------------------------------- library ----------------- */

/* Display text of the last error on this thread (thiserror strings, errors.rs:10-32). */
const char* vdb_last_error_message(void);
/* Number of usable gfx950 devices; 0 when the HIP runtime reports none. */
int vdb_device_count(void);
void vdb_default_options(vdb_options* out);

/* ---------------------------------------------------------------- metric.rs --------------- */

/* score(a, b, metric), metric.rs:4-12.  na != nb -> VDB_ERR_PANIC "Vectors must be the same length". */
vdb_status vdb_score(const float* a, size_t na, const float* b, size_t nb, vdb_metric metric, float* out);

/* Fused streaming scan (extension): out[q*n + i] = score(queries[q], base[i]) for host buffers. */
vdb_status vdb_score_scan(const float* queries, size_t nq, const float* base, size_t n, size_t dim,
                          vdb_metric metric, vdb_arith arith, int device, float* out);
/* Same on device-resident buffers: queries dense (dim floats each), base rows `stride` floats apart
 * (FAST needs 16-byte aligned rows zero-padded to a multiple of 4 floats); enqueued on `stream`. */
vdb_status vdb_score_scan_dev(const float* d_queries, size_t nq, const float* d_base, size_t n, size_t dim, size_t stride,
                              vdb_metric metric, vdb_arith arith, float* d_out, void* stream);

/* ---------------------------------------------------------------- payload.rs -------------- */

vdb_value* vdb_value_int(int64_t v);
vdb_value* vdb_value_float(double v);
vdb_value* vdb_value_str(const char* s, size_t len);
vdb_value* vdb_value_bool(int v);
vdb_value* vdb_value_list_int(const int64_t* v, size_t n);
vdb_value* vdb_value_list_float(const double* v, size_t n);
vdb_value* vdb_value_list_str(const char* const* s, const size_t* lens, size_t n);
vdb_value* vdb_value_list_bool(const uint8_t* v, size_t n);
void vdb_value_free(vdb_value* v);
int vdb_value_type(const vdb_value* v);
/* derived PartialEq, payload.rs:8 */
int vdb_value_equals(const vdb_value* a, const vdb_value* b);
/* PayloadValue::compare_scalar, payload.rs:150-186.  *is_some = 0 stands for None. */
vdb_status vdb_value_compare_scalar(const vdb_value* a, vdb_cmp_op op, const vdb_value* b, int* is_some, int* out);
/* PayloadValue::evaluate_list_query, payload.rs:188-229.  `n` is the length (LENGTH) or the
 * element index (ELEMENT_COMPARE); `v` is unused for LENGTH. */
vdb_status vdb_value_evaluate_list_query(const vdb_value* self, vdb_list_query kind, vdb_cmp_op op, size_t n,
                                         const vdb_value* v, int* is_some, int* out);

vdb_payload* vdb_payload_new(void);                                             /* Payload::default, payload.rs:255-259 */
void vdb_payload_free(vdb_payload* p);
vdb_status vdb_payload_set(vdb_payload* p, const char* key, const vdb_value* v); /* Payload::set, payload.rs:27-29 (copies v) */
const vdb_value* vdb_payload_get(const vdb_payload* p, const char* key);        /* Payload::get, payload.rs:32-34 (borrowed) */
/* Payload::compare_field, payload.rs:36-114 */
vdb_status vdb_payload_compare_field(const vdb_payload* p, const char* field, vdb_cmp_op op, const vdb_value* other, int* out);
/* Payload::evaluate_list_field, payload.rs:117-128 */
vdb_status vdb_payload_evaluate_list_field(const vdb_payload* p, const char* field, vdb_list_query kind, vdb_cmp_op op,
                                           size_t n, const vdb_value* v, int* out);

/* ---------------------------------------------------------------- filters.rs -------------- */

vdb_filter* vdb_filter_match(const char* key, const vdb_value* v);                   /* Filter::Match,   filters.rs:7-10  */
vdb_filter* vdb_filter_compare(const char* key, vdb_cmp_op op, const vdb_value* v);  /* Filter::Compare, filters.rs:11-15 */
vdb_filter* vdb_filter_and(const vdb_filter* const* children, size_t n);             /* Filter::And,     filters.rs:16 (copies) */
vdb_filter* vdb_filter_or(const vdb_filter* const* children, size_t n);              /* Filter::Or,      filters.rs:17 */
vdb_filter* vdb_filter_not(const vdb_filter* child);                                 /* Filter::Not,     filters.rs:18 */
void vdb_filter_free(vdb_filter* f);
/* evaluate_filter, filters.rs:23-75 */
vdb_status vdb_filter_evaluate(const vdb_filter* f, const vdb_payload* p, int* out);

/* ---------------------------------------------------------------- stores.rs --------------- */

vdb_payload_index* vdb_payload_index_new(void);                                          /* stores.rs:11-15 */
void vdb_payload_index_free(vdb_payload_index* x);
vdb_status vdb_payload_index_insert(vdb_payload_index* x, uint64_t id, const vdb_payload* p);   /* stores.rs:18-31 */
vdb_status vdb_payload_index_remove(vdb_payload_index* x, uint64_t id, const vdb_payload* p);   /* stores.rs:34-52 */
/* query_exact, stores.rs:55-61.  *is_some = 0 for None.  Ids ascending; *out_n is the full count
 * even when it exceeds cap (call again with a larger buffer). */
vdb_status vdb_payload_index_query_exact(const vdb_payload_index* x, const char* key, const vdb_value* v,
                                         uint64_t* out_ids, size_t cap, size_t* out_n, int* is_some);
/* all_for_key, stores.rs:74-82 */
vdb_status vdb_payload_index_all_for_key(const vdb_payload_index* x, const char* key,
                                         uint64_t* out_ids, size_t cap, size_t* out_n, int* is_some);

/* ---------------------------------------------------------------- hnsw.rs ----------------- */

/* HNSWIndex::new, hnsw.rs:77-94.  opts may be NULL (defaults: seed 0, device 0, FAST, LITERAL). */
vdb_status vdb_index_new(vdb_metric metric, size_t m, size_t ef, size_t max_level_cap, size_t dim,
                         const vdb_options* opts, vdb_index** out);
void vdb_index_free(vdb_index* x);
/* HNSWIndex::insert, hnsw.rs:119-208: duplicate id is a silent OK; wrong length -> VECTOR_LENGTH_MISMATCH. */
vdb_status vdb_index_insert(vdb_index* x, uint64_t id, const float* v, size_t len);
/* Batched build (extension): the points of one call search the graph as it stood before the
 * call (on the GPU) and are then linked in order with the rules of hnsw.rs:175-205. */
vdb_status vdb_index_insert_batch(vdb_index* x, const uint64_t* ids, const float* v, size_t n, size_t len);
/* HNSWIndex::search, hnsw.rs:513-554. */
vdb_status vdb_index_search(const vdb_index* x, const float* q, size_t len, size_t top_k,
                            vdb_scored_point* out, size_t* out_n);
/* Batched search (extension; the GPU's native shape).  queries is nq x len row-major on the host;
 * out is nq x top_k, out_n[q] results are valid in row q.  ef_search = 0 uses the index ef. */
vdb_status vdb_index_search_batch(const vdb_index* x, const float* queries, size_t nq, size_t len, size_t top_k,
                                  size_t ef_search, vdb_scored_point* out, uint32_t* out_n, vdb_query_stats* stats);
/* Same with queries and results resident in device memory, enqueued on `stream` (hipStream_t,
 * NULL = the index's own stream) without a host synchronisation.  d_stats may be NULL; pass it when
 * ef_search exceeds 256: only that path can run out of visited-table space, which it reports as
 * flags & 1 in d_stats with out_n = 0 for the query (the host-buffer call repeats such a query by itself). */
vdb_status vdb_index_search_batch_dev(const vdb_index* x, const float* d_queries, size_t nq, size_t top_k,
                                      size_t ef_search, vdb_scored_point* d_out, uint32_t* d_out_n,
                                      vdb_query_stats* d_stats, void* stream);
/* Exact k-NN over all live points (extension, no reference function; defined by score() +
 * normalize_score() + (sort_key asc, id asc)): MFMA contraction + fused top-k + strict re-rank.
 * 1 <= top_k <= 256 (0 returns nothing); a larger top_k is VDB_ERR_ARGUMENT.  Fewer live points than top_k: all of them. */
vdb_status vdb_index_exact_knn_batch(const vdb_index* x, const float* queries, size_t nq, size_t len, size_t top_k,
                                     vdb_scored_point* out, uint32_t* out_n);
vdb_status vdb_index_exact_knn_batch_dev(const vdb_index* x, const float* d_queries, size_t nq, size_t top_k,
                                         vdb_scored_point* d_out, uint32_t* d_out_n, void* stream);
/* The same as vdb_index_exact_knn_batch_dev with NO allocation (no counterpart in the reference; for callers that
 * cannot allocate: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  The
 * candidate pass works in a 16 MiB scratch array that is part of the library's code object (one copy per GPU, held
 * until the process ends).  Limits: FAST index, 1 <= top_k <= 28, nq < 2^32,
 * K + 2 * rows + nq * (K + 2 + (top_k <= 12 ? 17 : 33)) + 8 <= 4 Mi words with K the padded row length (64, 128, 256, else the
 * next multiple of 64), and no edit pending (after inserts or mark_deleted call vdb_index_sync_device first); anything
 * else — an empty index and top_k = 0 included, which the allocating entry points answer with zero counts — is
 * VDB_ERR_ARGUMENT and nothing is launched.  The call makes no allocation; it does query the device (current device,
 * occupancy, the scratch symbol's address on first use), so under graph capture make one call before the capture begins.
 * Every call on one GPU shares the scratch array: calls that overlap on one GPU (any index, any stream) race, so order
 * them on one stream or with events. */
vdb_status vdb_index_exact_knn_batch_dev_noalloc(const vdb_index* x, const float* d_queries, size_t nq, size_t top_k,
                                                 vdb_scored_point* d_out, uint32_t* d_out_n, void* stream);

/* hnsw.rs:111-117.  An id the index does not hold is remembered (the reference's HashSet takes any
 * id, hnsw.rs:112): a point inserted under it later is a tombstone from birth. */
vdb_status vdb_index_mark_deleted(vdb_index* x, uint64_t id);
int    vdb_index_contains(const vdb_index* x, uint64_t id);                      /* hnsw.rs:691-693 */
size_t vdb_index_len(const vdb_index* x);                                        /* hnsw.rs:695-697 */
int    vdb_index_metric(const vdb_index* x);                                     /* hnsw.rs:707-709 */
size_t vdb_index_m(const vdb_index* x);                                          /* hnsw.rs:711-713 */
size_t vdb_index_ef(const vdb_index* x);                                         /* hnsw.rs:715-717 */
size_t vdb_index_max_level_cap(const vdb_index* x);                              /* hnsw.rs:719-721 */
size_t vdb_index_dim(const vdb_index* x);                                        /* hnsw.rs:723-725 */
size_t vdb_index_current_max_level(const vdb_index* x);                          /* hnsw.rs:740-742 */
/* get_entry_point, hnsw.rs:736-738: returns 0 and leaves *out alone when there is none. */
int    vdb_index_get_entry_point(const vdb_index* x, uint64_t* out);
vdb_status vdb_index_set_entry_point(vdb_index* x, uint64_t id);                 /* hnsw.rs:744-746 */
vdb_status vdb_index_set_current_max_level(vdb_index* x, size_t level);          /* hnsw.rs:748-750 */
/* get_vector, hnsw.rs:727-734 (stored = normalised for Cosine).  NOT_FOUND for unknown/deleted. */
vdb_status vdb_index_get_vector(const vdb_index* x, uint64_t id, float* out, size_t cap);
/* layer_neighbors, hnsw.rs:699-701.  *is_some = 0 for None; literal list incl. self and duplicates. */
vdb_status vdb_index_layer_neighbors(const vdb_index* x, size_t level, uint64_t id,
                                     uint64_t* out_ids, size_t cap, size_t* out_n, int* is_some);
/* maybe_normalize, hnsw.rs:752-764 */
vdb_status vdb_index_maybe_normalize(const vdb_index* x, const float* v, size_t len, float* out);
/* normalize_score, hnsw.rs:103-108 */
float  vdb_index_normalize_score(const vdb_index* x, float raw);
/* add_bidirectional_edge, hnsw.rs:328-331 */
vdb_status vdb_index_add_bidirectional_edge(vdb_index* x, size_t level, uint64_t a, uint64_t b);
/* greedy_search_layer_unfiltered, hnsw.rs:333-368 */
vdb_status vdb_index_greedy_search_layer_unfiltered(const vdb_index* x, const float* q, size_t len, uint64_t entry,
                                                    size_t level, uint64_t* out_id);

/* Graph hand-over (extension).  Import replaces the index contents with a graph built elsewhere
 * (the "reference-built graph" of the north-star path); export returns the literal lists.
 *   ids[n], levels[n], vectors[n*dim] (stored form), deleted[n] (0/1, may be NULL),
 *   per layer l in [0, n_layers): owners (slot indices that own a list at l), offsets
 *   (len owners+1) into a flat neighbour array of slot indices.
 * Everything is validated before the index is touched (ids distinct, owners distinct per layer and
 * below n, offsets as described at vdb_layer_csr, neighbours below n, entry id known): a bad graph
 * is VDB_ERR_ARGUMENT and leaves the index as it was. */
typedef struct {
    size_t n_lists; const uint32_t* owners; const uint64_t* offsets; const uint32_t* neighbors;
    size_t n_edges;      /* length of `neighbors`; offsets[0] == 0, offsets ascend, offsets[n_lists] == n_edges */
} vdb_layer_csr;
vdb_status vdb_index_import_graph(vdb_index* x, size_t n, const uint64_t* ids, const uint32_t* levels,
                                  const float* vectors, const uint8_t* deleted,
                                  size_t n_layers, const vdb_layer_csr* layers,
                                  int has_entry, uint64_t entry_id, size_t current_max_level);
/* Sizes first (arrays NULL), then data.  Layer l needs n_lists[l] owners, n_lists[l]+1 offsets
 * and n_edges[l] neighbours. */
vdb_status vdb_index_export_sizes(const vdb_index* x, size_t* n, size_t* n_layers, size_t* n_lists, size_t* n_edges, size_t cap_layers);
vdb_status vdb_index_export_points(const vdb_index* x, uint64_t* ids, uint32_t* levels, float* vectors, uint8_t* deleted);
vdb_status vdb_index_export_layer(const vdb_index* x, size_t level, uint32_t* owners, uint64_t* offsets, uint32_t* neighbors);
/* Graph file (extension; the reference keeps nothing on disk, README.md:32).  One little-endian
 * file, every array padded to a multiple of 8 bytes:
 *   header, 112 bytes: char magic[8] = "VDBGRAF1"; u32 version = 1; u32 metric; u64 m, ef,
 *     max_level_cap, dim, n, n_layers; u64 has_entry, entry_id, current_max_level; u64 level_draws,
 *     sample_draws (positions of the two counter-based random streams); u64 seed
 *   u64 ids[n]; u32 levels[n]; u8 deleted[n]; f32 vectors[n * dim] (stored form)
 *   per layer: u64 n_lists, n_edges; u32 owners[n_lists]; u64 offsets[n_lists + 1];
 *     u32 neighbors[n_edges]                       (the arrays of vdb_index_export_layer)
 * load replaces the contents of an index created with the same metric and dimension and takes over
 * the file's seed and stream positions, so that inserts after a load continue the saved build.
 * A file that cannot be opened, read or written is VDB_ERR_IO; one whose contents are inconsistent
 * is VDB_ERR_ARGUMENT and leaves the index as it was. */
vdb_status vdb_index_save(const vdb_index* x, const char* path);
vdb_status vdb_index_load(vdb_index* x, const char* path);
/* Segment file (extension; the state of segment.rs:11-18 on disk): the graph file of the segment's index, followed by
 *   char magic[8] = "VDBSEGM1"; u64 next_id, n_deleted, n_payloads; u64 deleted_ids[n_deleted] (ascending);
 *   per payload, ascending id: u64 id, n_fields; per field (ascending key): string key; u64 kind (vdb_value_kind); value:
 *     Int i64 | Float f64 | Bool u64 | Str string | ListInt u64 n + i64[n] | ListFloat u64 n + f64[n] |
 *     ListStr u64 n + n strings | ListBool u64 n + u8[n];  string = u64 length + bytes; everything padded to 8 bytes.
 * vdb_index_load reads the index part of such a file.  vdb_segment_load replaces the contents of a segment whose
 * index was created with the same metric and dimension: vectors, lists, tombstones, payloads, deleted set, next_id and
 * the random streams; the inverted index is rebuilt from the payloads of the live points.  Errors as for the graph file;
 * the segment is untouched unless the whole file was read and found consistent. */
vdb_status vdb_segment_save(const vdb_segment* s, const char* path);
vdb_status vdb_segment_load(vdb_segment* s, const char* path);
/* Device-side footprint of the packed graph actually searched (after self-loop / duplicate removal). */
vdb_status vdb_index_device_info(const vdb_index* x, size_t* n_slots, size_t* row_stride_floats,
                                 size_t* l0_edges, size_t* upper_edges, size_t* bytes_total);
/* Push pending host-side edits to the device and wait (searches do this lazily). */
vdb_status vdb_index_sync_device(vdb_index* x);

/* ---------------------------------------------------------------- segment.rs -------------- */

/* Segment::new, segment.rs:21-29 — takes ownership of the index (do not free it afterwards). */
vdb_status vdb_segment_new(vdb_index* hnsw, vdb_segment** out);
void vdb_segment_free(vdb_segment* s);
/* Segment::insert, segment.rs:32-61 (payload may be NULL); ids are 1, 2, 3 ... */
vdb_status vdb_segment_insert(vdb_segment* s, const float* v, size_t len, const vdb_payload* payload, uint64_t* out_id);
/* Batched insert (extension; payloads[i] may be NULL, payloads itself may be NULL). */
vdb_status vdb_segment_insert_batch(vdb_segment* s, const float* v, size_t n, size_t len,
                                    const vdb_payload* const* payloads, uint64_t* out_ids);
/* Batched insert with the payloads given as columns (extension; the bulk form of Payload::set, payload.rs:27-29):
 * every point gets the n_keys fields keys[j] -> cols[j][i]; kinds[j] is VDB_T_INT (cols[j] = int64_t[n]),
 * VDB_T_FLOAT (double[n]) or VDB_T_BOOL (uint8_t[n]).  Same result as vdb_segment_insert_batch with such payloads. */
vdb_status vdb_segment_insert_batch_columns(vdb_segment* s, const float* v, size_t n, size_t len, size_t n_keys,
                                            const char* const* keys, const int32_t* kinds, const void* const* cols, uint64_t* out_ids);
vdb_status vdb_segment_get_vector(const vdb_segment* s, uint64_t id, float* out, size_t cap);     /* segment.rs:64-69 */
vdb_status vdb_segment_delete(vdb_segment* s, uint64_t id);                                       /* segment.rs:71-99 */
vdb_status vdb_segment_search(const vdb_segment* s, const float* q, size_t len, size_t top_k,
                              vdb_scored_point* out, size_t* out_n);                             /* segment.rs:103-119 */
vdb_status vdb_segment_search_with_filter(const vdb_segment* s, const float* q, size_t len, size_t top_k,
                                          const vdb_filter* f, vdb_scored_point* out, size_t* out_n);   /* segment.rs:121-147 */
vdb_status vdb_segment_search_unfiltered(const vdb_segment* s, const float* q, size_t len, size_t top_k,
                                         vdb_scored_point* out, size_t* out_n);                  /* segment.rs:152-154 */
vdb_status vdb_segment_post_filter(const vdb_segment* s, const float* q, size_t len, size_t top_k,
                                   const vdb_filter* f, vdb_scored_point* out, size_t* out_n);   /* segment.rs:221-249 */
/* Batched Segment::search / search_with_filter (extension): f == NULL -> search. */
vdb_status vdb_segment_search_batch(const vdb_segment* s, const float* queries, size_t nq, size_t len, size_t top_k,
                                    size_t ef_search, const vdb_filter* f, vdb_scored_point* out, uint32_t* out_n);
const vdb_payload* vdb_segment_get_payload(const vdb_segment* s, uint64_t id);   /* segment.rs:157-159 (borrowed) */
int vdb_segment_is_deleted(const vdb_segment* s, uint64_t id);                   /* segment.rs:162-164 */
vdb_status vdb_segment_purge(vdb_segment* s);                                    /* segment.rs:166-217 */
const vdb_index* vdb_segment_hnsw(const vdb_segment* s);                         /* segment.rs:252-254 (borrowed) */
vdb_index* vdb_segment_hnsw_mut(vdb_segment* s);                                 /* test access to the owned index */
const vdb_payload_index* vdb_segment_payload_index(const vdb_segment* s);        /* segment.rs:261-263 (borrowed) */
size_t vdb_segment_payload_count(const vdb_segment* s);                          /* payloads().len(), segment.rs:257-259 */
/* HNSWIndex::in_place_filtered_search called on the segment's index with its payloads and
 * payload index (hnsw.rs:585-689); f may be NULL. */
vdb_status vdb_segment_in_place_filtered_search(const vdb_segment* s, const float* q, size_t len, size_t top_k,
                                                const vdb_filter* f, vdb_scored_point* out, size_t* out_n);
/* The same on device buffers (extension): queries, results and counts are device-resident, the work is
 * enqueued on `stream` (NULL = the index's own stream) and the call returns without waiting for it.  With a
 * filter the segment must be in VDB_FILTER_MASKED mode (the literal walk of hnsw.rs:585-689 picks its entry and
 * reports filter errors on the host). */
vdb_status vdb_segment_search_batch_dev(const vdb_segment* s, const float* d_queries, size_t nq, size_t top_k,
                                        size_t ef_search, const vdb_filter* f, vdb_scored_point* d_out, uint32_t* d_out_n,
                                        vdb_query_stats* d_stats /* nullable; zeros when the passing rows were scanned */, void* stream);
/* Compile a filter to its device bitmaps (pass / error bits per slot) and return the number of
 * passing live points (extension; searches do this on demand and cache the result).  The tree of
 * filters.rs:23-75 is evaluated on the GPU over the segment's payload columns: Match on a scalar scatters the
 * posting list of the inverted index (stores.rs:55-61) when it is short and scans the key's column otherwise,
 * Compare (payload.rs:36-114) scans the column, And / Or / Not combine bitmaps word by word with the
 * reference's short-circuit and Err propagation. */
vdb_status vdb_segment_compile_filter(const vdb_segment* s, const vdb_filter* f, size_t* n_pass, size_t* n_err);
/* Host copies of the compiled bitmaps, (len + 31) / 32 words each: bit i of word w belongs to the
 * (32 w + i)-th inserted point still held by the index (insertion order; after a purge: ascending id).
 * pass = evaluate_filter is Ok(true), err = it is Err; bits of tombstoned points are unspecified. */
vdb_status vdb_segment_filter_bitmaps(const vdb_segment* s, const vdb_filter* f, uint32_t* pass, uint32_t* err, size_t cap_words);

/* ---------------------------------------------------------------- scale-out --------------- */

/* k-way merge of per-shard top-k lists (extension; the reference is single-segment).
 * d_lists holds n_shards x nq x k records (shard-major) with d_counts[shard*nq + q] valid
 * entries per row, each row sorted by (sort_key asc, id asc); d_out receives nq x k merged rows. */
vdb_status vdb_topk_merge_dev(const vdb_scored_point* d_lists, const uint32_t* d_counts, size_t n_shards,
                              size_t nq, size_t k, vdb_scored_point* d_out, uint32_t* d_out_n, void* stream);

/* Sharded search (extension; the reference holds ONE in-memory segment, src/segment/segment.rs:11-18, and has no
 * distributed layer).  The base set is split into contiguous id ranges, one shard (an index or a segment with its own
 * graph) per rank and GPU.  A group is one RCCL communicator over the ranks; every rank searches the full query batch
 * against its shard, the per-shard top-k rows are exchanged with one all-gather (records, then counts) on the caller's
 * stream and merged by (sort_key asc, id asc) on every rank, so every rank receives the same nq x top_k rows.
 *
 * vdb_shard_group_unique_id: rank 0 draws the communicator's id (ncclUniqueId, 128 bytes) and hands it to the other
 * ranks by whatever channel launched them; vdb_shard_group_new is collective over the ranks (ncclCommInitRank) and
 * binds the calling rank to `device`.  Calls on one group are serialised (its staging buffers are shared); use one
 * group per stream to overlap exchanges.  vdb_shard_group_size / _rank report what the communicator itself says. */
typedef struct vdb_shard_group vdb_shard_group;
vdb_status vdb_shard_group_unique_id(void* out, size_t cap);
vdb_status vdb_shard_group_new(const void* unique_id, size_t id_bytes, int rank, int world, int device, vdb_shard_group** out);
void vdb_shard_group_free(vdb_shard_group* g);
int vdb_shard_group_size(const vdb_shard_group* g);
int vdb_shard_group_rank(const vdb_shard_group* g);
/* All-gather + merge of rows this rank has already computed (d_lists: nq x k, d_counts: nq). */
vdb_status vdb_shard_exchange_dev(vdb_shard_group* g, const vdb_scored_point* d_lists, const uint32_t* d_counts, size_t nq, size_t k,
                                  vdb_scored_point* d_out, uint32_t* d_out_n, void* stream);
/* HNSWIndex::search (hnsw.rs:513-554) over every shard: vdb_index_search_batch_dev on this rank's index, then the exchange. */
vdb_status vdb_shard_search_batch_dev(vdb_shard_group* g, const vdb_index* x, const float* d_q, size_t nq, size_t top_k, size_t ef_search,
                                      vdb_scored_point* d_out, uint32_t* d_out_n, void* stream);
/* Exact k-NN over every shard: the merged rows equal the exact k-NN of the whole set. */
vdb_status vdb_shard_exact_knn_batch_dev(vdb_shard_group* g, const vdb_index* x, const float* d_q, size_t nq, size_t top_k,
                                         vdb_scored_point* d_out, uint32_t* d_out_n, void* stream);
/* Segment::search / search_with_filter (segment.rs:103-147) over every shard: each rank's segment owns its slice of
 * the payloads, of every filter bitmap and of the tombstones; `f` may be NULL.  A shard without live points
 * contributes no rows (for one segment that is SearchError, segment.rs:104-107). */
vdb_status vdb_shard_segment_search_batch_dev(vdb_shard_group* g, const vdb_segment* s, const float* d_q, size_t nq, size_t top_k,
                                              size_t ef_search, const vdb_filter* f, vdb_scored_point* d_out, uint32_t* d_out_n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VDBMI_H */
