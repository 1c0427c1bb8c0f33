//! Raw bindings of include/vdbmi.h: every entry point of the header (the second `extern` block was produced from the
//! header by tools/gen_rust_ffi.py; enums travel as `c_int`).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int, c_void};

#[repr(C)] pub struct vdb_index { _p: [u8; 0] }
#[repr(C)] pub struct vdb_segment { _p: [u8; 0] }
#[repr(C)] pub struct vdb_shard_group { _p: [u8; 0] }
#[repr(C)] pub struct vdb_value { _p: [u8; 0] }
#[repr(C)] pub struct vdb_payload { _p: [u8; 0] }
#[repr(C)] pub struct vdb_filter { _p: [u8; 0] }
#[repr(C)] pub struct vdb_payload_index { _p: [u8; 0] }

/// ScoredPoint, src/vector/hnsw.rs:11-16
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct vdb_scored_point { pub id: u64, pub raw_score: f32, pub sort_key: f32 }

/// per-query work counters (vdbmi.h: vdb_query_stats)
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct vdb_query_stats { pub rows_scored: u32, pub expanded: u32, pub adj_entries: u32, pub flags: u32 }

/// one layer of a graph handed over in CSR form (vdbmi.h: vdb_layer_csr)
#[repr(C)] #[derive(Clone, Copy)]
pub struct vdb_layer_csr { pub n_lists: usize, pub owners: *const u32, pub offsets: *const u64, pub neighbors: *const u32, pub n_edges: usize }

#[repr(C)] #[derive(Clone, Copy)]
pub struct vdb_options { pub seed: u64, pub device: i32, pub arith: i32, pub filter_mode: i32, pub beam_width: i32 }

pub const VDB_OK: c_int = 0;
pub const VDB_ERR_NOT_FOUND: c_int = 1;
pub const VDB_ERR_VECTOR_LENGTH_MISMATCH: c_int = 2;
pub const VDB_ERR_IO: c_int = 3;
pub const VDB_ERR_INVALID_PAYLOAD: c_int = 6;
pub const VDB_ERR_SEARCH: c_int = 7;
pub const VDB_ERR_PANIC: c_int = 99;

extern "C" {
    pub fn vdb_last_error_message() -> *const c_char;
    pub fn vdb_default_options(out: *mut vdb_options);
    pub fn vdb_score(a: *const f32, na: usize, b: *const f32, nb: usize, metric: c_int, out: *mut f32) -> c_int;

    pub fn vdb_value_int(v: i64) -> *mut vdb_value;
    pub fn vdb_value_float(v: c_double) -> *mut vdb_value;
    pub fn vdb_value_str(s: *const c_char, len: usize) -> *mut vdb_value;
    pub fn vdb_value_bool(v: c_int) -> *mut vdb_value;
    pub fn vdb_value_list_int(v: *const i64, n: usize) -> *mut vdb_value;
    pub fn vdb_value_list_float(v: *const c_double, n: usize) -> *mut vdb_value;
    pub fn vdb_value_list_str(s: *const *const c_char, lens: *const usize, n: usize) -> *mut vdb_value;
    pub fn vdb_value_list_bool(v: *const u8, n: usize) -> *mut vdb_value;
    pub fn vdb_value_free(v: *mut vdb_value);
    pub fn vdb_payload_new() -> *mut vdb_payload;
    pub fn vdb_payload_free(p: *mut vdb_payload);
    pub fn vdb_payload_set(p: *mut vdb_payload, key: *const c_char, v: *const vdb_value) -> c_int;
    pub fn vdb_filter_match(key: *const c_char, v: *const vdb_value) -> *mut vdb_filter;
    pub fn vdb_filter_compare(key: *const c_char, op: c_int, v: *const vdb_value) -> *mut vdb_filter;
    pub fn vdb_filter_and(children: *const *const vdb_filter, n: usize) -> *mut vdb_filter;
    pub fn vdb_filter_or(children: *const *const vdb_filter, n: usize) -> *mut vdb_filter;
    pub fn vdb_filter_not(child: *const vdb_filter) -> *mut vdb_filter;
    pub fn vdb_filter_free(f: *mut vdb_filter);

    pub fn vdb_index_new(metric: c_int, m: usize, ef: usize, max_level_cap: usize, dim: usize, opts: *const vdb_options, out: *mut *mut vdb_index) -> c_int;
    pub fn vdb_index_free(x: *mut vdb_index);
    pub fn vdb_index_insert(x: *mut vdb_index, id: u64, v: *const f32, len: usize) -> c_int;
    pub fn vdb_index_insert_batch(x: *mut vdb_index, ids: *const u64, v: *const f32, n: usize, len: usize) -> c_int;
    pub fn vdb_index_search(x: *const vdb_index, q: *const f32, len: usize, top_k: usize, out: *mut vdb_scored_point, out_n: *mut usize) -> c_int;
    pub fn vdb_index_search_batch(x: *const vdb_index, q: *const f32, nq: usize, len: usize, top_k: usize, ef_search: usize,
                                  out: *mut vdb_scored_point, out_n: *mut u32, stats: *mut c_void) -> c_int;
    pub fn vdb_index_mark_deleted(x: *mut vdb_index, id: u64) -> c_int;
    pub fn vdb_index_contains(x: *const vdb_index, id: u64) -> c_int;
    pub fn vdb_index_len(x: *const vdb_index) -> usize;
    pub fn vdb_index_current_max_level(x: *const vdb_index) -> usize;
    pub fn vdb_index_get_entry_point(x: *const vdb_index, out: *mut u64) -> c_int;
    pub fn vdb_index_set_entry_point(x: *mut vdb_index, id: u64) -> c_int;
    pub fn vdb_index_set_current_max_level(x: *mut vdb_index, level: usize) -> c_int;
    pub fn vdb_index_get_vector(x: *const vdb_index, id: u64, out: *mut f32, cap: usize) -> c_int;
    pub fn vdb_index_layer_neighbors(x: *const vdb_index, level: usize, id: u64, out: *mut u64, cap: usize, out_n: *mut usize, is_some: *mut c_int) -> c_int;
    pub fn vdb_index_maybe_normalize(x: *const vdb_index, v: *const f32, len: usize, out: *mut f32) -> c_int;
    pub fn vdb_index_add_bidirectional_edge(x: *mut vdb_index, level: usize, a: u64, b: u64) -> c_int;
    pub fn vdb_index_greedy_search_layer_unfiltered(x: *const vdb_index, q: *const f32, len: usize, entry: u64, level: usize, out_id: *mut u64) -> c_int;

    pub fn vdb_segment_new(hnsw: *mut vdb_index, out: *mut *mut vdb_segment) -> c_int;
    pub fn vdb_segment_free(s: *mut vdb_segment);
    pub fn vdb_segment_insert(s: *mut vdb_segment, v: *const f32, len: usize, payload: *const vdb_payload, out_id: *mut u64) -> c_int;
    pub fn vdb_segment_delete(s: *mut vdb_segment, id: u64) -> c_int;
    pub fn vdb_segment_purge(s: *mut vdb_segment) -> c_int;
    pub fn vdb_segment_is_deleted(s: *const vdb_segment, id: u64) -> c_int;
    pub fn vdb_segment_hnsw_mut(s: *mut vdb_segment) -> *mut vdb_index;
    pub fn vdb_segment_search(s: *const vdb_segment, q: *const f32, len: usize, top_k: usize, out: *mut vdb_scored_point, out_n: *mut usize) -> c_int;
    pub fn vdb_segment_search_with_filter(s: *const vdb_segment, q: *const f32, len: usize, top_k: usize, f: *const vdb_filter,
                                          out: *mut vdb_scored_point, out_n: *mut usize) -> c_int;
    pub fn vdb_segment_search_unfiltered(s: *const vdb_segment, q: *const f32, len: usize, top_k: usize, out: *mut vdb_scored_point, out_n: *mut usize) -> c_int;
    pub fn vdb_segment_post_filter(s: *const vdb_segment, q: *const f32, len: usize, top_k: usize, f: *const vdb_filter,
                                   out: *mut vdb_scored_point, out_n: *mut usize) -> c_int;
    pub fn vdb_segment_in_place_filtered_search(s: *const vdb_segment, q: *const f32, len: usize, top_k: usize, f: *const vdb_filter,
                                                out: *mut vdb_scored_point, out_n: *mut usize) -> c_int;
    pub fn vdb_segment_search_batch(s: *const vdb_segment, q: *const f32, nq: usize, len: usize, top_k: usize, ef_search: usize, f: *const vdb_filter,
                                    out: *mut vdb_scored_point, out_n: *mut u32) -> c_int;
    pub fn vdb_segment_save(s: *const vdb_segment, path: *const c_char) -> c_int;
    pub fn vdb_segment_load(s: *mut vdb_segment, path: *const c_char) -> c_int;
}

// the rest of the header, declared mechanically (tools/gen_rust_ffi.py)
extern "C" {
    pub fn vdb_device_count() -> c_int;
    pub fn vdb_score_scan(queries: *const f32, nq: usize, base: *const f32, n: usize, dim: usize, metric: c_int, arith: c_int, device: c_int, out: *mut f32) -> c_int;
    pub fn vdb_score_scan_dev(d_queries: *const f32, nq: usize, d_base: *const f32, n: usize, dim: usize, stride: usize, metric: c_int, arith: c_int, d_out: *mut f32, stream: *mut c_void) -> c_int;
    pub fn vdb_value_type(v: *const vdb_value) -> c_int;
    pub fn vdb_value_equals(a: *const vdb_value, b: *const vdb_value) -> c_int;
    pub fn vdb_value_compare_scalar(a: *const vdb_value, op: c_int, b: *const vdb_value, is_some: *mut c_int, out: *mut c_int) -> c_int;
    pub fn vdb_value_evaluate_list_query(self_: *const vdb_value, kind: c_int, op: c_int, n: usize, v: *const vdb_value, is_some: *mut c_int, out: *mut c_int) -> c_int;
    pub fn vdb_payload_get(p: *const vdb_payload, key: *const c_char) -> *const vdb_value;
    pub fn vdb_payload_compare_field(p: *const vdb_payload, field: *const c_char, op: c_int, other: *const vdb_value, out: *mut c_int) -> c_int;
    pub fn vdb_payload_evaluate_list_field(p: *const vdb_payload, field: *const c_char, kind: c_int, op: c_int, n: usize, v: *const vdb_value, out: *mut c_int) -> c_int;
    pub fn vdb_filter_evaluate(f: *const vdb_filter, p: *const vdb_payload, out: *mut c_int) -> c_int;
    pub fn vdb_payload_index_new() -> *mut vdb_payload_index;
    pub fn vdb_payload_index_free(x: *mut vdb_payload_index);
    pub fn vdb_payload_index_insert(x: *mut vdb_payload_index, id: u64, p: *const vdb_payload) -> c_int;
    pub fn vdb_payload_index_remove(x: *mut vdb_payload_index, id: u64, p: *const vdb_payload) -> c_int;
    pub fn vdb_payload_index_query_exact(x: *const vdb_payload_index, key: *const c_char, v: *const vdb_value, out_ids: *mut u64, cap: usize, out_n: *mut usize, is_some: *mut c_int) -> c_int;
    pub fn vdb_payload_index_all_for_key(x: *const vdb_payload_index, key: *const c_char, out_ids: *mut u64, cap: usize, out_n: *mut usize, is_some: *mut c_int) -> c_int;
    pub fn vdb_index_search_batch_dev(x: *const vdb_index, d_queries: *const f32, nq: usize, top_k: usize, ef_search: usize, d_out: *mut vdb_scored_point, d_out_n: *mut u32, d_stats: *mut vdb_query_stats, stream: *mut c_void) -> c_int;
    pub fn vdb_index_exact_knn_batch(x: *const vdb_index, queries: *const f32, nq: usize, len: usize, top_k: usize, out: *mut vdb_scored_point, out_n: *mut u32) -> c_int;
    pub fn vdb_index_exact_knn_batch_dev(x: *const vdb_index, d_queries: *const f32, nq: usize, top_k: usize, d_out: *mut vdb_scored_point, d_out_n: *mut u32, stream: *mut c_void) -> c_int;
    pub fn vdb_index_exact_knn_batch_dev_noalloc(x: *const vdb_index, d_queries: *const f32, nq: usize, top_k: usize, d_out: *mut vdb_scored_point, d_out_n: *mut u32, stream: *mut c_void) -> c_int;
    pub fn vdb_index_metric(x: *const vdb_index) -> c_int;
    pub fn vdb_index_m(x: *const vdb_index) -> usize;
    pub fn vdb_index_ef(x: *const vdb_index) -> usize;
    pub fn vdb_index_max_level_cap(x: *const vdb_index) -> usize;
    pub fn vdb_index_dim(x: *const vdb_index) -> usize;
    pub fn vdb_index_normalize_score(x: *const vdb_index, raw: f32) -> f32;
    pub fn vdb_index_import_graph(x: *mut vdb_index, n: usize, ids: *const u64, levels: *const u32, vectors: *const f32, deleted: *const u8, n_layers: usize, layers: *const vdb_layer_csr, has_entry: c_int, entry_id: u64, current_max_level: usize) -> c_int;
    pub fn vdb_index_export_sizes(x: *const vdb_index, n: *mut usize, n_layers: *mut usize, n_lists: *mut usize, n_edges: *mut usize, cap_layers: usize) -> c_int;
    pub fn vdb_index_export_points(x: *const vdb_index, ids: *mut u64, levels: *mut u32, vectors: *mut f32, deleted: *mut u8) -> c_int;
    pub fn vdb_index_export_layer(x: *const vdb_index, level: usize, owners: *mut u32, offsets: *mut u64, neighbors: *mut u32) -> c_int;
    pub fn vdb_index_save(x: *const vdb_index, path: *const c_char) -> c_int;
    pub fn vdb_index_load(x: *mut vdb_index, path: *const c_char) -> c_int;
    pub fn vdb_index_device_info(x: *const vdb_index, n_slots: *mut usize, row_stride_floats: *mut usize, l0_edges: *mut usize, upper_edges: *mut usize, bytes_total: *mut usize) -> c_int;
    pub fn vdb_index_sync_device(x: *mut vdb_index) -> c_int;
    pub fn vdb_segment_insert_batch(s: *mut vdb_segment, v: *const f32, n: usize, len: usize, payloads: *const *const vdb_payload, out_ids: *mut u64) -> c_int;
    pub fn vdb_segment_insert_batch_columns(s: *mut vdb_segment, v: *const f32, n: usize, len: usize, n_keys: usize, keys: *const *const c_char, kinds: *const i32, cols: *const *const c_void, out_ids: *mut u64) -> c_int;
    pub fn vdb_segment_get_vector(s: *const vdb_segment, id: u64, out: *mut f32, cap: usize) -> c_int;
    pub fn vdb_segment_get_payload(s: *const vdb_segment, id: u64) -> *const vdb_payload;
    pub fn vdb_segment_hnsw(s: *const vdb_segment) -> *const vdb_index;
    pub fn vdb_segment_payload_index(s: *const vdb_segment) -> *const vdb_payload_index;
    pub fn vdb_segment_payload_count(s: *const vdb_segment) -> usize;
    pub fn vdb_segment_search_batch_dev(s: *const vdb_segment, d_queries: *const f32, nq: usize, top_k: usize, ef_search: usize, f: *const vdb_filter, d_out: *mut vdb_scored_point, d_out_n: *mut u32, d_stats: *mut vdb_query_stats, stream: *mut c_void) -> c_int;
    pub fn vdb_segment_compile_filter(s: *const vdb_segment, f: *const vdb_filter, n_pass: *mut usize, n_err: *mut usize) -> c_int;
    pub fn vdb_segment_filter_bitmaps(s: *const vdb_segment, f: *const vdb_filter, pass: *mut u32, err: *mut u32, cap_words: usize) -> c_int;
    pub fn vdb_topk_merge_dev(d_lists: *const vdb_scored_point, d_counts: *const u32, n_shards: usize, nq: usize, k: usize, d_out: *mut vdb_scored_point, d_out_n: *mut u32, stream: *mut c_void) -> c_int;
    // scale-out: one shard per rank, one RCCL communicator per group (include/vdbmi.h, shard group section)
    pub fn vdb_shard_group_unique_id(out: *mut c_void, cap: usize) -> c_int;
    pub fn vdb_shard_group_new(unique_id: *const c_void, id_bytes: usize, rank: c_int, world: c_int, device: c_int, out: *mut *mut vdb_shard_group) -> c_int;
    pub fn vdb_shard_group_free(g: *mut vdb_shard_group);
    pub fn vdb_shard_group_size(g: *const vdb_shard_group) -> c_int;
    pub fn vdb_shard_group_rank(g: *const vdb_shard_group) -> c_int;
    pub fn vdb_shard_exchange_dev(g: *mut vdb_shard_group, d_lists: *const vdb_scored_point, d_counts: *const u32, nq: usize, k: usize, d_out: *mut vdb_scored_point, d_out_n: *mut u32, stream: *mut c_void) -> c_int;
    pub fn vdb_shard_search_batch_dev(g: *mut vdb_shard_group, x: *const vdb_index, d_q: *const f32, nq: usize, top_k: usize, ef_search: usize, d_out: *mut vdb_scored_point, d_out_n: *mut u32, stream: *mut c_void) -> c_int;
    pub fn vdb_shard_exact_knn_batch_dev(g: *mut vdb_shard_group, x: *const vdb_index, d_q: *const f32, nq: usize, top_k: usize, d_out: *mut vdb_scored_point, d_out_n: *mut u32, stream: *mut c_void) -> c_int;
    pub fn vdb_shard_segment_search_batch_dev(g: *mut vdb_shard_group, s: *const vdb_segment, d_q: *const f32, nq: usize, top_k: usize, ef_search: usize, f: *const vdb_filter, d_out: *mut vdb_scored_point, d_out_n: *mut u32, stream: *mut c_void) -> c_int;
}
