//! Scale-out over several GPUs: one shard (an `HNSWIndex` or a `Segment` with its own graph) per rank, one RCCL
//! communicator per group.  The reference holds a single in-memory segment (src/segment/segment.rs:11-18); sharding is
//! the scaling axis the library adds.  A sharded search is the rank's own search, one all-gather of the per-shard top-k
//! on the caller's HIP stream and a k-way merge by (sort_key, id) on every rank; queries, results and counts are device
//! buffers (`nq * top_k` records, `nq` counts).
use std::ffi::c_void;
use std::ptr;

use crate::ffi;
use crate::segment::segment::Segment;
use crate::vector::hnsw::{check, HNSWIndex};
use vectordb::utils::errors::DBError;

/// Bytes of the communicator id that rank 0 creates and hands to every other rank (by any channel the caller has).
pub const UNIQUE_ID_BYTES: usize = 128;

pub struct ShardGroup {
    g: *mut ffi::vdb_shard_group,
}

unsafe impl Send for ShardGroup {}

impl ShardGroup {
    /// Rank 0 calls this once; the bytes go to every rank of the group.
    pub fn unique_id() -> Result<[u8; UNIQUE_ID_BYTES], DBError> {
        let mut id = [0u8; UNIQUE_ID_BYTES];
        check(unsafe { ffi::vdb_shard_group_unique_id(id.as_mut_ptr() as *mut c_void, id.len()) })?;
        Ok(id)
    }

    /// Collective: every rank of the group calls it with the same id (`ncclCommInitRank`).
    pub fn new(unique_id: &[u8; UNIQUE_ID_BYTES], rank: i32, world: i32, device: i32) -> Result<Self, DBError> {
        let mut g = ptr::null_mut();
        check(unsafe { ffi::vdb_shard_group_new(unique_id.as_ptr() as *const c_void, unique_id.len(), rank, world, device, &mut g) })?;
        Ok(ShardGroup { g })
    }

    /// Ranks of the communicator as RCCL reports them (`ncclCommCount`).
    pub fn size(&self) -> i32 { unsafe { ffi::vdb_shard_group_size(self.g) } }
    pub fn rank(&self) -> i32 { unsafe { ffi::vdb_shard_group_rank(self.g) } }

    /// `HNSWIndex::search` for `nq` queries over every rank's shard.  `stream` null: the index's own stream.
    ///
    /// # Safety
    /// `d_queries`, `d_out`, `d_out_n` are device buffers of `nq * dim` floats, `nq * top_k` records and `nq` counts
    /// that stay valid until `stream` has run the call.
    pub unsafe fn search_batch_dev(&self, shard: &HNSWIndex, d_queries: *const f32, nq: usize, top_k: usize, ef_search: usize,
                                   d_out: *mut ffi::vdb_scored_point, d_out_n: *mut u32, stream: *mut c_void) -> Result<(), DBError> {
        check(ffi::vdb_shard_search_batch_dev(self.g, shard.h, d_queries, nq, top_k, ef_search, d_out, d_out_n, stream))
    }

    /// Exact k-NN (brute force, (key, id) order) over every rank's shard.
    ///
    /// # Safety
    /// As `search_batch_dev`.
    pub unsafe fn exact_knn_batch_dev(&self, shard: &HNSWIndex, d_queries: *const f32, nq: usize, top_k: usize,
                                      d_out: *mut ffi::vdb_scored_point, d_out_n: *mut u32, stream: *mut c_void) -> Result<(), DBError> {
        check(ffi::vdb_shard_exact_knn_batch_dev(self.g, shard.h, d_queries, nq, top_k, d_out, d_out_n, stream))
    }

    /// `Segment::search` / `Segment::search_with_filter` over every rank's segment: each rank owns its slice of the
    /// payloads, of every filter bitmap and of the tombstones; a shard without live points contributes nothing.
    ///
    /// # Safety
    /// As `search_batch_dev`; `filter` null: unfiltered.
    pub unsafe fn segment_search_batch_dev(&self, shard: &Segment, d_queries: *const f32, nq: usize, top_k: usize, ef_search: usize,
                                           filter: *const ffi::vdb_filter, d_out: *mut ffi::vdb_scored_point, d_out_n: *mut u32,
                                           stream: *mut c_void) -> Result<(), DBError> {
        check(ffi::vdb_shard_segment_search_batch_dev(self.g, shard.raw(), d_queries, nq, top_k, ef_search, filter, d_out, d_out_n, stream))
    }

    /// All-gather + merge of per-shard lists computed elsewhere.
    ///
    /// # Safety
    /// As `search_batch_dev`; `d_lists` / `d_counts` hold this rank's `nq * k` records and `nq` counts.
    pub unsafe fn exchange_dev(&self, d_lists: *const ffi::vdb_scored_point, d_counts: *const u32, nq: usize, k: usize,
                               d_out: *mut ffi::vdb_scored_point, d_out_n: *mut u32, stream: *mut c_void) -> Result<(), DBError> {
        check(ffi::vdb_shard_exchange_dev(self.g, d_lists, d_counts, nq, k, d_out, d_out_n, stream))
    }
}

impl Drop for ShardGroup {
    fn drop(&mut self) { unsafe { ffi::vdb_shard_group_free(self.g) } }
}
