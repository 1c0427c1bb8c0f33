//! `vectordb::vector` on the GPU: `metric::score` and `hnsw::{ScoredPoint, HNSWIndex}` with the reference's
//! signatures (src/vector/metric.rs:4, src/vector/hnsw.rs:12-16, :59-765).
pub mod metric {
    use crate::ffi;
    use vectordb::utils::types::{DistanceMetric, Vector};

    /// metric.rs:4-12.  Panics on unequal lengths like the reference (`assert_eq!`, metric.rs:5).
    pub fn score(a: &Vector, b: &Vector, metric: DistanceMetric) -> f32 {
        let mut out = 0f32;
        let rc = unsafe { ffi::vdb_score(a.as_ptr(), a.len(), b.as_ptr(), b.len(), metric as i32, &mut out) };
        crate::vector::hnsw::check(rc).unwrap();
        out
    }
}

pub mod hnsw {
    use crate::ffi;
    use std::collections::HashMap;
    use std::ffi::CStr;
    use vectordb::utils::errors::DBError;
    use vectordb::utils::types::{DistanceMetric, PointId, Score, Vector};

    /// hnsw.rs:11-23: equality looks at `sort_key` only.
    #[derive(Clone, Debug)]
    pub struct ScoredPoint { pub id: PointId, pub raw_score: Score, pub sort_key: Score }
    impl PartialEq for ScoredPoint { fn eq(&self, o: &Self) -> bool { self.sort_key == o.sort_key } }
    impl From<ffi::vdb_scored_point> for ScoredPoint {
        fn from(p: ffi::vdb_scored_point) -> Self { ScoredPoint { id: p.id, raw_score: p.raw_score, sort_key: p.sort_key } }
    }

    pub(crate) fn last_message() -> String {
        unsafe { CStr::from_ptr(ffi::vdb_last_error_message()).to_string_lossy().into_owned() }
    }
    /// Status code -> DBError (errors.rs:9-33); VDB_ERR_PANIC re-raises the reference's panic.
    pub(crate) fn check(rc: i32) -> Result<(), DBError> {
        match rc {
            ffi::VDB_OK => Ok(()),
            ffi::VDB_ERR_NOT_FOUND => Err(DBError::NotFound(last_message().split_whitespace().nth(3).and_then(|t| t.parse().ok()).unwrap_or(0))),
            ffi::VDB_ERR_VECTOR_LENGTH_MISMATCH => {
                // "Vector length mismatch: expected {expected}, got {actual}"
                let m = last_message();
                let nums: Vec<usize> = m.split(|c: char| !c.is_ascii_digit()).filter_map(|t| t.parse().ok()).collect();
                Err(DBError::VectorLengthMismatch { expected: nums.first().copied().unwrap_or(0), actual: nums.get(1).copied().unwrap_or(0) })
            }
            ffi::VDB_ERR_IO => Err(DBError::IOError(std::io::Error::new(std::io::ErrorKind::Other, last_message()))),
            ffi::VDB_ERR_INVALID_PAYLOAD => Err(DBError::InvalidPayload(last_message().trim_start_matches("Invalid payload: ").to_string())),
            ffi::VDB_ERR_SEARCH => Err(DBError::SearchError(last_message().trim_start_matches("Search failed: ").to_string())),
            ffi::VDB_ERR_PANIC => panic!("{}", last_message()),
            _ => Err(DBError::SearchError(last_message())),      // device failure: never a silent CPU fallback
        }
    }

    /// hnsw.rs:59-73.  Owns the library handle (or borrows the one inside a segment) and mirrors what the API hands
    /// out by reference: the stored vectors (normalised for Cosine, hnsw.rs:138-139).
    pub struct HNSWIndex {
        pub(crate) h: *mut ffi::vdb_index,
        pub(crate) owned: bool,
        metric: DistanceMetric, m: usize, ef: usize, max_level_cap: usize, dim: usize,
        pub(crate) vectors: HashMap<PointId, Vector>,
        neighbors_scratch: std::cell::RefCell<HashMap<(usize, PointId), Vec<PointId>>>,
    }
    unsafe impl Send for HNSWIndex {}
    impl Drop for HNSWIndex { fn drop(&mut self) { if self.owned && !self.h.is_null() { unsafe { ffi::vdb_index_free(self.h) } } } }

    impl HNSWIndex {
        pub fn new(metric: DistanceMetric, m: usize, ef: usize, max_level_cap: usize, dim: usize) -> Self {                 // hnsw.rs:77
            let mut h = std::ptr::null_mut();
            check(unsafe { ffi::vdb_index_new(metric as i32, m, ef, max_level_cap, dim, std::ptr::null(), &mut h) }).unwrap();
            HNSWIndex { h, owned: true, metric, m, ef, max_level_cap, dim, vectors: HashMap::new(), neighbors_scratch: Default::default() }
        }
        pub fn normalize_score(&self, raw: f32) -> f32 { if matches!(self.metric, DistanceMetric::Dot) { -raw } else { raw } }   // hnsw.rs:103
        pub fn mark_deleted(&mut self, point_id: PointId) { check(unsafe { ffi::vdb_index_mark_deleted(self.h, point_id) }).unwrap() }   // hnsw.rs:111
        pub fn insert(&mut self, point_id: PointId, vector: Vector) -> Result<(), DBError> {                                 // hnsw.rs:119
            check(unsafe { ffi::vdb_index_insert(self.h, point_id, vector.as_ptr(), vector.len()) })?;
            self.remember(point_id);
            Ok(())
        }
        /// Extension: batched build (include/vdbmi.h, vdb_index_insert_batch).
        pub fn insert_batch(&mut self, ids: &[PointId], vectors: &[f32]) -> Result<(), DBError> {
            check(unsafe { ffi::vdb_index_insert_batch(self.h, ids.as_ptr(), vectors.as_ptr(), ids.len(), self.dim) })?;
            for &id in ids { self.remember(id); }
            Ok(())
        }
        pub(crate) fn remember(&mut self, id: PointId) {
            let mut v = vec![0f32; self.dim];
            if unsafe { ffi::vdb_index_get_vector(self.h, id, v.as_mut_ptr(), v.len()) } == ffi::VDB_OK { self.vectors.entry(id).or_insert(v); }
        }
        pub fn add_bidirectional_edge(&mut self, level: usize, a: PointId, b: PointId) {                                       // hnsw.rs:328
            check(unsafe { ffi::vdb_index_add_bidirectional_edge(self.h, level, a, b) }).unwrap()
        }
        pub fn greedy_search_layer_unfiltered(&self, query: &Vector, entry: PointId, level: usize) -> PointId {                // hnsw.rs:333
            let mut out = 0u64;
            check(unsafe { ffi::vdb_index_greedy_search_layer_unfiltered(self.h, query.as_ptr(), query.len(), entry, level, &mut out) }).unwrap();
            out
        }
        pub fn search(&self, query: &Vector, top_k: usize) -> Result<Vec<ScoredPoint>, DBError> {                              // hnsw.rs:513
            let mut out = vec![ffi::vdb_scored_point::default(); top_k.max(1)];
            let mut n = 0usize;
            check(unsafe { ffi::vdb_index_search(self.h, query.as_ptr(), query.len(), top_k, out.as_mut_ptr(), &mut n) })?;
            Ok(out[..n].iter().map(|p| (*p).into()).collect())
        }
        /// Extension: the GPU's native shape — `queries` is nq x dim row-major; row q of the result holds `counts[q]` points.
        pub fn search_batch(&self, queries: &[f32], top_k: usize, ef_search: usize) -> Result<Vec<Vec<ScoredPoint>>, DBError> {
            let nq = queries.len() / self.dim;
            let mut out = vec![ffi::vdb_scored_point::default(); nq * top_k.max(1)];
            let mut cnt = vec![0u32; nq];
            check(unsafe { ffi::vdb_index_search_batch(self.h, queries.as_ptr(), nq, self.dim, top_k, ef_search, out.as_mut_ptr(), cnt.as_mut_ptr(), std::ptr::null_mut()) })?;
            Ok((0..nq).map(|q| out[q * top_k..q * top_k + cnt[q] as usize].iter().map(|p| (*p).into()).collect()).collect())
        }
        pub fn contains(&self, point_id: &PointId) -> bool { unsafe { ffi::vdb_index_contains(self.h, *point_id) != 0 } }      // hnsw.rs:691
        pub fn len(&self) -> usize { unsafe { ffi::vdb_index_len(self.h) } }                                                    // hnsw.rs:695
        pub fn layer_neighbors(&self, level: usize, point_id: PointId) -> Option<Vec<PointId>> {                               // hnsw.rs:699 (owned copy)
            let (mut n, mut some) = (0usize, 0i32);
            check(unsafe { ffi::vdb_index_layer_neighbors(self.h, level, point_id, std::ptr::null_mut(), 0, &mut n, &mut some) }).unwrap();
            if some == 0 { return None; }
            let mut ids = vec![0u64; n.max(1)];
            check(unsafe { ffi::vdb_index_layer_neighbors(self.h, level, point_id, ids.as_mut_ptr(), ids.len(), &mut n, &mut some) }).unwrap();
            ids.truncate(n);
            self.neighbors_scratch.borrow_mut().insert((level, point_id), ids.clone());
            Some(ids)
        }
        pub fn iter_vectors(&self) -> impl Iterator<Item = (&PointId, &Vector)> { self.vectors.iter() }                        // hnsw.rs:703
        pub fn metric(&self) -> DistanceMetric { self.metric }
        pub fn m(&self) -> usize { self.m }
        pub fn ef(&self) -> usize { self.ef }
        pub fn max_level_cap(&self) -> usize { self.max_level_cap }
        pub fn dim(&self) -> usize { self.dim }
        pub fn get_vector(&self, point_id: &PointId) -> Option<&Vector> {                                                      // hnsw.rs:727 (hides tombstones)
            let mut probe = vec![0f32; self.dim];
            if unsafe { ffi::vdb_index_get_vector(self.h, *point_id, probe.as_mut_ptr(), probe.len()) } != ffi::VDB_OK { return None; }
            self.vectors.get(point_id)
        }
        pub fn get_entry_point(&self) -> Option<u64> {                                                                         // hnsw.rs:736
            let mut id = 0u64;
            if unsafe { ffi::vdb_index_get_entry_point(self.h, &mut id) } != 0 { Some(id) } else { None }
        }
        pub fn current_max_level(&self) -> usize { unsafe { ffi::vdb_index_current_max_level(self.h) } }
        pub fn set_entry_point(&mut self, point_id: PointId) { check(unsafe { ffi::vdb_index_set_entry_point(self.h, point_id) }).unwrap() }
        pub fn set_current_max_level(&mut self, level: usize) { check(unsafe { ffi::vdb_index_set_current_max_level(self.h, level) }).unwrap() }
        pub fn maybe_normalize(&self, vec: &Vector) -> Vector {                                                               // hnsw.rs:752
            let mut out = vec![0f32; vec.len()];
            check(unsafe { ffi::vdb_index_maybe_normalize(self.h, vec.as_ptr(), vec.len(), out.as_mut_ptr()) }).unwrap();
            out
        }
    }
}
