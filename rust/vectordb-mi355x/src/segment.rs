//! `vectordb::segment::segment::Segment` (src/segment/segment.rs:11-264) over `vdb_segment`.
pub mod segment {
    use crate::ffi;
    use crate::vector::hnsw::{check, HNSWIndex, ScoredPoint};
    use std::collections::{HashMap, HashSet};
    use std::ffi::CString;
    use vectordb::payload_storage::filters::Filter;
    use vectordb::payload_storage::stores::PayloadIndex;
    use vectordb::utils::errors::DBError;
    use vectordb::utils::payload::{Payload, PayloadValue};
    use vectordb::utils::types::{PointId, Vector};

    /// Temporary C-side copies of a PayloadValue / Payload / Filter; the library copies what it keeps.
    struct CValue(*mut ffi::vdb_value);
    impl Drop for CValue { fn drop(&mut self) { unsafe { ffi::vdb_value_free(self.0) } } }
    fn c_value(v: &PayloadValue) -> CValue {
        unsafe {
            CValue(match v {
                PayloadValue::Int(i) => ffi::vdb_value_int(*i),
                PayloadValue::Float(f) => ffi::vdb_value_float(f.into_inner()),
                PayloadValue::Str(s) => ffi::vdb_value_str(s.as_ptr() as *const _, s.len()),
                PayloadValue::Bool(b) => ffi::vdb_value_bool(*b as i32),
                PayloadValue::ListInt(l) => ffi::vdb_value_list_int(l.as_ptr(), l.len()),
                PayloadValue::ListFloat(l) => { let d: Vec<f64> = l.iter().map(|x| x.into_inner()).collect(); ffi::vdb_value_list_float(d.as_ptr(), d.len()) }
                PayloadValue::ListStr(l) => {
                    let ptrs: Vec<*const std::os::raw::c_char> = l.iter().map(|s| s.as_ptr() as *const _).collect();
                    let lens: Vec<usize> = l.iter().map(|s| s.len()).collect();
                    ffi::vdb_value_list_str(ptrs.as_ptr(), lens.as_ptr(), l.len())
                }
                PayloadValue::ListBool(l) => { let d: Vec<u8> = l.iter().map(|b| *b as u8).collect(); ffi::vdb_value_list_bool(d.as_ptr(), d.len()) }
            })
        }
    }
    struct CPayload(*mut ffi::vdb_payload);
    impl Drop for CPayload { fn drop(&mut self) { unsafe { ffi::vdb_payload_free(self.0) } } }
    fn c_payload(p: &Payload) -> CPayload {
        let c = CPayload(unsafe { ffi::vdb_payload_new() });
        for (k, v) in p.0.iter() {
            let key = CString::new(k.as_str()).expect("payload keys hold no NUL");
            let val = c_value(v);
            check(unsafe { ffi::vdb_payload_set(c.0, key.as_ptr(), val.0) }).unwrap();
        }
        c
    }
    struct CFilter(*mut ffi::vdb_filter);
    impl Drop for CFilter { fn drop(&mut self) { unsafe { ffi::vdb_filter_free(self.0) } } }
    fn c_filter(f: &Filter) -> CFilter {
        unsafe {
            CFilter(match f {
                Filter::Match { key, value } => { let k = CString::new(key.as_str()).unwrap(); let v = c_value(value); ffi::vdb_filter_match(k.as_ptr(), v.0) }
                Filter::Compare { key, op, value } => { let k = CString::new(key.as_str()).unwrap(); let v = c_value(value); ffi::vdb_filter_compare(k.as_ptr(), *op as i32, v.0) }
                Filter::And(kids) | Filter::Or(kids) => {
                    let cs: Vec<CFilter> = kids.iter().map(c_filter).collect();                      // children are copied by the library
                    let ptrs: Vec<*const ffi::vdb_filter> = cs.iter().map(|c| c.0 as *const _).collect();
                    if matches!(f, Filter::And(_)) { ffi::vdb_filter_and(ptrs.as_ptr(), ptrs.len()) } else { ffi::vdb_filter_or(ptrs.as_ptr(), ptrs.len()) }
                }
                Filter::Not(inner) => { let c = c_filter(inner); ffi::vdb_filter_not(c.0) }
            })
        }
    }

    /// segment.rs:11-18.  The library owns index, payload columns, inverted index and tombstones; this side mirrors
    /// what the API returns by reference.
    pub struct Segment {
        s: *mut ffi::vdb_segment,
        hnsw: HNSWIndex,                                  // borrows the index inside `s`
        payloads: HashMap<PointId, Payload>,
        payload_index: PayloadIndex,
        deleted: HashSet<PointId>,
    }
    unsafe impl Send for Segment {}
    impl Drop for Segment { fn drop(&mut self) { unsafe { ffi::vdb_segment_free(self.s) } } }
    impl Segment {
        /// The library's handle, for the sharded entry points (crate::shard).
        pub(crate) fn raw(&self) -> *const ffi::vdb_segment { self.s }
    }

    type Call = unsafe extern "C" fn(*const ffi::vdb_segment, *const f32, usize, usize, *const ffi::vdb_filter, *mut ffi::vdb_scored_point, *mut usize) -> i32;

    impl Segment {
        pub fn new(mut hnsw: HNSWIndex) -> Self {                                                                          // segment.rs:21
            let mut s = std::ptr::null_mut();
            check(unsafe { ffi::vdb_segment_new(hnsw.h, &mut s) }).unwrap();       // takes ownership of the index handle
            hnsw.owned = false;
            Segment { s, hnsw, payloads: HashMap::new(), payload_index: PayloadIndex::new(), deleted: HashSet::new() }
        }
        pub fn insert(&mut self, vector: Vector, payload: Option<Payload>) -> Result<PointId, DBError> {                   // segment.rs:32
            let c = payload.as_ref().map(c_payload);
            let mut id = 0u64;
            check(unsafe { ffi::vdb_segment_insert(self.s, vector.as_ptr(), vector.len(), c.as_ref().map_or(std::ptr::null(), |p| p.0 as *const _), &mut id) })?;
            self.hnsw.remember(id);
            if let Some(p) = payload { self.payload_index.insert(id, &p); self.payloads.insert(id, p); }
            Ok(id)
        }
        pub fn get_vector(&self, point_id: PointId) -> Option<&Vector> {                                                   // segment.rs:64
            if self.deleted.contains(&point_id) { return None; }
            self.hnsw.get_vector(&point_id)
        }
        pub fn delete(&mut self, point_id: PointId) -> Result<(), DBError> {                                               // segment.rs:71
            if self.deleted.contains(&point_id) || !self.hnsw.contains(&point_id) { return Ok(()); }
            let old_handle = self.hnsw.h;
            check(unsafe { ffi::vdb_segment_delete(self.s, point_id) })?;
            // a purge (segment.rs:88-95) replaces and frees the index inside the segment: take the current handle before
            // anything else goes through it
            self.hnsw.h = unsafe { ffi::vdb_segment_hnsw_mut(self.s) };
            if let Some(p) = self.payloads.get(&point_id) { self.payload_index.remove(point_id, p); }
            self.deleted.insert(point_id);
            let purged = self.hnsw.h != old_handle || unsafe { ffi::vdb_index_contains(self.hnsw.h, point_id) } == 0;
            if purged { self.after_purge(); }
            Ok(())
        }
        fn after_purge(&mut self) {                                                // segment.rs:166-217: deleted points are gone for good
            let gone: Vec<PointId> = self.deleted.drain().collect();
            for id in gone { self.payloads.remove(&id); self.hnsw.vectors.remove(&id); }
            // purge re-inserts the live rows (Cosine rows are normalised again, hnsw.rs:138): refresh the mirrored vectors
            let dim = unsafe { ffi::vdb_index_dim(self.hnsw.h) };
            for (id, v) in self.hnsw.vectors.iter_mut() {
                let mut row = vec![0f32; dim];
                if unsafe { ffi::vdb_index_get_vector(self.hnsw.h, *id, row.as_mut_ptr(), dim) } == 0 { *v = row; }
            }
        }
        fn run(&self, call: Call, query: &Vector, top_k: usize, filter: Option<&Filter>) -> Result<Vec<ScoredPoint>, DBError> {
            let c = filter.map(c_filter);
            let mut out = vec![ffi::vdb_scored_point::default(); top_k.max(1) * 4];
            let mut n = 0usize;
            check(unsafe { call(self.s, query.as_ptr(), query.len(), top_k, c.as_ref().map_or(std::ptr::null(), |f| f.0 as *const _), out.as_mut_ptr(), &mut n) })?;
            Ok(out[..n].iter().map(|p| (*p).into()).collect())
        }
        pub fn search(&self, query: &Vector, top_k: usize) -> Result<Vec<ScoredPoint>, DBError> {                          // segment.rs:103
            let mut out = vec![ffi::vdb_scored_point::default(); top_k.max(1)];
            let mut n = 0usize;
            check(unsafe { ffi::vdb_segment_search(self.s, query.as_ptr(), query.len(), top_k, out.as_mut_ptr(), &mut n) })?;
            Ok(out[..n].iter().map(|p| (*p).into()).collect())
        }
        pub fn search_with_filter(&self, query: &Vector, top_k: usize, filter: Option<&Filter>) -> Result<Vec<ScoredPoint>, DBError> {   // segment.rs:121
            self.run(ffi::vdb_segment_search_with_filter, query, top_k, filter)
        }
        pub fn search_unfiltered(&self, query: &Vector, top_k: usize) -> Result<Vec<ScoredPoint>, DBError> {               // segment.rs:152
            let mut out = vec![ffi::vdb_scored_point::default(); top_k.max(1)];
            let mut n = 0usize;
            check(unsafe { ffi::vdb_segment_search_unfiltered(self.s, query.as_ptr(), query.len(), top_k, out.as_mut_ptr(), &mut n) })?;
            Ok(out[..n].iter().map(|p| (*p).into()).collect())
        }
        pub fn post_filter(&self, query: &Vector, top_k: usize, filter: Option<&Filter>) -> Result<Vec<ScoredPoint>, DBError> {          // segment.rs:221
            self.run(ffi::vdb_segment_post_filter, query, top_k, filter)
        }
        /// Extension: batched Segment::search / search_with_filter (include/vdbmi.h, vdb_segment_search_batch).
        pub fn search_batch(&self, queries: &[f32], top_k: usize, ef_search: usize, filter: Option<&Filter>) -> Result<Vec<Vec<ScoredPoint>>, DBError> {
            let dim = self.hnsw.dim();
            let nq = queries.len() / dim;
            let c = filter.map(c_filter);
            let mut out = vec![ffi::vdb_scored_point::default(); nq * top_k.max(1)];
            let mut cnt = vec![0u32; nq];
            check(unsafe { ffi::vdb_segment_search_batch(self.s, queries.as_ptr(), nq, dim, top_k, ef_search, c.as_ref().map_or(std::ptr::null(), |f| f.0 as *const _),
                                                         out.as_mut_ptr(), cnt.as_mut_ptr()) })?;
            Ok((0..nq).map(|q| out[q * top_k..q * top_k + cnt[q] as usize].iter().map(|p| (*p).into()).collect()).collect())
        }
        pub fn get_payload(&self, point_id: PointId) -> Option<&Payload> { self.payloads.get(&point_id) }                  // segment.rs:157
        pub fn is_deleted(&self, point_id: PointId) -> bool { self.deleted.contains(&point_id) }                            // segment.rs:162
        pub fn purge(&mut self) -> Result<(), DBError> {                                                                   // segment.rs:166
            check(unsafe { ffi::vdb_segment_purge(self.s) })?;
            self.after_purge();
            Ok(())
        }
        pub fn hnsw(&self) -> &HNSWIndex { &self.hnsw }                                                                     // segment.rs:252
        pub fn payloads(&self) -> &HashMap<PointId, Payload> { &self.payloads }                                            // segment.rs:257
        pub fn payload_index(&self) -> &PayloadIndex { &self.payload_index }                                               // segment.rs:261
    }
}
