//! `vectordb` (rosharma719/vectordb, src/lib.rs:1-4) with the search path on an MI355X.
//!
//! `utils` and `payload_storage` are the reference's modules.  `vector` and `segment` keep the reference's names and
//! signatures; every distance, graph walk and filter mask runs in libvdbmi (include/vdbmi.h), and the wrappers keep
//! the host-side mirrors the API hands out by reference (`get_vector`, `payloads`, `payload_index`).
pub mod ffi;
pub mod segment;
pub mod shard;
pub mod vector;

pub use vectordb::payload_storage;
pub use vectordb::utils;
