// Host side of libvdbmi: the HNSW index (graph producer + device-resident search) and the segment
// wrapper, mirroring src/vector/hnsw.rs and src/segment/segment.rs.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <set>
#include <unordered_map>
#include <vector>

#include "kernels.hpp"
#include "payload.hpp"

namespace vdb {

// Device allocation that grows geometrically and keeps its contents.
template <class T> struct DevBuf {
    T* p = nullptr; size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void reserve(size_t n, hipStream_t s, bool keep = true) {
        if (n <= cap) return;
        size_t ncap = std::max<size_t>(n, cap + cap / 2 + 64);
        T* np = nullptr;
        VDB_HIP(hipMalloc(&np, ncap * sizeof(T)));
        if (p) {
            if (keep && cap) VDB_HIP(hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s));
            VDB_HIP(hipStreamSynchronize(s));
            VDB_HIP(hipFree(p));
        }
        p = np; cap = ncap;
    }
};

struct HostLayer {                       // layers[l]: HashMap<PointId, Vec<PointId>> of hnsw.rs:60, by slot
    std::vector<int32_t> idx;            // slot -> list index, -1 = the slot owns no list here
    std::vector<std::vector<uint32_t>> lists;
    std::vector<uint32_t> owner;
};

// A Filter compiled against the segment's payloads: device bitmaps over the slots.
struct FilterBits {
    DevBuf<uint32_t> d_pass, d_err;      // filter true / evaluate_filter returned Err; a slot without payload has neither bit
    DevBuf<uint32_t> d_rows;             // live passing slots, ascending (built when the narrow-filter scan will use it)
    bool have_rows = false;
    size_t n_slots = 0, n_pass = 0, n_err = 0;   // counts over live (not tombstoned) slots
    uint64_t version = 0;
    uint64_t last_used = 0;              // cache clock of the segment (least recently used entries are dropped)
};

// One payload key as columns over the slots (host mirror + device copy, appended to as points arrive).
struct PayloadColumn {
    std::vector<uint8_t> kind;           // 0 = the slot's payload lacks the key (or the slot has no payload), else 1 + vdb_value_kind
    std::vector<int64_t> bits;           // Int value / f64 bit pattern / Bool as 0, 1 / dictionary code of a Str / 0 for lists
    std::vector<uint32_t> list_slots;    // slots whose value is a list: the leaves that can be true for a list are finished on the host
    std::map<std::string, uint32_t> str_code; std::vector<std::string> strs;   // dictionary of the key's Str values
    DevBuf<uint8_t> d_kind; DevBuf<int64_t> d_bits; size_t on_device = 0;
};

}  // namespace vdb

struct vdb_index {
    // configuration (hnsw.rs:59-73)
    vdb_metric metric; size_t m, ef, max_level_cap, dim; double level_scale;
    vdb_options opts; uint32_t stride;
    // host mirror of the graph — the literal structures the reference keeps
    std::vector<uint64_t> ids; std::unordered_map<uint64_t, uint32_t> slot_of;
    std::vector<uint32_t> levels; std::vector<float> hvecs;     // n x dim, stored form
    std::vector<vdb::HostLayer> layers;
    std::vector<uint8_t> deleted; size_t n_deleted = 0; std::set<uint64_t> live;
    std::set<uint64_t> pending_deleted;      // ids marked deleted before they were inserted (hnsw.rs:112 takes any id)
    std::optional<uint64_t> entry_point; size_t current_max_level = 0;
    uint64_t level_draws = 0, sample_draws = 0;
    // device state
    hipStream_t stream = nullptr;
    vdb::DevBuf<float> d_vecs; vdb::DevBuf<uint64_t> d_ids, d_udesc;
    vdb::DevBuf<uint32_t> d_adj0, d_ovf, d_upper_base, d_pool, d_deleted, d_ublk, d_ublk_ref;
    bool have_ublk = false; vdb::DevBuf<uint8_t> d_upper_cnt;
    size_t rows_on_device = 0; bool graph_dirty = true, deleted_dirty = true;
    // incremental device copy of layer 0: lists edited since the last sync, the degree each block holds on the
    // device, the part of `ovf` in use; upper layers are re-packed as a whole when any of their lists changed
    std::vector<uint32_t> dirty0, h_deg0, clean_stamp; std::vector<uint8_t> dirty0_mark;
    uint32_t clean_epoch = 0; bool upper_dirty = true, need_full = true;
    size_t ovf_live = 0;                                      // entries of `ovf` that a block still points at
    size_t ovf_used = 0, upper_words = 0, upper_slots = 0, upper_lists_packed = 0; uint64_t entry_at_pack = ~0ull;
    // host mirror of the packed upper layers, kept so that a few changed lists can be patched instead of re-packing
    std::vector<uint32_t> h_upper_base; std::vector<uint8_t> h_upper_cnt; std::vector<uint64_t> h_udesc;
    std::vector<uint64_t> dirty_upper;          // (level << 32 | slot) of upper lists changed since the last sync, with repeats
    bool dirty_upper_lost = false;              // too many to remember: re-pack
    size_t pool_used = 0, pool_live = 0;        // words of `pool` in use on the device / referenced by a descriptor
    vdb::DevBuf<uint32_t> s_delta;
    // appends to layer-0 lists whose device block is otherwise current: (slot, position in the packed list, neighbour)
    std::vector<uint32_t> delta_slot, delta_pos, delta_val;
    void verify_layer0();                                    // VDB_VERIFY_DEVICE=1: device blocks against a fresh pack of the host lists
    size_t l0_edges = 0, upper_edges = 0, device_bytes = 0, l0_edges_acc = 0;
    // per-call scratch (guarded by mu)
    mutable std::mutex mu;
    mutable void* h_pin = nullptr; mutable size_t h_pin_bytes = 0;     // pinned staging of small search batches (one query per call is the reference's shape)
    void* pinned(size_t bytes) const;
    mutable vdb::DevBuf<float> s_q, s_raw; mutable vdb::DevBuf<vdb_scored_point> s_out, s_cand;
    mutable vdb::DevBuf<uint32_t> s_n, s_slots, s_qslots, s_lvl, s_off, s_nb, s_nbc, s_bin, s_bout, s_hash, s_spill, s_spill_lit; mutable vdb::DevBuf<vdb_query_stats> s_stats;
    // Scratch per stream (per-query flags, exact k-NN candidates): device-buffer calls return before their kernels have
    // run, so calls on different streams must not share it.  At most STREAM_SCRATCH_MAX streams are remembered; the
    // least recently used entry goes after a device-wide synchronisation (its stream may no longer exist, so it cannot
    // be waited for by handle).
    struct StreamScratch { vdb::DevBuf<vdb_query_stats> stats; vdb::DevBuf<uint32_t> knn_words; vdb::DevBuf<vdb_scored_point> knn_lists; uint64_t last_used = 0; };
    static constexpr size_t STREAM_SCRATCH_MAX = 16;
    mutable std::map<hipStream_t, StreamScratch> s_by_stream; mutable uint64_t s_stream_clock = 0;
    StreamScratch& stream_scratch(hipStream_t s) const;       // mu held

    vdb_index(vdb_metric metric, size_t m, size_t ef, size_t cap, size_t dim, const vdb_options& o);
    ~vdb_index();

    size_t len() const { return ids.size(); }
    float normalize_score(float raw) const { return metric == VDB_DOT ? -raw : raw; }
    const std::vector<uint32_t>* list(size_t level, uint32_t slot) const;
    std::vector<uint32_t>& list_or_default(size_t level, uint32_t slot);
    std::optional<uint32_t> first_live_slot() const;
    void use_device() const;

    // graph producer
    void insert(uint64_t id, const float* v, size_t len);
    void insert_batch(const uint64_t* pids, const float* v, size_t n, size_t len);
    void mark_deleted(uint64_t id);
    void add_bidirectional_edge(size_t level, uint32_t a, uint32_t b);
    void import_graph(size_t n, const uint64_t* pids, const uint32_t* lv, const float* vecs, const uint8_t* del,
                      size_t n_layers, const vdb_layer_csr* csr, bool has_entry, uint64_t entry, size_t cur_max);

    // device
    // upload rows / packed graph / tombstones if stale.  for_build: a round of the batched builder may search upper
    // layers that lack the newest few percent of their nodes (they are re-packed as a whole, which costs O(n))
    // upper_slack: 0 = the device copy must be complete; k = packed upper layers may lack up to 1/k of their lists
    static constexpr uint32_t UPPER_SLACK_BUILD = 16;
    void sync_device(uint32_t upper_slack = 0);
    void pack_layer0_full(); void patch_layer0(); void pack_upper(); bool patch_upper(); void verify_upper();
    void clean_list(const std::vector<uint32_t>& src, uint32_t self, std::vector<uint32_t>& out);
    static void fill_block(uint32_t* blk, const std::vector<uint32_t>& lst, std::vector<uint32_t>& ovf, size_t ovf_base);
    vdb::GraphView view() const;
    struct FilterArgs { int mode = vdb::MODE_SEARCH; const vdb::FilterBits* bits = nullptr; uint32_t entry_slot = vdb::NONE32; };
    void search_batch(const float* q, size_t nq, size_t top_k, size_t ef_search, const FilterArgs& fa,
                      vdb_scored_point* out, uint32_t* out_n, vdb_query_stats* stats);
    // search_batch_dev with the index mutex already held
    void search_dev_locked(const float* d_q, size_t nq, size_t top_k, size_t ef_search, vdb_scored_point* d_out, uint32_t* d_out_n,
                           vdb_query_stats* d_stats, hipStream_t s, const FilterArgs* fa);
    void search_batch_dev(const float* d_q, size_t nq, size_t top_k, size_t ef_search, vdb_scored_point* d_out, uint32_t* d_out_n,
                          vdb_query_stats* d_stats, hipStream_t s, const FilterArgs* fa = nullptr);
    // rowmap (device, n_rows live slots): exact k-NN over those slots only, read where they lie
    void exact_knn(const float* q, size_t nq, size_t top_k, bool dev, vdb_scored_point* out, uint32_t* out_n, hipStream_t s,
                   const uint32_t* rowmap = nullptr, size_t n_rows = 0);
    // exact k-NN on device buffers without any allocation (static scratch of the library; FAST index, top_k <= 28, no pending edits)
    void exact_knn_static(const float* d_q, size_t nq, size_t top_k, vdb_scored_point* d_out, uint32_t* d_out_n, hipStream_t s);
    std::vector<float> score_rows(uint32_t query_slot, const std::vector<uint32_t>& slots);
    std::vector<float> score_all_rows(uint32_t query_slot);
    std::vector<vdb_scored_point> search_from_row(uint32_t row_slot, uint32_t entry_slot, size_t ef_search);
    // the same for many stored rows in one launch: out holds rows.size() x ef_search records, cnt the valid ones per row
    void search_from_rows(const std::vector<uint32_t>& rows, uint32_t entry_slot, size_t ef_search,
                          std::vector<vdb_scored_point>& out, std::vector<uint32_t>& cnt);
    // raw[i] = score(row query_slots[i], row slots[i]) in one launch
    std::vector<float> score_pairs(const std::vector<uint32_t>& query_slots, const std::vector<uint32_t>& slots);
    uint32_t greedy_unfiltered(const float* q, uint32_t entry_slot, size_t level);
    std::vector<float> maybe_normalize(const float* v);

 private:
    uint32_t append_rows(const uint64_t* pids, const float* v, size_t n);
    size_t draw_level();
    struct BuildResult { std::vector<uint32_t> off, nb, cnt; };
    BuildResult build_search(const std::vector<uint32_t>& slots, uint32_t entry_slot, bool lazy_upper = false);
    void link_point(uint32_t slot, const BuildResult& br, size_t qi);
    uint32_t pick_hash_log2(size_t ef, bool& global) const;
    uint32_t beam_width() const;       // nodes expanded per step of the FAST layer-0 beam (vdb_options.beam_width, VDB_BEAM_WIDTH)
    void run_search(vdb::SearchArgs a, size_t nq, hipStream_t s, bool retry_on_overflow, const float* d_q_all);
};

struct vdb_segment {
    std::unique_ptr<vdb_index> hnsw; vdb_payload_index payload_index;
    std::unordered_map<uint64_t, vdb_payload> payloads; std::set<uint64_t> deleted; uint64_t next_id = 1;   // segment.rs:21-29
    uint64_t payload_version = 1;
    // Compiled filters by canonical text.  Searches are const and may run from several host threads: the cache is
    // guarded by filter_mu, every entry is built as a new object and handed out as a shared pointer, so a search
    // keeps its bitmaps alive while another thread replaces or evicts the entry.  At most FILTER_CACHE_MAX entries.
    static constexpr size_t FILTER_CACHE_MAX = 16;
    mutable std::mutex filter_mu;
    mutable std::map<std::string, std::shared_ptr<vdb::FilterBits>> filter_cache;
    mutable uint64_t filter_clock = 0;
    // Bitmaps that launches queued on a caller's stream still read: held here, per stream, until an event recorded
    // behind those launches has completed (checked on that stream's next call), whatever happens to the cache entry.
    struct InFlight { std::shared_ptr<const vdb::FilterBits> bits; hipEvent_t done; };
    mutable std::map<hipStream_t, std::vector<InFlight>> in_flight; mutable std::vector<hipEvent_t> spare_events;
    void hold_until_done(std::shared_ptr<const vdb::FilterBits> bits, hipStream_t s) const;
    ~vdb_segment();

    // payload columns and the bitmap of slots that own a payload: what the device-side filter compiler reads
    mutable std::map<std::string, vdb::PayloadColumn> columns;
    mutable std::vector<uint32_t> has_bits; mutable vdb::DevBuf<uint32_t> d_has; mutable bool has_dirty = true; mutable size_t has_words_on_device = 0;
    void note_payload(uint32_t slot, const vdb_payload& p);
    void rebuild_columns();

    explicit vdb_segment(vdb_index* h) : hnsw(h) {}
    uint64_t insert(const float* v, size_t len, const vdb_payload* p);
    void insert_batch(const float* v, size_t n, size_t len, const vdb_payload* const* p, uint64_t* out_ids);
    const float* get_vector(uint64_t id) const;
    void del(uint64_t id);
    void purge();
    size_t live_count() const { return hnsw->len() - deleted.size(); }
    std::vector<vdb_scored_point> search(const float* q, size_t len, size_t top_k) const;
    std::vector<vdb_scored_point> search_with_filter(const float* q, size_t len, size_t top_k, const vdb_filter* f) const;
    std::vector<vdb_scored_point> post_filter(const float* q, size_t len, size_t top_k, const vdb_filter* f) const;
    std::vector<vdb_scored_point> in_place_filtered_search(const float* q, size_t len, size_t top_k, const vdb_filter* f) const;
    void search_batch(const float* q, size_t nq, size_t len, size_t top_k, size_t ef_search, const vdb_filter* f,
                      vdb_scored_point* out, uint32_t* out_n) const;
    // the same on device buffers, enqueued on `s` without a host synchronisation (masked filter mode when f is given)
    void search_batch_dev(const float* d_q, size_t nq, size_t top_k, size_t ef_search, const vdb_filter* f,
                          vdb_scored_point* d_out, uint32_t* d_out_n, vdb_query_stats* d_stats, hipStream_t s) const;
    std::shared_ptr<const vdb::FilterBits> compile(const vdb_filter& f) const;
    // host copies of a compiled filter's bitmaps ((len + 31) / 32 words each): what the tests compare with evaluate_filter
    void filter_bitmaps(const vdb_filter& f, uint32_t* pass, uint32_t* err) const;
    void build_filter_aware_edges(uint64_t id, const vdb_payload& p, const std::vector<std::string>& keys);
    void filter_aware_edges_batch(const std::vector<uint64_t>& pids, const vdb_payload* const* p);

 private:
    // useful_k: leading results the caller keeps when none of them is a tombstone (top_k may be an over-fetch)
    void filtered_batch(const float* q, size_t nq, size_t len, size_t top_k, size_t useful_k, size_t ef_search, const vdb_filter* f,
                        vdb_scored_point* out, uint32_t* out_n) const;
    std::optional<uint64_t> find_entry_point_matching_filter(const vdb_filter& f) const;
};
