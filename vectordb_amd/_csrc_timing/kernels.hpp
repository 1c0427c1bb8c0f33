// Host-visible launchers of the gfx950 kernels (definitions in kernels.hip).
#pragma once
#include "common.hpp"

namespace vdb {

constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr uint32_t MAX_EF = 1024;          // result list lives in LDS: 8 B per entry
constexpr uint32_t MAX_DIM = 2048;         // 64 lanes x 8 chunks x 4 floats

constexpr uint32_t ADJ_BLOCK = 64;         // dwords per layer-0 adjacency block
constexpr uint32_t ADJ_INLINE = 62;        // neighbours held in the block itself
// Packed adjacency descriptor: low 40 bits = first index into the neighbour pool, high 24 = degree.
__host__ __device__ inline uint64_t make_desc(uint64_t off, uint32_t deg) { return off | ((uint64_t)deg << 40); }
constexpr uint32_t UREF_BITS = 26;         // packed upper reference: low 26 bits first descriptor, high 6 bits list count
__host__ __device__ inline uint32_t make_uref(uint32_t base, uint32_t cnt) { return cnt ? (base | (cnt << UREF_BITS)) : 0u; }

// Read-only view of the device-resident graph handed to every kernel.
struct GraphView {
    const float* vecs;          // n x stride floats, rows zero-padded to a multiple of 4
    uint32_t stride, dim, n;
    const uint64_t* ids;        // slot -> PointId
    const uint32_t* adj0;       // layer 0: one 64-dword block per slot [degree, 62 neighbours, overflow start]
    const uint32_t* ovf;        // layer 0: neighbours beyond the 62nd of high-degree nodes
    const uint32_t* upper_base; // per slot: first index into udesc, NONE32 when the slot owns no upper list
    const uint8_t* upper_cnt;   // per slot: lists owned at layers 1..upper_cnt
    const uint64_t* udesc;
    const uint32_t* pool;       // upper layers: neighbour slots (all lists: self-loops and duplicates removed)
    // Fused upper-layer blocks, one per descriptor (index upper_base[slot] + level - 1), or nullptr:
    // ublk: 64 dwords [degree, first 62 neighbours, pool index of the 63rd]; ublk_ref: for the
    // neighbour in the same word its packed (upper_base | upper_cnt << 26), so a greedy move needs
    // no look-up before the next block can be requested.
    const uint32_t* ublk; const uint32_t* ublk_ref;
    const uint32_t* deleted;    // bitmap (1 = tombstone) or nullptr when nothing is deleted
    int metric;
    // Exact k-NN over a subset (filtered search): position p of the scan stands for slot rowmap[p] and n counts
    // positions; nullptr = every slot in order.  The rows are read where they lie, nothing is copied.
    const uint32_t* rowmap;
};

enum SearchMode : int { MODE_SEARCH = 0, MODE_BUILD = 1, MODE_MASKED = 2, MODE_LITERAL = 3 };

struct SearchArgs {
    GraphView g;
    int mode, arith;
    const float* queries;        // nq x dim floats (MODE_BUILD: unused)
    const uint32_t* query_slots; // MODE_BUILD: the stored row that acts as the query
    uint32_t nq;
    uint32_t entry, max_level;   // entry slot (NONE32: empty index)
    uint32_t fallback_live;      // smallest-id live slot, used when the entry is tombstoned
    uint32_t ef, top_k;
    uint32_t beam_width;         // register-resident beam: nodes expanded per step (1..4)
    uint32_t force_generic;      // 1: LDS-resident list with HBM spill tables even where the register beam applies
    uint32_t team;               // 0: one wavefront per query; 2 / 4: a team of that many wavefronts per query (search_team.hpp)
    uint32_t team_width;         // team: nodes expanded per step in total (1 .. 4 x team)
    vdb_scored_point* out;       // nq x top_k
    uint32_t* out_n;             // nq
    vdb_query_stats* stats;      // nq or nullptr
    uint32_t hash_log2;          // LDS visited table capacity = 1 << hash_log2 ids
    uint32_t tag_log2;           // register-resident beam: 16-bit visited table of 1 << tag_log2 entries (slots below 2^(tag_log2 + 13))
    uint32_t* spill_pool;        // spill tables in HBM for queries that outgrow the LDS table
    uint32_t* spill_locks;       // one word per spill table: 0 free, 1 taken (zeroed before the launch)
    uint32_t spill_tables, spill_log2;
    // MODE_BUILD
    const uint32_t* q_levels;    // node level per query
    const uint32_t* nb_off;      // prefix sum of (level + 1) per query
    uint32_t m;
    uint32_t* nb_out;            // [(nb_off[q] + l) * m + j]
    uint32_t* nb_cnt;            // [nb_off[q] + l]
    // MODE_MASKED / MODE_LITERAL
    const uint32_t* pass_bits;   // filter passes (errors count as false)
    const uint32_t* err_bits;    // evaluate_filter returned Err
    const uint32_t* q_entry;     // MODE_LITERAL: per-query entry (nullptr: args.entry)
    vdb_scored_point* cand_scratch;   // MODE_LITERAL: nq x cand_cap candidate queue
    uint32_t cand_cap;
};

// flags reported per query in vdb_query_stats.flags
enum : uint32_t { QF_HASH_OVERFLOW = 1u, QF_FILTER_ERROR = 2u, QF_CAND_OVERFLOW = 4u };

// 64-entry chunks of the register-resident layer-0 result list (FAST arithmetic; unfiltered search up to ef 512 in
// lists of 1, 2, 3, 4, 6 or 8 chunks; graph construction and the masked filtered walk up to ef 256); 0 = the list
// lives in LDS.
inline uint32_t reg_list_chunks(const SearchArgs& a) {
    // (the masked walk keeps its passing list in one 64-entry register chunk: top_k up to 64)
    const bool mode_ok = a.mode == MODE_SEARCH || a.mode == MODE_BUILD || (a.mode == MODE_MASKED && a.top_k <= 64);
    const uint32_t ef_max = a.mode == MODE_SEARCH ? 512u : 256u;
    if (a.force_generic || ((uint64_t)a.g.n >> (a.tag_log2 + 13)) != 0 || a.arith != VDB_ARITH_FAST || !mode_ok || a.ef == 0 || a.ef > ef_max) return 0;
    const uint32_t c = (a.ef + 63u) / 64u;
    return c <= 4 ? c : (c <= 6 ? 6u : 8u);
}

void launch_search(const SearchArgs& a, hipStream_t s);
// Wavefronts per query a launch of nq unfiltered FAST searches takes (0 = one): small launches cannot fill the chip
// with one wavefront per query and run each walk on a team instead.
uint32_t team_size_for(const SearchArgs& a);
size_t search_lds_bytes(const SearchArgs& a);

// out[q * n + i] = score(queries[q], base[i]); queries nq x dim (dense), base rows at `stride`.
void launch_score_scan(const float* queries, uint32_t nq, const float* base, uint32_t n, uint32_t dim, uint32_t stride,
                       int metric, int arith, float* out, hipStream_t s);
// raw[i] = score(vecs[query_slot], vecs[slots[i]]) for i < n (stored rows on both sides).
void launch_score_rows(const GraphView& g, uint32_t query_slot, const uint32_t* slots, uint32_t n, int arith, float* raw, hipStream_t s);
// raw[i] = score(vecs[query_slots[i]], vecs[slots[i]]) for i < n: independent pairs of stored rows, strict arithmetic.
void launch_score_pairs(const GraphView& g, const uint32_t* query_slots, const uint32_t* slots, uint32_t n, float* raw, hipStream_t s);
// In-place row normalisation (hnsw.rs:752-764): rows first..first+n of `vecs`; zero-norm rows stay.
void launch_normalize_rows(float* vecs, uint32_t stride, uint32_t dim, uint32_t first, uint32_t n, hipStream_t s);
// Exact k-NN (strict arithmetic, (sort_key asc, id asc)) over live rows.
void launch_exact_knn(const GraphView& g, const float* queries, uint32_t nq, uint32_t top_k, vdb_scored_point* out, uint32_t* out_n,
                      vdb_scored_point* scratch, uint32_t n_chunks, hipStream_t s);
// adj0 block idx[i] <- blocks[i] (64 dwords each): device copy of the layer-0 lists edited on the host.
void launch_patch_blocks(uint32_t* adj0, const uint32_t* idx, const uint32_t* blocks, uint32_t n, hipStream_t s);
// dst[idx[i]] = val[i] for i < n (distinct idx).
void launch_patch_words(uint32_t* dst, const uint32_t* idx, const uint32_t* val, uint32_t n, hipStream_t st);
// adj0 block slot[i]: neighbour word 1 + pos[i] <- val[i], degree word <- max(degree, pos[i] + 1): appends to layer-0 lists.
void launch_patch_entries(uint32_t* adj0, const uint32_t* slot, const uint32_t* pos, const uint32_t* val, uint32_t n, hipStream_t s);
uint32_t exact_knn_chunks(uint32_t n);
// FAST exact k-NN: fp32 MFMA candidate pass + strict re-rank (knn_mfma.hip).  scratch holds knn_mfma_scratch_words() dwords.
bool knn_mfma_supported(const GraphView& g, uint32_t top_k);
size_t knn_mfma_scratch_words(uint32_t n, uint32_t nq, uint32_t stride);
// hint: out / out_n already hold an approximate answer of top_k actual rows per query (the graph's); the candidate pass starts
// from the bounds it gives.  The result does not depend on it.
void launch_exact_knn_mfma(const GraphView& g, const float* queries, uint32_t nq, uint32_t top_k, vdb_scored_point* out, uint32_t* out_n,
                           uint32_t* scratch, hipStream_t s, bool hint = false);
// The same pass on a static scratch area inside the library's code object (no allocation; 16 MiB per GPU, shared by every
// call on that GPU): throws when 2 * rows + queries * candidates-per-chunk words do not fit.
void launch_exact_knn_mfma_static(const GraphView& g, const float* queries, uint32_t nq, uint32_t top_k, vdb_scored_point* out, uint32_t* out_n,
                                  hipStream_t s);
size_t knn_mfma_static_words();
// ---- payload filters on the device (filter_kernels.hip) ----
enum : int { FILTER_LEAF_MATCH = 0, FILTER_LEAF_COMPARE = 1 };
enum : int { FILTER_AND = 0, FILTER_OR = 1, FILTER_NOT = 2, FILTER_TRUE = 3, FILTER_FALSE = 4 };
struct FilterLeafArgs {
    const uint8_t* kind;        // per slot of the key's column: 0 = the payload lacks the key, else 1 + vdb_value_kind
    const int64_t* bits;        // per slot: Int value, f64 bit pattern, Bool as 0 / 1, dictionary code of a Str
    uint32_t n_col;             // slots the column covers (further slots lack the key)
    uint32_t n, words;          // slots of the segment, bitmap words
    int leaf, op, want_kind;    // FILTER_LEAF_*, vdb_cmp_op, kind of the filter's value
    int64_t want_bits;          // the filter's value in the encoding of `bits`
    const uint8_t* table;       // Str values: outcome per dictionary code (computed once per leaf on the host)
    uint32_t table_n;
    const uint32_t* has;        // bitmap of the slots that own a payload
    uint32_t* pass; uint32_t* err;
};
void launch_filter_leaf(const FilterLeafArgs& a, hipStream_t s);
void launch_filter_scatter(const uint32_t* slots, uint32_t m, uint32_t* pass, hipStream_t s);
void launch_filter_patch(const uint32_t* slots, const uint8_t* tri, uint32_t m, uint32_t* pass, uint32_t* err, hipStream_t s);
// acc (ap, ae) <- acc OP b; `has` = slots that own a payload (Not and the constants are confined to them)
void launch_filter_combine(int op, uint32_t* ap, uint32_t* ae, const uint32_t* bp, const uint32_t* be, const uint32_t* has, uint32_t words, hipStream_t s);
// clears bits beyond n and adds the live (not tombstoned) pass / err counts to counts[0] / counts[1] (zero them first)
void launch_filter_count(uint32_t* pass, uint32_t* err, const uint32_t* deleted, uint32_t n, uint32_t* counts, hipStream_t s);
// rows <- ascending slots with pass & ~deleted; block_sums holds filter_compact_blocks(n) words of scratch
uint32_t filter_compact_blocks(uint32_t n);
void launch_filter_compact(const uint32_t* pass, const uint32_t* deleted, uint32_t n, uint32_t* block_sums, uint32_t* rows, hipStream_t s);

void launch_topk_merge(const vdb_scored_point* lists, const uint32_t* counts, uint32_t n_shards, uint32_t nq, uint32_t k,
                       vdb_scored_point* out, uint32_t* out_n, hipStream_t s);

}  // namespace vdb
