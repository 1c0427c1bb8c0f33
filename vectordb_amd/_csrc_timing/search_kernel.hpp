// HNSW search on one wavefront per query: greedy descent over the upper layers (hnsw.rs:333-368),
// then the ef-beam on one layer (hnsw.rs:420-511), plus the build-time, bitmap-masked and literal
// filtered variants.  MODE, the lane-group shape (G lanes x CH 16-byte chunks per row, PASSES row
// groups in flight), the arithmetic (STRICT) and optionally the metric are template parameters so
// that each instantiation carries only its own code and scalar state.
#pragma once
#include "device_common.hpp"

namespace vdb {
namespace {

#ifdef VDB_TIMING   // diagnostic build: per-phase shader-clock totals of the layer-0 beam
#define VDB_T_MARK(i) do { const unsigned long long _t = clock64(); tacc[i] += _t - tlast; tlast = _t; } while (0)
#else
#define VDB_T_MARK(i) do { } while (0)
#endif

template <int MODE, int G, int CH, int PASSES, bool STRICT, int METRIC, int NC>
struct Searcher {
#ifdef VDB_TIMING
    unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#endif
    static constexpr int GG = G > 0 ? G : 8, CC = CH > 0 ? CH : 1, PP = PASSES > 0 ? PASSES : 1;
    static constexpr int RPP = WAVE / GG;          // rows one wave instruction covers
    static constexpr int BATCH = RPP * PP;         // rows scored together by score_batch
    static constexpr int NCC = NC > 0 ? NC : 1;
    static constexpr int CAND_CAP = NC > 0 ? 256 : WAVE;   // candidate staging (cand / craw) entries
    uint2* perm = nullptr;                         // NC * 64 {key, slot} pairs: staging area of the register-resident list
    TagSet tags;
    const SearchArgs& a;
    const GraphView& gv;
    float* qlds; uint32_t* cand; float* craw;
    SortedList R, P;                      // R: beam / result list, P: filter-passing list (MODE_MASKED)
    Visited vis;
    QueryRegs<GG, CC> qr;
    uint32_t rows_scored = 0, expanded = 0, adj_entries = 0, flags = 0, err_slot = 0, spill_idx = NONE32;

    __device__ __forceinline__ Searcher(const SearchArgs& args) : a(args), gv(args.g) {}

    __device__ __forceinline__ int metric() const { return METRIC >= 0 ? METRIC : gv.metric; }
    __device__ __forceinline__ float key_of(float raw) const { return metric() == VDB_DOT ? -raw : raw; }    // hnsw.rs:103-108
    __device__ __forceinline__ bool is_deleted(uint32_t s) const { return gv.deleted && bit_test(gv.deleted, s); }

    __device__ __forceinline__ void reset_visited() {
        const int lane = lane_id();
        const uint4 empty4 = make_uint4(HASH_EMPTY, HASH_EMPTY, HASH_EMPTY, HASH_EMPTY);
        for (uint32_t i = lane; i <= vis.mask; i += WAVE) reinterpret_cast<uint4*>(vis.tab)[i] = empty4;
        vis.count = 0;
        if (vis.spill) { for (uint32_t i = lane; i <= vis.smask; i += WAVE) reinterpret_cast<uint4*>(vis.spill)[i] = empty4; __threadfence_block(); }
        vis.scount = 0;
        wave_sync();
    }
    __device__ __forceinline__ void note_visited(uint32_t m) { if (vis.spill) vis.scount += m; else vis.count += m; }
    // Called between neighbour chunks (all lanes): freezes the LDS table and takes a spill table
    // from the pool when it is nearly full.  false = no room anywhere; the query is abandoned
    // and flagged for the host.
    __device__ __forceinline__ bool visited_room() {
        if (!vis.spill) {
            if (vis.count <= vis.limit) return true;
            uint32_t idx = NONE32;
            if (a.spill_pool) {
                if (lane_id() == 0) {                  // any free table; a resident wave holds at most one
                    uint32_t t = (blockIdx.x * 2654435761u) % a.spill_tables;
                    for (uint32_t tries = 0; tries < a.spill_tables; ++tries) {
                        if (atomicCAS(&a.spill_locks[t], 0u, 1u) == 0u) { idx = t; break; }
                        t = (t + 1 == a.spill_tables) ? 0 : t + 1;
                    }
                }
                idx = readlane_u(idx, 0);
            }
            if (idx == NONE32) { flags |= QF_HASH_OVERFLOW; return false; }
            spill_idx = idx;
            vis.spill = a.spill_pool + ((size_t)idx << a.spill_log2);
            vis.smask = (1u << (a.spill_log2 - 2)) - 1; vis.sshift = 32 - (a.spill_log2 - 2);
            vis.slimit = (1u << a.spill_log2) - (1u << a.spill_log2) / 4 - (NC > 0 ? 4 * WAVE : WAVE);
            for (uint32_t i = lane_id(); i <= vis.smask; i += WAVE)
                reinterpret_cast<uint4*>(vis.spill)[i] = make_uint4(HASH_EMPTY, HASH_EMPTY, HASH_EMPTY, HASH_EMPTY);
            vis.scount = 0;
            __threadfence_block();
            wave_sync();
            return true;
        }
        if (vis.scount > vis.slimit) { flags |= QF_HASH_OVERFLOW; return false; }
        return true;
    }

    __device__ __forceinline__ void score(int m) {
        if (STRICT) score_cands_strict(gv, qlds, cand, m, craw);
        else score_cands_fast<GG, CC, PP, METRIC>(gv, qr, cand, m, craw);
        wave_sync();
        rows_scored += m;
    }
    __device__ __forceinline__ float score_one(uint32_t s) {
        if (lane_id() == 0) cand[0] = s;
        wave_sync();
        score(1);
        return craw[0];
    }

    // hnsw.rs:333-368: move to the FIRST neighbour that improves, rescan, at most 1000 steps.
    __device__ __forceinline__ uint32_t greedy(uint32_t entry, uint32_t level) {
        const int lane = lane_id();
        uint32_t cur = entry;
        float kcur = 0.f; bool have_cur = false;
        bool changed = true; int steps = 0;
        while (changed && steps < 1000) {
            ++steps; changed = false;
            const Adj A = open_adj(gv, level, cur);
            if (A.deg == 0) continue;
            ++expanded; adj_entries += A.deg;
            const uint32_t nchunks = A.chunks();
            for (uint32_t c = 0; c < nchunks && !changed; ++c) {
                uint32_t nb;
                const bool in = A.get(c, nb);
                const bool ok = in && !is_deleted(nb);
                const uint64_t mk = ballot(ok);
                const int m = popc64(mk);
                if (m == 0) continue;
                if (!have_cur) { kcur = key_of(score_one(cur)); have_cur = true; }
                if (ok) cand[popc64(mk & lanemask_lt(lane))] = nb;
                wave_sync();
                score(m);
                const float kl = lane < m ? key_of(craw[lane]) : 0.f;
                const uint64_t bm = ballot(lane < m && kl < kcur);
                if (bm) {
                    const int j = __ffsll((unsigned long long)bm) - 1;
                    cur = cand[j]; kcur = readlane_f(kl, j); changed = true;
                }
                wave_sync();
            }
        }
        return cur;
    }

    // greedy() on the fused upper-layer blocks: a move costs two dependent round trips (the node's
    // block, then the rows of its neighbours) because every neighbour carries the reference of its
    // own blocks.  `ref` is the packed reference of `cur` and follows it; the key of `cur` is
    // carried from level to level.  Same moves as greedy(): the first improving neighbour in list
    // order, at most 1000 steps per level.
    __device__ __forceinline__ uint32_t greedy_fused(uint32_t cur, uint32_t& ref, uint32_t level, float& kcur, bool& have_cur) {
        const int lane = lane_id();
        bool changed = true; int steps = 0;
        while (changed && steps < 1000) {
            ++steps; changed = false;
            if (level > (ref >> UREF_BITS)) break;                               // no list at this level
            const size_t blk = ((size_t)(ref & ((1u << UREF_BITS) - 1u)) + level - 1) * ADJ_BLOCK;
            const uint32_t w = gv.ublk[blk + lane], wr = gv.ublk_ref[blk + lane];
            const uint32_t deg = readlane_u(w, 0);
            if (deg == 0) break;
            ++expanded; adj_entries += deg;
            const uint32_t nchunks = deg <= ADJ_INLINE ? 1u : 1u + (deg - ADJ_INLINE + 63u) / 64u;
            const uint32_t* rest = gv.pool + readlane_u(w, 63);
            for (uint32_t c = 0; c < nchunks && !changed; ++c) {
                uint32_t nb, nr; bool in;
                if (c == 0) { nb = w; nr = wr; in = (uint32_t)(lane - 1) < (deg < ADJ_INLINE ? deg : ADJ_INLINE); }
                else {
                    const uint32_t i = (c - 1) * 64u + lane;
                    in = ADJ_INLINE + i < deg;
                    nb = in ? rest[i] : 0u;
                    nr = in ? make_uref(gv.upper_base[nb], gv.upper_cnt[nb]) : 0u;
                }
                const bool ok = in && !is_deleted(nb);
                const uint64_t mk = ballot(ok);
                const int m = popc64(mk);
                if (m == 0) continue;
                const uint32_t rnk = rank_below(mk);
                if (ok) cand[rnk] = nb;
                const bool with_cur = !have_cur && m < WAVE;                      // score `cur` in the same pass
                if (!have_cur && !with_cur) { kcur = key_of(score_one(cur)); have_cur = true; if (ok) cand[rnk] = nb; }
                if (with_cur && lane == 0) cand[m] = cur;
                wave_sync();
                int scored = m;
                if (STRICT || with_cur) score(with_cur ? m + 1 : m);
                else {
                    // in list order, one batch of rows at a time: the first batch that holds an improving
                    // neighbour decides the move, the rest of the list is never fetched
                    scored = 0;
                    while (scored < m) {
                        const int cnt = (m - scored) < BATCH ? (m - scored) : BATCH;
                        score_cands_fast<GG, CC, PP, METRIC>(gv, qr, cand + scored, cnt, craw + scored);
                        wave_sync();
                        rows_scored += cnt;
                        const bool mine = lane >= scored && lane < scored + cnt;
                        scored += cnt;
                        if (ballot(mine && key_of(craw[lane]) < kcur)) break;
                    }
                }
                if (with_cur) { kcur = key_of(craw[m]); have_cur = true; }
                const float kl = lane < scored ? key_of(craw[lane]) : 0.f;
                const uint64_t bm = ballot(lane < scored && kl < kcur);
                if (bm) {
                    const int j = __ffsll((unsigned long long)bm) - 1;
                    const uint64_t src = ballot(ok && rnk == (uint32_t)j);       // the lane that holds candidate j
                    const int sl = __ffsll((unsigned long long)src) - 1;
                    cur = readlane_u(nb, sl); ref = readlane_u(nr, sl); kcur = readlane_f(kl, j); changed = true;
                }
                wave_sync();
            }
        }
        return cur;
    }

    // hnsw.rs:370-418: scan the WHOLE list keeping the best filter-passing neighbour (strict <, so
    // the first of equal minima wins), repeat until a scan makes no move; no step cap.  A
    // neighbour whose filter evaluation is an Err aborts the search (hnsw.rs:392).
    __device__ __forceinline__ uint32_t greedy_filtered(uint32_t entry, uint32_t level) {
        const int lane = lane_id();
        uint32_t cur = entry;
        float kcur = 0.f; bool have_cur = false;
        bool changed = true;
        while (changed) {
            changed = false;
            const Adj A = open_adj(gv, level, cur);
            if (A.deg == 0) break;
            ++expanded; adj_entries += A.deg;
            uint32_t best = cur;
            const uint32_t nchunks = A.chunks();
            for (uint32_t c = 0; c < nchunks; ++c) {
                uint32_t nb;
                const bool in = A.get(c, nb);
                bool ok = in && !is_deleted(nb);
                if (a.pass_bits) {
                    const uint64_t em = ballot(ok && bit_test(a.err_bits, nb));
                    if (em) { flags |= QF_FILTER_ERROR; err_slot = readlane_u(nb, __ffsll((unsigned long long)em) - 1); return cur; }
                    ok = ok && bit_test(a.pass_bits, nb);
                }
                const uint64_t mk = ballot(ok);
                const int m = popc64(mk);
                if (m == 0) continue;
                if (!have_cur) { kcur = key_of(score_one(cur)); have_cur = true; }
                if (ok) cand[popc64(mk & lanemask_lt(lane))] = nb;
                wave_sync();
                score(m);
                float bk = (lane < m) ? key_of(craw[lane]) : 0.f;
                int bj = (lane < m && bk < kcur) ? lane : WAVE;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const float ok_ = __shfl_xor(bk, off); const int oj = __shfl_xor(bj, off);
                    if (oj < WAVE && (bj == WAVE || ok_ < bk || (ok_ == bk && oj < bj))) { bk = ok_; bj = oj; }
                }
                if (bj < WAVE) { best = cand[bj]; kcur = bk; changed = true; }
                wave_sync();
            }
            cur = best;
        }
        return cur;
    }

    // hnsw.rs:420-511 on `level`, starting from `entry`.  On return R holds the results sorted.
    // MASKED additionally keeps the ef best filter-passing nodes in P.  FUSED: layer 0.
    template <bool MASKED, bool FUSED> __device__ __forceinline__ void beam(uint32_t entry, uint32_t level, uint32_t ef) {
        const int lane = lane_id();
        reset_visited();
        R.n = 0; R.cap = ef;
        if (MASKED) { P.n = 0; P.cap = ef; }
        wave_sync();
        uint32_t start = entry;
        if (is_deleted(entry) && a.fallback_live != NONE32) start = a.fallback_live;      // hnsw.rs:440-444
        const float raw0 = score_one(start);
        R.insert(key_of(raw0), start);
        if (lane == 0) vis.insert(start);
        note_visited(1);
        if (MASKED && !is_deleted(start) && (!a.pass_bits || bit_test(a.pass_bits, start))) P.insert(key_of(raw0), start);
        wave_sync();
        // The adjacency block of the runner-up is requested while the current node is processed:
        // after the merge it usually is the next node to expand and its neighbours are already
        // in registers (one dependent memory round trip per expansion instead of two).
        uint32_t pf_slot = NONE32, pf_word = 0;
        uint32_t scan_from = 0;               // every entry before this chunk start is expanded
#ifdef VDB_TIMING
        tlast = clock64();
#endif
        for (;;) {
            // best unexpanded entry of R == the candidate queue's minimum (evicted candidates can
            // only re-surface on an exact key tie); also note the second best
            int sel = -1;
            uint32_t cur = 0, cur2 = NONE32;
            for (uint32_t b = scan_from; b < R.n && cur2 == NONE32; b += WAVE) {
                const uint32_t i = b + lane;
                const uint32_t v = (i < R.n) ? R.val[i] : FLAG_EXPANDED;
                uint64_t um = ballot(!(v & FLAG_EXPANDED));
                if (um && sel < 0) { const int l = __ffsll((unsigned long long)um) - 1; sel = (int)b + l; cur = readlane_u(v, l); um &= um - 1; }
                if (um && sel >= 0) cur2 = readlane_u(v, __ffsll((unsigned long long)um) - 1);
            }
            if (sel < 0) break;
            scan_from = (uint32_t)sel & ~63u;
            if (lane == 0) R.val[sel] = cur | FLAG_EXPANDED;
            wave_sync();
            VDB_T_MARK(0);
            Adj A;
            if (FUSED) {
                const uint32_t w = (cur == pf_slot) ? pf_word : load_adj_word(gv, cur);
                A = open_adj0(gv, w);
                pf_slot = NONE32;
            } else A = open_adj(gv, level, cur);
            VDB_T_MARK(1);
            if (A.deg == 0) continue;
            ++expanded; adj_entries += A.deg;
            const uint32_t nchunks = A.chunks();
            uint32_t lowest = NONE32;          // smallest list position any candidate of this expansion took
            for (uint32_t c = 0; c < nchunks; ++c) {
                uint32_t nb;
                const bool in = A.get(c, nb);
                bool fresh = in && !is_deleted(nb);                           // hnsw.rs:475
                if (fresh) fresh = vis.insert(nb);
                const uint64_t mk = ballot(fresh);
                const int m = popc64(mk);
                VDB_T_MARK(2);
                if (m == 0) continue;
                note_visited(m);
                if (fresh) cand[popc64(mk & lanemask_lt(lane))] = nb;
                wave_sync();
                // request the runner-up's block right before the row gather: the two fly together
                // (issued any earlier, the wait for this node's own block would also wait for it)
                if (FUSED && cur2 != NONE32 && pf_slot == NONE32) { pf_slot = cur2; pf_word = load_adj_word(gv, cur2); }
                score(m);
                VDB_T_MARK(3);
                const float kj = (lane < m) ? key_of(craw[lane]) : 0.f;
                const uint32_t sj = (lane < m) ? cand[lane] : 0u;
                if (MASKED) P.offer(lane < m && (!a.pass_bits || bit_test(a.pass_bits, sj)), kj, sj);
                const uint32_t lo = R.offer(lane < m, kj, sj);                // hnsw.rs:486-499, in list order
                lowest = lo < lowest ? lo : lowest;
                VDB_T_MARK(4);
                if (!visited_room()) return;
                VDB_T_MARK(5);
            }
            if (lowest != NONE32 && (lowest & ~63u) < scan_from) scan_from = lowest & ~63u;
            if (FUSED && cur2 != NONE32 && pf_slot == NONE32) { pf_slot = cur2; pf_word = load_adj_word(gv, cur2); }   // nothing new was scored
        }
    }

    // Scores cand[0 .. m) into craw[0 .. m) (both LDS) for the register-resident beam: Euclidean
    // scores stay squared (the beam orders by them and takes the root once, when it hands the
    // list over), sums run on packed two-float FMAs.  Lanes beyond the last candidate repeat it,
    // so no lane is masked off.  FULL: a row fills all G x CH chunks, the stride is a constant.
    template <bool FULL> __device__ __forceinline__ void score_rows(int m) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const int lane = lane_id(), g = lane % GG, r = lane / GG;
        const uint32_t stride = FULL ? (uint32_t)(GG * CC * 4) : gv.stride;
        const uint32_t nchunks = stride >> 2;
        const float* const vecs = gv.vecs;
        for (int base = 0; base < m; base += BATCH) {
            float4 x[PP][CC];
            uint32_t cs[PP];
#pragma unroll
            for (int p = 0; p < PP; ++p) {
                const int ci = base + p * RPP + r;
                cs[p] = cand[ci < m ? ci : m - 1];
            }
#pragma unroll
            for (int p = 0; p < PP; ++p) {
                if (p == 0 || base + p * RPP < m) {
                    const float4* row = reinterpret_cast<const float4*>(vecs + (size_t)cs[p] * stride);
#pragma unroll
                    for (int j = 0; j < CC; ++j) {
                        const uint32_t c = g + GG * j;
                        x[p][j] = row[(FULL || c < nchunks) ? c : 0u];
                    }
                }
            }
#ifdef VDB_TIMING_SCORE   // diagnostic build: row wait and arithmetic of the scoring apart (phases 6 and 7)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            VDB_T_MARK(6);
#endif
#pragma unroll
            for (int p = 0; p < PP; ++p) {
                if (p == 0 || base + p * RPP < m) {
                    f2 acc = {0.f, 0.f}, accb = {0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < CC; ++j) {
                        float4 v = x[p][j];
                        if (!FULL && (uint32_t)(g + GG * j) >= nchunks) v = make_float4(0.f, 0.f, 0.f, 0.f);
                        const f2 q0 = {qr.q[j].x, qr.q[j].y}, q1 = {qr.q[j].z, qr.q[j].w};
                        const f2 v0 = {v.x, v.y}, v1 = {v.z, v.w};
                        if (metric() == VDB_EUCLIDEAN) {
                            const f2 d0 = q0 - v0, d1 = q1 - v1;
                            acc = __builtin_elementwise_fma(d0, d0, acc); acc = __builtin_elementwise_fma(d1, d1, acc);
                        } else {
                            acc = __builtin_elementwise_fma(q0, v0, acc); acc = __builtin_elementwise_fma(q1, v1, acc);
                            if (metric() == VDB_COSINE) { accb = __builtin_elementwise_fma(v0, v0, accb); accb = __builtin_elementwise_fma(v1, v1, accb); }
                        }
                    }
                    const float sum = group_sum<GG>(acc.x + acc.y);
                    float raw = sum;
                    if (metric() == VDB_COSINE) {
                        const float sb = group_sum<GG>(accb.x + accb.y);
                        raw = 1.0f - sum / (qr.qnorm * __builtin_sqrtf(sb) + 1e-10f);
                    }
                    const int ci = base + p * RPP + r;
                    if (g == 0 && ci < m) craw[ci] = raw;
                }
            }
#ifdef VDB_TIMING_SCORE
            VDB_T_MARK(7);
#endif
        }
        wave_sync();
    }

    // hnsw.rs:420-511 on layer 0 with the result list held in registers: entry c*64 + lane of the
    // ascending list lives in (rk[c], rv[c]) of that lane, empty positions hold (+inf, expanded).
    // Each step takes the `width` best unexpanded entries (a ballot over the expanded flags),
    // probes the visited table for all their inline neighbours at once, scores the unvisited
    // ones and merges them by rank — the outcome of one insert per candidate in list order
    // (hnsw.rs:486-499); the permutation goes through `perm` in LDS once per 64 candidates.
    // width == 1 is the reference's strictly best-first order.  Euclidean keys are squared
    // distances until the list is handed over in R (LDS).  The visited set is the TagSet in LDS,
    // which forgets ids instead of filling up (see TagSet): no query can outgrow it.
    template <bool MASKED> __device__ __forceinline__ void beam_reg(uint32_t entry, uint32_t ef, uint32_t width) {
        constexpr int EMAX = 4;
        const int lane = lane_id();
        const float INF = __int_as_float(0x7f800000);
        const bool full = gv.stride == (uint32_t)(GG * CC * 4);
        uint32_t* const par_lds = reinterpret_cast<uint32_t*>(craw);     // parents pass through the head of craw
        const TagSet ts = tags;
        const uint32_t* const adj0 = gv.adj0;
        float rk[NCC]; uint32_t rv[NCC];
#pragma unroll
        for (int c = 0; c < NCC; ++c) { rk[c] = INF; rv[c] = FLAG_EXPANDED; }
        const uint32_t cap = ef;
        // MASKED: the best filter-passing rows among everything scored, ascending, entry `lane` in this lane;
        // the walk itself does not depend on it (hnsw.rs has no such mode: extension of the masked filter mode)
        float pk = INF; uint32_t pv = 0, pn = 0; float p_worst = INF;
        const uint32_t pcap = a.top_k < (uint32_t)WAVE ? (a.top_k ? a.top_k : 1u) : (uint32_t)WAVE;
        auto offer_passing = [&](bool have, float kj, uint32_t sj) {
            bool ok = have && (pn < pcap || kj < p_worst);
            if (ok && a.pass_bits) ok = bit_test(a.pass_bits, sj);
            uint64_t om = ballot(ok);
            while (om) {
                const int l = __ffsll((unsigned long long)om) - 1; om &= om - 1;
                const float k = readlane_f(kj, l); const uint32_t s = readlane_u(sj, l);
                if (pn >= pcap && !(k < p_worst)) continue;
                if (ballot((uint32_t)lane < pn && pk == k && pv == s)) continue;          // scored twice (forgotten by the visited set)
                const uint32_t pos = popc64(ballot((uint32_t)lane < pn && pk <= k));      // equal keys: earlier arrivals first
                const float uk = __shfl_up(pk, 1); const uint32_t uv = (uint32_t)__shfl_up((int)pv, 1);
                if ((uint32_t)lane > pos) { pk = uk; pv = uv; }
                if ((uint32_t)lane == pos) { pk = k; pv = s; }
                pn = pn < pcap ? pn + 1 : pcap;
                if ((uint32_t)lane >= pn) { pk = INF; pv = 0; }
                p_worst = pn >= pcap ? readlane_f(pk, (int)pcap - 1) : INF;
            }
        };
        uint32_t n = 0;
        bool forgot = false;             // the visited set replaced an id: candidates are checked against the list from here on
        float worst = INF;
        {   // clear the visited set
            const uint32_t words = (ts.idmask >> 15) * 2 + 2;
            for (uint32_t i = lane * 4; i < words; i += WAVE * 4) *reinterpret_cast<uint4*>(ts.tab + i) = make_uint4(0u, 0u, 0u, 0u);
            wave_sync();
        }
        uint32_t start = entry;
        if (is_deleted(entry) && a.fallback_live != NONE32) start = a.fallback_live;      // hnsw.rs:440-444
        if (lane == 0) { cand[0] = start; ts.insert(start); }
        wave_sync();
        if (full) score_rows<true>(1); else score_rows<false>(1);
        rows_scored += 1;
        {
            const float k0 = key_of(craw[0]);
            if (lane == 0) { rk[0] = k0; rv[0] = start; }
            n = 1;
            if (cap == 1) worst = k0;
            if (MASKED) offer_passing(lane == 0 && !is_deleted(start), k0, start);
        }
        wave_sync();
#ifdef VDB_TIMING
        tlast = clock64();
        tacc[9] = wall_clock64();            // beam start (100 MHz ticks)
#endif
        for (;;) {
            // ---- the `width` best unexpanded entries, in list order: rank by ballot, ids through LDS ----
            uint32_t seen = 0;
#pragma unroll
            for (int c = 0; c < NCC; ++c) {
                if (seen < width && (uint32_t)(c * WAVE) < n) {
                    const bool un = (int32_t)rv[c] >= 0;
                    const uint64_t um = ballot(un);
                    const uint32_t rnk = seen + rank_below(um);
                    const bool isp = un && rnk < width;
                    if (isp) { par_lds[rnk] = rv[c]; rv[c] |= FLAG_EXPANDED; }
                    seen += popc64(um);
                }
            }
            const uint32_t np = seen < width ? seen : width;
            if (np == 0) break;
            wave_sync();
            const uint4 pv = *reinterpret_cast<const uint4*>(par_lds);
            const uint32_t par[EMAX] = {pv.x, pv.y, pv.z, pv.w};
            VDB_T_MARK(0);
            uint32_t w[EMAX];
#pragma unroll
            for (int e = 0; e < EMAX; ++e) w[e] = adj0[(size_t)par[(uint32_t)e < np ? e : 0] * ADJ_BLOCK + lane];
            wave_sync();                     // parents are consumed before cand/craw are written again
#ifdef VDB_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // charge the adjacency round trip to its own phase
#endif
            VDB_T_MARK(1);
            // ---- round 0: the inline neighbours (block words 1..62) of all parents at once; further
            // rounds: one overflow chunk of a high-degree parent each ----
            uint32_t hub_e = 0, hub_ch = 1, hub_left = 0;     // overflow chunks of high-degree parents still to visit
            for (uint32_t round = 0; ; ++round) {
                int m = 0;
                if (round == 0) {
                    uint32_t nb[EMAX], w0[EMAX], tag[EMAX], widx[EMAX], expect[EMAX], desired[EMAX], old[EMAX], half[EMAX];
                    uint2 bk[EMAX];
                    bool tr[EMAX], fresh[EMAX], slow[EMAX], full_b[EMAX];
#pragma unroll
                    for (int e = 0; e < EMAX; ++e) {
                        nb[e] = w[e];
                        ts.locate(nb[e], w0[e], tag[e]);
                        bk[e] = *reinterpret_cast<const uint2*>(ts.tab + w0[e]);
                    }
#pragma unroll
                    for (int e = 0; e < EMAX; ++e) {
                        uint32_t deg = (uint32_t)e < np ? readlane_u(w[e], 0) : 0u;
                        if (deg) { ++expanded; adj_entries += deg; }
                        hub_left += deg > ADJ_INLINE ? (deg - ADJ_INLINE + 63u) / 64u : 0u;
                        deg = deg < ADJ_INLINE ? deg : ADJ_INLINE;
                        bool want = (uint32_t)(lane - 1) < deg;                           // words 1..deg hold neighbours
                        if (gv.deleted) want = want && !bit_test(gv.deleted, nb[e]);      // hnsw.rs:475
                        const bool pres = TagSet::present(bk[e], tag[e]);
                        const uint32_t u = TagSet::used(bk[e]);
                        tr[e] = want & !pres;
                        full_b[e] = tr[e] & (u == TagSet::WAYS);                          // full bucket: one of its ids is forgotten
                        const uint32_t en = TagSet::entry_for(u, tag[e]);
                        TagSet::slot_of(bk[e], en, tag[e], widx[e], expect[e], desired[e]);
                        half[e] = (en & 1u) * 16u;
                    }
#ifdef VDB_TIMING
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                    VDB_T_MARK(6);
#pragma unroll
                    for (int e = 0; e < EMAX; ++e) { old[e] = 0; if (tr[e]) old[e] = atomicCAS(&ts.tab[w0[e] + widx[e]], expect[e], desired[e]); }
#ifdef VDB_TIMING
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                    VDB_T_MARK(7);
                    bool any_slow = false;
#pragma unroll
                    for (int e = 0; e < EMAX; ++e) {
                        fresh[e] = tr[e] & (old[e] == expect[e]);
                        // the word changed under us: if our entry now holds our tag, another lane carried the
                        // same id (a neighbour shared by two parents) and it counts as visited
                        const bool same = ((old[e] >> half[e]) & 0xFFFFu) == tag[e];
                        slow[e] = tr[e] & (old[e] != expect[e]) & !same;
                        any_slow = any_slow | slow[e] | full_b[e];
                    }
                    if (ballot(any_slow)) {
#pragma unroll
                        for (int e = 0; e < EMAX; ++e) {
                            int r = 0;
                            if (ballot(slow[e])) {
                                if (slow[e]) r = ts.insert(nb[e]);
                                if (slow[e]) fresh[e] = r != 0;
                            }
                            forgot = forgot | (ballot(full_b[e] | (r == 2)) != 0);
                        }
                    }
                    VDB_T_MARK(8);
#pragma unroll
                    for (int e = 0; e < EMAX; ++e) {
                        const uint64_t mk = ballot(fresh[e]);
                        if (fresh[e]) cand[m + rank_below(mk)] = nb[e];
                        m += popc64(mk);
                    }
                } else {
                    if (hub_left == 0) break;
                    --hub_left;
                    Adj A;
                    for (;;) {                                                            // next overflow chunk, in parent order
                        uint32_t we = w[0];
#pragma unroll
                        for (int t = 1; t < EMAX; ++t) if (hub_e == (uint32_t)t) we = w[t];
                        A = open_adj0(gv, we);
                        if (hub_ch < A.chunks()) break;
                        ++hub_e; hub_ch = 1;
                    }
                    uint32_t nb;
                    const bool in = A.get(hub_ch, nb);
                    ++hub_ch;
                    bool fresh = in && !is_deleted(nb);
                    int r = 0;
                    if (fresh) r = ts.insert(nb);
                    fresh = r != 0;
                    forgot = forgot | (ballot(r == 2) != 0);
                    const uint64_t mk = ballot(fresh);
                    if (fresh) cand[rank_below(mk)] = nb;
                    m = popc64(mk);
                }
                VDB_T_MARK(2);
                if (m == 0) continue;
                rows_scored += m;
                wave_sync();
                if (full) score_rows<true>(m); else score_rows<false>(m);
                VDB_T_MARK(3);
                // ---- merge cand[0..m) / craw[0..m), 64 candidates per round, lane j <-> candidate j ----
                // (all candidates are read into registers first: the permutation reuses their LDS area)
                float kjr[CAND_CAP / WAVE]; uint32_t sjr[CAND_CAP / WAVE];
#pragma unroll
                for (int r = 0; r < CAND_CAP / WAVE; ++r) {
                    const int j = r * WAVE + lane;
                    kjr[r] = j < m ? key_of(craw[j]) : INF;
                    sjr[r] = j < m ? cand[j] : 0u;
                }
                wave_sync();
                for (int base = 0; base < m; base += WAVE) {
                    const int j = base + lane;
                    const bool have = j < m;
                    float kj = kjr[0]; uint32_t sj = sjr[0];
#pragma unroll
                    for (int r = 1; r < CAND_CAP / WAVE; ++r) if (base == r * WAVE) { kj = kjr[r]; sj = sjr[r]; }
                    if (MASKED) offer_passing(have, kj, sj);
                    bool pass = have && (n < cap || kj < worst);                  // hnsw.rs:486
                    uint64_t pm = ballot(pass);
                    if (pm == 0) continue;
                    if (forgot) {
                        // a candidate the visited set had forgotten may already be in the list (same key, same id),
                        // or sit twice among the candidates: an id inserted by one lane can be replaced before
                        // the lane that carries it for another parent looks it up
                        uint64_t dup = 0, it = pm;
                        while (it) {
                            const int l = __ffsll((unsigned long long)it) - 1; it &= it - 1;
                            const float k = readlane_f(kj, l);
                            const uint64_t before = pm & ((1ull << l) - 1ull);
                            uint64_t eq = ballot(kj == k) & before;               // equal keys are rare: ids are compared only then
#pragma unroll
                            for (int c = 0; c < NCC; ++c) eq |= ballot(rk[c] == k);
                            if (eq) {
                                const uint32_t s = readlane_u(sj, l);
                                uint64_t hit = ballot(kj == k && sj == s) & before;
#pragma unroll
                                for (int c = 0; c < NCC; ++c) hit |= ballot(rk[c] == k && (rv[c] & ~FLAG_EXPANDED) == s);
                                if (hit) dup |= 1ull << l;
                            }
                        }
                        pm &= ~dup;
                        pass = pass && !((dup >> lane) & 1ull);
                        if (pm == 0) continue;
                    }
                    if (!pass) kj = INF;
                    const uint32_t mp = popc64(pm);
                    uint32_t sh[NCC], mypos = 0;
#pragma unroll
                    for (int c = 0; c < NCC; ++c) sh[c] = 0;
                    uint64_t it = pm;
                    while (it) {
                        const int l = __ffsll((unsigned long long)it) - 1; it &= it - 1;
                        const float k = readlane_f(kj, l);
                        uint32_t pos = 0;
#pragma unroll
                        for (int c = 0; c < NCC; ++c) {
                            const bool gt = rk[c] > k;                            // existing entries that move back by one
                            pos += WAVE - popc64(ballot(gt));
                            sh[c] += gt ? 1u : 0u;
                        }
                        const uint64_t lt = ballot(kj < k), eq = ballot(kj == k) & pm;
                        pos += popc64(lt | (eq & ((1ull << l) - 1ull)));          // candidates among themselves: list order
                        if (lane == l) mypos = pos;
                    }
#pragma unroll
                    for (int c = 0; c < NCC; ++c) {
                        const uint32_t dst = c * WAVE + lane + sh[c];
                        if (dst < NCC * WAVE) perm[dst] = make_uint2(__float_as_uint(rk[c]), rv[c]);
                    }
                    if (pass && mypos < NCC * WAVE) perm[mypos] = make_uint2(__float_as_uint(kj), sj);
                    n = (n + mp < cap) ? n + mp : cap;
                    wave_sync();
#pragma unroll
                    for (int c = 0; c < NCC; ++c) {
                        const uint2 e = perm[c * WAVE + lane];
                        rk[c] = __uint_as_float(e.x); rv[c] = e.y;
                    }
                    if (NCC > 4) {
                        // lists of six and eight chunks also serve ef values that end one chunk earlier
                        float wk = INF;
#pragma unroll
                        for (int c = NCC - 2; c < NCC; ++c) {
                            if ((uint32_t)(c * WAVE + lane) >= cap) { rk[c] = INF; rv[c] = FLAG_EXPANDED; }
                            if ((uint32_t)c == ((cap - 1) >> 6)) wk = rk[c];
                        }
                        wave_sync();
                        worst = (n >= cap) ? readlane_f(wk, (int)((cap - 1) & 63u)) : INF;
                    } else {
                    if ((uint32_t)((NCC - 1) * WAVE + lane) >= cap) { rk[NCC - 1] = INF; rv[NCC - 1] = FLAG_EXPANDED; }
                    wave_sync();
                    worst = (n >= cap) ? readlane_f(rk[NCC - 1], (int)((cap - 1) & 63u)) : INF;
                    }
                }
                VDB_T_MARK(4);
            }
        }
        // hand the list over in LDS (Euclidean: root of the squared keys)
#pragma unroll
        for (int c = 0; c < NCC; ++c) {
            const uint32_t i = c * WAVE + lane;
            if (i < n) { R.key[i] = (metric() == VDB_EUCLIDEAN) ? __builtin_sqrtf(rk[c]) : rk[c]; R.val[i] = rv[c]; }
        }
        R.n = n; R.cap = cap;
        if (MASKED) {
            if ((uint32_t)lane < pn) { P.key[lane] = (metric() == VDB_EUCLIDEAN) ? __builtin_sqrtf(pk) : pk; P.val[lane] = pv; }
            P.n = pn; P.cap = pcap;
        }
        wave_sync();
    }

    // Candidate queue of the literal filtered walk: std::collections::BinaryHeap<ScoredPoint> (hnsw.rs:638) in this
    // query's slice of HBM scratch, restated operation by operation so that equal keys leave it in the reference's
    // order.  ScoredPoint's Ord is inverted (hnsw.rs:25-39): a <= b in heap terms means a.sort_key >= b.sort_key, the
    // smallest key sits at the top.  One lane runs the dependent chain of loads; the slice belongs to this wavefront.
    static __device__ __forceinline__ void heap_sift_up(vdb_scored_point* C, uint32_t pos, const vdb_scored_point elt) {
        while (pos > 0) {                                                  // BinaryHeap::sift_up(0, pos)
            const uint32_t parent = (pos - 1) / 2;
            const vdb_scored_point pe = C[parent];
            if (elt.sort_key >= pe.sort_key) break;                        // hole.element() <= hole.get(parent)
            C[pos] = pe; pos = parent;
        }
        C[pos] = elt;
    }
    static __device__ __forceinline__ void heap_push(vdb_scored_point* C, uint32_t len, const vdb_scored_point elt) { heap_sift_up(C, len, elt); }
    // BinaryHeap::pop: the last element takes the root's place, sinks to the bottom along the smaller-key children
    // (equal keys: the right child) and rises again (sift_down_to_bottom + sift_up).  len = entries before the pop.
    static __device__ __forceinline__ vdb_scored_point heap_pop(vdb_scored_point* C, uint32_t len) {
        const vdb_scored_point last = C[len - 1];
        const uint32_t end = len - 1;
        if (end == 0) return last;
        const vdb_scored_point top = C[0];
        uint32_t pos = 0, child = 1;
        const uint32_t lim = end >= 2 ? end - 2 : 0;
        while (child <= lim) {
            vdb_scored_point ce = C[child];
            const vdb_scored_point re = C[child + 1];
            if (ce.sort_key >= re.sort_key) { ++child; ce = re; }          // hole.get(child) <= hole.get(child + 1)
            C[pos] = ce; pos = child; child = 2 * pos + 1;
        }
        if (child == end - 1) { C[pos] = C[child]; pos = child; }
        heap_sift_up(C, pos, last);
        return top;
    }

    // hnsw.rs:585-689 as written.  C is the candidate heap above; R (inverted order in the reference) keeps the ef
    // LARGEST keys among entry + passing nodes and `bound` is its smallest key; output is descending by key.
    __device__ __forceinline__ void literal_filtered(uint32_t q, uint32_t entry) {
        const int lane = lane_id();
        vdb_scored_point* C = a.cand_scratch + (size_t)q * a.cand_cap;
        uint32_t cn = 0;
        reset_visited();
        R.n = 0; R.cap = a.ef;           // stored ascending by NEGATED key => position 0 is the largest key
        wave_sync();
        const float raw0 = score_one(entry);
        const float key0 = key_of(raw0);
        if (lane == 0) { vdb_scored_point e; e.id = entry; e.raw_score = raw0; e.sort_key = key0; heap_push(C, 0, e); vis.insert(entry); }
        cn = 1; note_visited(1);
        R.insert(-key0, entry);                                            // hnsw.rs:641-644: entry admitted unfiltered
        float bound = key0;                                                // hnsw.rs:646
        wave_sync();
        for (;;) {
            if (cn == 0) break;
            // pop the minimum of C (hnsw.rs:648)
            uint32_t cur = 0; float bk = 0.f;
            if (lane == 0) { const vdb_scored_point t = heap_pop(C, cn); cur = (uint32_t)t.id; bk = t.sort_key; }
            cur = readlane_u(cur, 0); bk = readlane_f(bk, 0);
            --cn;
            if (bk > bound) break;                                         // hnsw.rs:649-651
            const Adj A = open_adj(gv, 0, cur);
            if (A.deg == 0) continue;
            ++expanded; adj_entries += A.deg;
            const uint32_t nchunks = A.chunks();
            for (uint32_t c = 0; c < nchunks; ++c) {
                uint32_t nb;
                const bool in = A.get(c, nb);
                bool fresh = in && !is_deleted(nb);
                if (fresh) fresh = vis.insert(nb);
                const uint64_t mk = ballot(fresh);
                const int m = popc64(mk);
                if (m == 0) continue;
                note_visited(m);
                if (fresh) cand[popc64(mk & lanemask_lt(lane))] = nb;
                wave_sync();
                score(m);
                if (cn + m > a.cand_cap) { flags |= QF_CAND_OVERFLOW; return; }
                const float rj = (lane < m) ? craw[lane] : 0.f;
                const float kj = key_of(rj);
                const uint32_t sj = (lane < m) ? cand[lane] : 0u;
                if (lane == 0) {                                           // hnsw.rs:666: every scored neighbour, in list order
                    for (int j = 0; j < m; ++j) { vdb_scored_point e; e.id = cand[j]; e.raw_score = craw[j]; e.sort_key = key_of(craw[j]); heap_push(C, cn + j, e); }
                }
                cn += m;
                uint64_t pm = ballot(lane < m && a.pass_bits && bit_test(a.pass_bits, sj));
                if (!a.pass_bits) pm = ballot(lane < m);
                while (pm) {                                               // hnsw.rs:673-681, in list order
                    const int j = __ffsll((unsigned long long)pm) - 1; pm &= pm - 1;
                    const float k = readlane_f(kj, j);
                    // push, then drop the SMALLEST key when over ef (inverted heap)
                    // (the list is ascending in -key, so its last element holds the smallest key)
                    if (R.n >= R.cap && !(-k < R.key[R.cap - 1])) {
                        bound = -R.key[R.cap - 1];                          // pushed and popped again at once
                        continue;
                    }
                    R.insert(-k, readlane_u(sj, j));
                    bound = -R.key[R.n - 1];                                // peek(): smallest key kept
                }
                wave_sync();
                if (!visited_room()) return;
            }
        }
    }

    __device__ __forceinline__ void emit(uint32_t q, const SortedList& L, bool negated) {
        const int lane = lane_id();
        const uint32_t n = L.n < a.top_k ? L.n : a.top_k;
        for (uint32_t i = lane; i < n; i += WAVE) {
            const uint32_t s = L.val[i] & ~FLAG_EXPANDED;
            const float k = negated ? -L.key[i] : L.key[i];
            vdb_scored_point sp; sp.id = gv.ids[s]; sp.sort_key = k; sp.raw_score = (metric() == VDB_DOT) ? -k : k;
            a.out[(size_t)q * a.top_k + i] = sp;
        }
        if (lane == 0) a.out_n[q] = n;
    }

    __device__ __forceinline__ void run(uint32_t q, char* smem) {
        const int lane = lane_id();
        // ---- carve LDS (every offset a multiple of 16 bytes) ----
        const uint32_t ef_cap = (a.ef + 63u) & ~63u;
        size_t off = 0;
        if (NC > 0 && (MODE == MODE_SEARCH || MODE == MODE_MASKED)) {
            // register-resident list: LDS only stages permutations, the query (until it is in
            // registers) and the final list, all in one area
            // (candidate ids + scores, then the permutation of the merge, which has read them into registers)
            size_t area = (size_t)CAND_CAP * 8;
            if ((size_t)NCC * WAVE * 8 > area) area = (size_t)NCC * WAVE * 8;
            if ((size_t)gv.stride * 4 > area) area = (size_t)gv.stride * 4;
            if (MODE == MODE_MASKED) {                     // the passing list is handed over behind the result list
                if ((size_t)(NCC + 1) * WAVE * 8 > area) area = (size_t)(NCC + 1) * WAVE * 8;
                P.key = reinterpret_cast<float*>(smem + (size_t)NCC * WAVE * 8); P.val = reinterpret_cast<uint32_t*>(P.key + WAVE);
                P.key2 = nullptr; P.val2 = nullptr;
            }
            perm = reinterpret_cast<uint2*>(smem);
            qlds = reinterpret_cast<float*>(smem);
            R.key = reinterpret_cast<float*>(smem); R.val = reinterpret_cast<uint32_t*>(smem) + NCC * WAVE;
            R.key2 = nullptr; R.val2 = nullptr;
            cand = reinterpret_cast<uint32_t*>(smem);
            craw = reinterpret_cast<float*>(smem + CAND_CAP * 4);
            off = (area + 15) & ~(size_t)15;
            tags.tab = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)2 << a.tag_log2;
        } else {
            R.key = reinterpret_cast<float*>(smem + off); off += ef_cap * 4;
            R.val = reinterpret_cast<uint32_t*>(smem + off); off += ef_cap * 4;
            R.key2 = reinterpret_cast<float*>(smem + off); off += ef_cap * 4;
            R.val2 = reinterpret_cast<uint32_t*>(smem + off); off += ef_cap * 4;
            if (MODE == MODE_MASKED) {
                P.key = reinterpret_cast<float*>(smem + off); off += ef_cap * 4;
                P.val = reinterpret_cast<uint32_t*>(smem + off); off += ef_cap * 4;
                P.key2 = reinterpret_cast<float*>(smem + off); off += ef_cap * 4;
                P.val2 = reinterpret_cast<uint32_t*>(smem + off); off += ef_cap * 4;
            }
            if (NC > 0) {
                perm = reinterpret_cast<uint2*>(smem + off); off += (size_t)NCC * WAVE * 8;
                tags.tab = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)2 << a.tag_log2;
            }
            cand = reinterpret_cast<uint32_t*>(smem + off); off += CAND_CAP * 4;
            craw = reinterpret_cast<float*>(smem + off); off += CAND_CAP * 4;
            qlds = reinterpret_cast<float*>(smem + off); off += (size_t)gv.stride * 4;
        }
        if (NC > 0) tags.idmask = (1u << (a.tag_log2 + 13)) - 1;
        vis.mask = (1u << (a.hash_log2 - 2)) - 1; vis.shift = 32 - (a.hash_log2 - 2);
        vis.limit = (1u << a.hash_log2) - (1u << a.hash_log2) / 4 - (NC > 0 ? 4 * WAVE : WAVE);   // room for one more step
        vis.tab = reinterpret_cast<uint32_t*>(smem + off);
        vis.spill = nullptr; vis.smask = 0; vis.sshift = 0; vis.scount = 0; vis.slimit = 0; vis.count = 0;

        if (lane == 0) a.out_n[q] = 0;
        if (a.entry == NONE32) return;                                     // hnsw.rs:515-518
#ifdef VDB_TIMING
        const unsigned long long t_begin = wall_clock64();
#endif

        // ---- query -> LDS (normalised for Cosine: hnsw.rs:526-536, 752-764) ----
        const float* qsrc = (MODE == MODE_BUILD) ? gv.vecs + (size_t)a.query_slots[q] * gv.stride : a.queries + (size_t)q * gv.dim;
        for (uint32_t i = lane; i < gv.stride; i += WAVE) qlds[i] = (i < gv.dim) ? qsrc[i] : 0.f;
        wave_sync();
        if (metric() == VDB_COSINE && MODE != MODE_BUILD && MODE != MODE_LITERAL) {
            float s = -0.0f;                                               // every lane runs the same sequential sum
            for (uint32_t i = 0; i < gv.dim; ++i) { float p = qlds[i] * qlds[i]; s = s + p; }
            const float norm = __builtin_sqrtf(s);
            wave_sync();
            if (norm != 0.0f) for (uint32_t i = lane; i < gv.dim; i += WAVE) qlds[i] = qlds[i] / norm;
            wave_sync();
        }
        if (!STRICT) { qr.load(qlds, gv.stride, metric()); wave_sync(); }

        if (MODE == MODE_LITERAL) {
            uint32_t cur = a.q_entry ? a.q_entry[q] : a.entry;
            for (uint32_t l = a.max_level; l >= 1 && !(flags & QF_FILTER_ERROR); --l) cur = greedy_filtered(cur, l);   // hnsw.rs:628-630
            if (!(flags & QF_FILTER_ERROR)) literal_filtered(q, cur);
            if (!(flags & (QF_HASH_OVERFLOW | QF_CAND_OVERFLOW | QF_FILTER_ERROR))) emit(q, R, true);
        } else if (MODE == MODE_BUILD) {
            const uint32_t lvl = a.q_levels[q];
            uint32_t cur = a.entry;
            if (gv.ublk) {
                uint32_t ref = make_uref(gv.upper_base[cur], gv.upper_cnt[cur]); float kc = 0.f; bool hc = false;
                for (uint32_t l = a.max_level; l > lvl; --l) cur = greedy_fused(cur, ref, l, kc, hc);
            } else
            for (uint32_t l = a.max_level; l > lvl; --l) cur = greedy(cur, l);          // hnsw.rs:169-173
            for (int l = (int)lvl; l >= 0; --l) {                                       // hnsw.rs:175-199
                if (l == 0) { if (NC > 0) beam_reg<false>(cur, a.ef, a.beam_width); else beam<false, true>(cur, 0, a.ef); }
                else beam<false, false>(cur, (uint32_t)l, a.ef);
                if (flags & QF_HASH_OVERFLOW) break;
                const uint32_t cnt = R.n < a.m ? R.n : a.m;
                const size_t row = (size_t)a.nb_off[q] + (uint32_t)l;
                for (uint32_t i = lane; i < cnt; i += WAVE) a.nb_out[row * a.m + i] = R.val[i] & ~FLAG_EXPANDED;
                if (lane == 0) a.nb_cnt[row] = cnt;
                if (R.n > 0) cur = R.val[0] & ~FLAG_EXPANDED;
                wave_sync();
            }
            if (a.out && !(flags & QF_HASH_OVERFLOW)) emit(q, R, false);                // layer-0 beam, full list
        } else {
            uint32_t cur = a.entry;
            if (gv.ublk) {
                uint32_t ref = make_uref(gv.upper_base[cur], gv.upper_cnt[cur]); float kc = 0.f; bool hc = false;
                for (uint32_t l = a.max_level; l >= 1; --l) cur = greedy_fused(cur, ref, l, kc, hc);
            } else
            for (uint32_t l = a.max_level; l >= 1; --l) cur = greedy(cur, l);           // hnsw.rs:538-541
            if (MODE == MODE_MASKED) {
                if (NC > 0) beam_reg<true>(cur, a.ef, a.beam_width); else beam<true, true>(cur, 0, a.ef);
                if (!(flags & QF_HASH_OVERFLOW)) emit(q, P, false);
            }
            else {
                if (NC > 0) beam_reg<false>(cur, a.ef, a.beam_width); else beam<false, true>(cur, 0, a.ef);
                if (!(flags & QF_HASH_OVERFLOW)) emit(q, R, false);
            }
        }
        if (spill_idx != NONE32) {                                          // hand the spill table back
            __threadfence();
            if (lane == 0) atomicExch(&a.spill_locks[spill_idx], 0u);
        }
#ifdef VDB_TIMING
        // diagnostic build only: adj_entries carries the wavefront's lifetime in 10 ns ticks, rows_scored its start tick
        adj_entries = (uint32_t)(wall_clock64() - t_begin); rows_scored = (uint32_t)t_begin;
        if (lane == 0 && a.out && a.top_k >= 5) {      // phase totals overwrite this query's result records
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.out + (size_t)q * a.top_k);
            tacc[9] -= t_begin; tacc[5] = wall_clock64() - t_begin;
            for (int i = 0; i < 10; ++i) dbg[i] = tacc[i];
        }
#endif
        // on a filter error `expanded` carries the slot whose evaluation failed
        if (a.stats && lane == 0) a.stats[q] = {rows_scored, (flags & QF_FILTER_ERROR) ? err_slot : expanded, adj_entries, flags};
    }
};

template <int MODE, int G, int CH, int PASSES, bool STRICT, int METRIC, int NC>
__global__ void __launch_bounds__(64) search_kernel(SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Searcher<MODE, G, CH, PASSES, STRICT, METRIC, NC> s(a);
    s.run(blockIdx.x, smem);
}

// FAST shapes: rows up to 128 floats use 8 lanes x 4 chunks (8 rows per wave instruction), up to
// 1024 floats the whole wave per row x 4 chunks, up to 2048 floats x 8 chunks.  NC > 0 selects the
// register-resident layer-0 list of NC x 64 entries (reg_list_chunks()).
template <int MODE, int METRIC, int NC>
void launch_search_shape(const SearchArgs& a, size_t lds, hipStream_t st) {
    const uint32_t chunks = a.g.stride / 4;
    constexpr int P8 = NC > 0 ? 2 : 4;             // row groups in flight per lane for short rows
    constexpr int P64 = NC > 0 ? 4 : 8;            // rows in flight per wave for long rows (x 2 chunks: half of it)
    // A launch of at most one wavefront per SIMD has no other wavefront to hide a row fetch behind: its walks keep
    // four times as many rows in flight per wavefront (more registers, the same sums in the same order).
    constexpr bool HAS_FEW = NC > 0 && (MODE == MODE_SEARCH || MODE == MODE_BUILD);
    const bool few = HAS_FEW && a.nq <= 1024;
    if (chunks <= 32) {
        if (few) hipLaunchKernelGGL((search_kernel<MODE, 8, 4, HAS_FEW ? 8 : P8, false, METRIC, NC>), dim3(a.nq), dim3(64), lds, st, a);
        else hipLaunchKernelGGL((search_kernel<MODE, 8, 4, P8, false, METRIC, NC>), dim3(a.nq), dim3(64), lds, st, a);
    } else if (chunks <= 256) {
        if (few) hipLaunchKernelGGL((search_kernel<MODE, 64, 4, HAS_FEW ? 16 : P64, false, METRIC, NC>), dim3(a.nq), dim3(64), lds, st, a);
        else hipLaunchKernelGGL((search_kernel<MODE, 64, 4, P64, false, METRIC, NC>), dim3(a.nq), dim3(64), lds, st, a);
    } else if (chunks <= 512) {
        if (few) hipLaunchKernelGGL((search_kernel<MODE, 64, 8, HAS_FEW ? 8 : P64 / 2, false, METRIC, NC>), dim3(a.nq), dim3(64), lds, st, a);
        else hipLaunchKernelGGL((search_kernel<MODE, 64, 8, P64 / 2, false, METRIC, NC>), dim3(a.nq), dim3(64), lds, st, a);
    }
    else throw Failure(err_arg("dimension above 2048 is not supported"));
}
template <int MODE, int METRIC>
void launch_search_fast(const SearchArgs& a, size_t lds, hipStream_t st) {
    if (MODE == MODE_SEARCH || MODE == MODE_BUILD || MODE == MODE_MASKED) {
        switch (reg_list_chunks(a)) {
            case 1: launch_search_shape<MODE, METRIC, 1>(a, lds, st); return;
            case 2: launch_search_shape<MODE, METRIC, 2>(a, lds, st); return;
            case 3: launch_search_shape<MODE, METRIC, 3>(a, lds, st); return;
            case 4: launch_search_shape<MODE, METRIC, 4>(a, lds, st); return;
            case 6: if constexpr (MODE == MODE_SEARCH) { launch_search_shape<MODE, METRIC, 6>(a, lds, st); return; } break;
            case 8: if constexpr (MODE == MODE_SEARCH) { launch_search_shape<MODE, METRIC, 8>(a, lds, st); return; } break;
            default: break;
        }
    }
    launch_search_shape<MODE, METRIC, 0>(a, lds, st);
}
template <int MODE>
void launch_search_strict(const SearchArgs& a, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((search_kernel<MODE, 0, 0, 0, true, -1, 0>), dim3(a.nq), dim3(64), lds, st, a);
}

}  // namespace
}  // namespace vdb
