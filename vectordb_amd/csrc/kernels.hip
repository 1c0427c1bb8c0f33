// gfx950 (MI355X) kernels of libvdbmi.  Wave64 throughout; every search block is ONE wavefront that
// owns one query, so no block-level barrier is ever a real s_barrier.
//
// This file is compiled with -ffp-contract=off: the STRICT paths rely on separate multiply and
// add roundings (metric.rs:15-35 never fuses); the FAST paths ask for FMAs explicitly.
#include "kernels.hpp"

#include <algorithm>

namespace vdb {

namespace {

constexpr int WAVE = 64;
constexpr uint32_t FLAG_EXPANDED = 0x80000000u;
constexpr uint32_t HASH_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int popc64(uint64_t m) { return __popcll(m); }
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (lane == 0) ? 0ull : (~0ull >> (64 - lane)); }
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
// LDS traffic of one wave is processed in issue order; this keeps the compiler from moving LDS
// accesses across the point and waits for outstanding ones.
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// Sum over aligned groups of G lanes; every lane of the group receives the sum.
template <int G> __device__ __forceinline__ float group_sum(float v) {
    v += dpp_f<0xB1>(v);                 // quad_perm [1,0,3,2]  (xor 1)
    v += dpp_f<0x4E>(v);                 // quad_perm [2,3,0,1]  (xor 2)
    if (G >= 8) v += dpp_f<0x141>(v);    // row_half_mirror      (joins the two quads of 8 lanes)
    if (G >= 16) v += dpp_f<0x140>(v);   // row_mirror           (joins the two halves of 16 lanes)
    if (G >= 32) {                       // rows of 16 lanes now hold their own sums
        float r0 = readlane_f(v, 0), r1 = readlane_f(v, 16), r2 = readlane_f(v, 32), r3 = readlane_f(v, 48);
        if (G == 64) v = (r0 + r1) + (r2 + r3);
        else v = (lane_id() < 32) ? (r0 + r1) : (r2 + r3);
    }
    return v;
}

__device__ __forceinline__ bool bit_test(const uint32_t* bits, uint32_t i) { return (bits[i >> 5] >> (i & 31)) & 1u; }

struct ListRef { const uint32_t* p; uint32_t n; };
__device__ __forceinline__ ListRef get_list(const GraphView& g, uint32_t level, uint32_t slot) {
    uint64_t d;
    if (level == 0) d = g.desc0[slot];
    else {
        uint32_t base = g.upper_base[slot];
        if (base == NONE32 || level > g.upper_cnt[slot]) return {nullptr, 0};
        d = g.udesc[base + level - 1];
    }
    return {g.pool + (d & 0xFFFFFFFFFFull), (uint32_t)(d >> 40)};
}

// ---------------------------------------------------------------------------------------------
// Distance arithmetic
// ---------------------------------------------------------------------------------------------

// STRICT: metric.rs:15-35 literally — one lane, one row, left to right, from -0.0.
__device__ float strict_pair(const float* __restrict__ a, const float* __restrict__ b, uint32_t dim, int metric, bool a_is_lds) {
    float s = -0.0f, sa = -0.0f, sb = -0.0f;
    uint32_t i = 0;
    for (; i + 4 <= dim; i += 4) {
        float4 x = *reinterpret_cast<const float4*>(a + i);
        float4 y = *reinterpret_cast<const float4*>(b + i);
        float xa[4] = {x.x, x.y, x.z, x.w}, ya[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (metric == VDB_EUCLIDEAN) { float d = xa[k] - ya[k]; float p = d * d; s = s + p; }
            else {
                float p = xa[k] * ya[k]; s = s + p;
                if (metric == VDB_COSINE) { float pa = xa[k] * xa[k]; sa = sa + pa; float pb = ya[k] * ya[k]; sb = sb + pb; }
            }
        }
    }
    for (; i < dim; ++i) {
        float x = a[i], y = b[i];
        if (metric == VDB_EUCLIDEAN) { float d = x - y; float p = d * d; s = s + p; }
        else {
            float p = x * y; s = s + p;
            if (metric == VDB_COSINE) { float pa = x * x; sa = sa + pa; float pb = y * y; sb = sb + pb; }
        }
    }
    if (metric == VDB_EUCLIDEAN) return __builtin_sqrtf(s);
    if (metric == VDB_DOT) return s;
    float den = __builtin_sqrtf(sa) * __builtin_sqrtf(sb);
    den = den + 1e-10f;
    float q = s / den;
    return 1.0f - q;
}

// FAST: G lanes share a row; lane g of the group owns 16-byte chunks g, g+G, g+2G, ...
template <int G, int CH> struct QueryRegs {
    float4 q[CH];
    float qnorm;     // sqrt(sum q^2) for Cosine
    __device__ void load(const float* qlds, uint32_t stride, int metric) {
        const int g = lane_id() % G;
        const uint32_t nchunks = stride >> 2;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            uint32_t c = g + G * j;
            q[j] = (c < nchunks) ? reinterpret_cast<const float4*>(qlds)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            acc = __builtin_fmaf(q[j].x, q[j].x, acc); acc = __builtin_fmaf(q[j].y, q[j].y, acc);
            acc = __builtin_fmaf(q[j].z, q[j].z, acc); acc = __builtin_fmaf(q[j].w, q[j].w, acc);
        }
        qnorm = (metric == VDB_COSINE) ? __builtin_sqrtf(group_sum<G>(acc)) : 0.f;
    }
};

// Scores cand[0..m) (LDS) into craw[0..m) (LDS).
template <int G, int CH, int PASSES>
__device__ void score_cands_fast(const GraphView& gv, const QueryRegs<G, CH>& qr, const uint32_t* cand, int m, float* craw) {
    constexpr int RPP = WAVE / G;
    const int lane = lane_id(), g = lane % G, r = lane / G;
    const uint32_t nchunks = gv.stride >> 2;
    for (int base = 0; base < m; base += RPP * PASSES) {
        float4 x[PASSES][CH];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int ci = base + p * RPP + r;
            const bool live = ci < m;
            const uint32_t s = live ? cand[ci] : 0u;
            const float4* row = reinterpret_cast<const float4*>(gv.vecs + (size_t)s * gv.stride);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const uint32_t c = g + G * j;
                x[p][j] = (live && c < nchunks) ? row[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int ci = base + p * RPP + r;
            float acc = 0.f, accb = 0.f;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const float4 q = qr.q[j], v = x[p][j];
                if (gv.metric == VDB_EUCLIDEAN) {
                    float d0 = q.x - v.x, d1 = q.y - v.y, d2 = q.z - v.z, d3 = q.w - v.w;
                    acc = __builtin_fmaf(d0, d0, acc); acc = __builtin_fmaf(d1, d1, acc);
                    acc = __builtin_fmaf(d2, d2, acc); acc = __builtin_fmaf(d3, d3, acc);
                } else {
                    acc = __builtin_fmaf(q.x, v.x, acc); acc = __builtin_fmaf(q.y, v.y, acc);
                    acc = __builtin_fmaf(q.z, v.z, acc); acc = __builtin_fmaf(q.w, v.w, acc);
                    if (gv.metric == VDB_COSINE) {
                        accb = __builtin_fmaf(v.x, v.x, accb); accb = __builtin_fmaf(v.y, v.y, accb);
                        accb = __builtin_fmaf(v.z, v.z, accb); accb = __builtin_fmaf(v.w, v.w, accb);
                    }
                }
            }
            acc = group_sum<G>(acc);
            float raw;
            if (gv.metric == VDB_EUCLIDEAN) raw = __builtin_sqrtf(acc);
            else if (gv.metric == VDB_DOT) raw = acc;
            else {
                accb = group_sum<G>(accb);
                raw = 1.0f - acc / (qr.qnorm * __builtin_sqrtf(accb) + 1e-10f);
            }
            if (g == 0 && ci < m) craw[ci] = raw;
        }
    }
}

__device__ void score_cands_strict(const GraphView& gv, const float* qlds, const uint32_t* cand, int m, float* craw) {
    const int lane = lane_id();
    if (lane < m) craw[lane] = strict_pair(qlds, gv.vecs + (size_t)cand[lane] * gv.stride, gv.dim, gv.metric, true);
}

// ---------------------------------------------------------------------------------------------
// Wave-cooperative sorted list in LDS (ascending key, arrival order among equal keys)
// ---------------------------------------------------------------------------------------------
struct SortedList {
    float* key; uint32_t* val; uint32_t n, cap;
    __device__ float worst() const { return key[cap - 1]; }
    // Caller guarantees n < cap or k < key[n-1].  All lanes call with the same (k, v).
    __device__ void insert(float k, uint32_t v) {
        const int lane = lane_id();
        uint32_t pos = 0;
        for (uint32_t b = 0; b < n; b += WAVE) {
            const uint32_t i = b + lane;
            pos += popc64(ballot(i < n && key[i] <= k));
        }
        const uint32_t last = (n < cap) ? n : cap - 1;       // highest index written by the shift
        for (int b = (int)(last & ~63u); b >= (int)(pos & ~63u); b -= WAVE) {
            const uint32_t i = (uint32_t)b + lane;
            const bool mv = i > pos && i <= last;
            float tk = 0.f; uint32_t tv = 0;
            if (mv) { tk = key[i - 1]; tv = val[i - 1]; }
            wave_sync();
            if (mv) { key[i] = tk; val[i] = tv; }
            wave_sync();
        }
        if (lane == 0) { key[pos] = k; val[pos] = v; }
        if (n < cap) ++n;
        wave_sync();
    }
};

// Visited set: open addressing over slot ids, linear probing.
struct Visited {
    uint32_t* tab; uint32_t mask, shift, count, limit;
    template <bool GLOBAL> __device__ bool insert(uint32_t s) {
        uint32_t h = (s * 2654435761u) >> shift;
        for (;;) {
            uint32_t old = atomicCAS(&tab[h], HASH_EMPTY, s);
            if (old == HASH_EMPTY) return true;
            if (old == s) return false;
            h = (h + 1) & mask;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// HNSW search: greedy descent over the upper layers (hnsw.rs:333-368) then the ef-beam on one
// layer (hnsw.rs:420-511).  One wavefront per query.
// ---------------------------------------------------------------------------------------------
template <int G, int CH, int PASSES, bool STRICT, bool GHASH>
struct Searcher {
    const SearchArgs& a;
    const GraphView& gv;
    float* qlds; uint32_t* cand; float* craw;
    SortedList R, P;                      // R: beam / result list, P: filter-passing list (MODE_MASKED)
    Visited vis;
    QueryRegs<(G > 0 ? G : 8), (CH > 0 ? CH : 1)> qr;
    uint32_t rows_scored = 0, expanded = 0, adj_entries = 0, flags = 0, err_slot = 0;

    __device__ Searcher(const SearchArgs& args) : a(args), gv(args.g) {}

    __device__ float key_of(float raw) const { return gv.metric == VDB_DOT ? -raw : raw; }    // hnsw.rs:103-108
    __device__ bool is_deleted(uint32_t s) const { return gv.deleted && bit_test(gv.deleted, s); }

    __device__ void score(int m) {
        if (STRICT) score_cands_strict(gv, qlds, cand, m, craw);
        else score_cands_fast<(G > 0 ? G : 8), (CH > 0 ? CH : 1), (PASSES > 0 ? PASSES : 1)>(gv, qr, cand, m, craw);
        wave_sync();
        rows_scored += m;
    }
    __device__ float score_one(uint32_t s) {
        if (lane_id() == 0) cand[0] = s;
        wave_sync();
        score(1);
        return craw[0];
    }

    // hnsw.rs:333-368: move to the FIRST neighbour that improves, rescan, at most 1000 steps.
    __device__ uint32_t greedy(uint32_t entry, uint32_t level) {
        const int lane = lane_id();
        uint32_t cur = entry;
        float kcur = 0.f; bool have_cur = false;
        bool changed = true; int steps = 0;
        while (changed && steps < 1000) {
            ++steps; changed = false;
            ListRef L = get_list(gv, level, cur);
            if (L.n == 0) continue;
            ++expanded; adj_entries += L.n;
            for (uint32_t b = 0; b < L.n && !changed; b += WAVE) {
                const uint32_t i = b + lane;
                uint32_t nb = (i < L.n) ? L.p[i] : 0u;
                const bool ok = (i < L.n) && !is_deleted(nb);
                const uint64_t mk = ballot(ok);
                const int m = popc64(mk);
                if (m == 0) continue;
                if (!have_cur) { kcur = key_of(score_one(cur)); have_cur = true; }
                if (ok) cand[popc64(mk & lanemask_lt(lane))] = nb;
                wave_sync();
                score(m);
                const bool better = lane < m && key_of(craw[lane]) < kcur;
                const uint64_t bm = ballot(better);
                if (bm) {
                    const int j = __ffsll((unsigned long long)bm) - 1;
                    cur = cand[j]; kcur = key_of(craw[j]); changed = true;
                }
                wave_sync();
            }
        }
        return cur;
    }

    // hnsw.rs:370-418: scan the WHOLE list keeping the best filter-passing neighbour (strict <, so
    // the first of equal minima wins), repeat until a scan makes no move; no step cap.  A
    // neighbour whose filter evaluation is an Err aborts the search (hnsw.rs:392).
    __device__ uint32_t greedy_filtered(uint32_t entry, uint32_t level) {
        const int lane = lane_id();
        uint32_t cur = entry;
        float kcur = 0.f; bool have_cur = false;
        bool changed = true;
        while (changed) {
            changed = false;
            ListRef L = get_list(gv, level, cur);
            if (L.n == 0) break;
            ++expanded; adj_entries += L.n;
            uint32_t best = cur;
            for (uint32_t b = 0; b < L.n; b += WAVE) {
                const uint32_t i = b + lane;
                const uint32_t nb = (i < L.n) ? L.p[i] : 0u;
                bool ok = (i < L.n) && !is_deleted(nb);
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
    // MASKED additionally keeps the ef best filter-passing nodes in P.
    template <bool MASKED> __device__ void beam(uint32_t entry, uint32_t level, uint32_t ef) {
        const int lane = lane_id();
        // reset visited + lists
        for (uint32_t i = lane; i <= vis.mask; i += WAVE) vis.tab[i] = HASH_EMPTY;
        vis.count = 0;
        R.n = 0; R.cap = ef;
        if (MASKED) { P.n = 0; P.cap = ef; }
        wave_sync();
        uint32_t start = entry;
        if (is_deleted(entry) && a.fallback_live != NONE32) start = a.fallback_live;      // hnsw.rs:440-444
        const float raw0 = score_one(start);
        R.insert(key_of(raw0), start);
        if (lane == 0) vis.template insert<GHASH>(start);
        vis.count = 1;
        if (MASKED && !is_deleted(start) && (!a.pass_bits || bit_test(a.pass_bits, start))) P.insert(key_of(raw0), start);
        wave_sync();
        for (;;) {
            // best unexpanded entry of R == the candidate queue's minimum (evicted candidates can
            // only re-surface on an exact key tie)
            int sel = -1;
            for (uint32_t b = 0; b < R.n; b += WAVE) {
                const uint32_t i = b + lane;
                const uint64_t um = ballot(i < R.n && !(R.val[i] & FLAG_EXPANDED));
                if (um) { sel = (int)b + __ffsll((unsigned long long)um) - 1; break; }
            }
            if (sel < 0) break;
            const uint32_t cur = R.val[sel];
            wave_sync();
            if (lane == 0) R.val[sel] = cur | FLAG_EXPANDED;
            wave_sync();
            ListRef L = get_list(gv, level, cur);
            if (L.n == 0) continue;
            ++expanded; adj_entries += L.n;
            for (uint32_t b = 0; b < L.n; b += WAVE) {
                const uint32_t i = b + lane;
                const uint32_t nb = (i < L.n) ? L.p[i] : 0u;
                bool fresh = (i < L.n) && !is_deleted(nb);                    // hnsw.rs:475
                if (fresh) fresh = vis.template insert<GHASH>(nb);
                const uint64_t mk = ballot(fresh);
                const int m = popc64(mk);
                if (m == 0) continue;
                vis.count += m;
                if (fresh) cand[popc64(mk & lanemask_lt(lane))] = nb;
                wave_sync();
                score(m);
                const float kj = (lane < m) ? key_of(craw[lane]) : 0.f;
                const uint32_t sj = (lane < m) ? cand[lane] : 0u;
                if (MASKED) {
                    const bool pj = lane < m && (!a.pass_bits || bit_test(a.pass_bits, sj));
                    uint64_t pm = ballot(pj && (P.n < P.cap || kj < P.worst()));
                    while (pm) {
                        const int j = __ffsll((unsigned long long)pm) - 1; pm &= pm - 1;
                        const float k = readlane_f(kj, j);
                        if (P.n >= P.cap && !(k < P.worst())) continue;
                        P.insert(k, readlane_u(sj, j));
                    }
                }
                // hnsw.rs:486: accepted iff the list is not full or key < worst, in list order
                uint64_t am = ballot(lane < m && (R.n < R.cap || kj < R.worst()));
                while (am) {
                    const int j = __ffsll((unsigned long long)am) - 1; am &= am - 1;
                    const float k = readlane_f(kj, j);
                    if (R.n >= R.cap && !(k < R.worst())) continue;
                    R.insert(k, readlane_u(sj, j));
                }
                wave_sync();
                if (vis.count > vis.limit) { flags |= QF_HASH_OVERFLOW; return; }
            }
        }
    }

    // hnsw.rs:585-689 as written.  C is an unsorted array in global scratch; R (inverted order in
    // the reference) keeps the ef LARGEST keys among entry + passing nodes and `bound` is its
    // smallest key; output is descending by key.
    __device__ void literal_filtered(uint32_t q, uint32_t entry) {
        const int lane = lane_id();
        vdb_scored_point* C = a.cand_scratch + (size_t)q * a.cand_cap;
        uint32_t cn = 0;
        for (uint32_t i = lane; i <= vis.mask; i += WAVE) vis.tab[i] = HASH_EMPTY;
        vis.count = 0;
        R.n = 0; R.cap = a.ef;           // stored ascending by NEGATED key => position 0 is the largest key
        wave_sync();
        const float raw0 = score_one(entry);
        const float key0 = key_of(raw0);
        if (lane == 0) { C[0].id = entry; C[0].raw_score = raw0; C[0].sort_key = key0; vis.template insert<GHASH>(entry); }
        cn = 1; vis.count = 1;
        R.insert(-key0, entry);                                            // hnsw.rs:641-644: entry admitted unfiltered
        float bound = key0;                                                // hnsw.rs:646
        __threadfence_block();
        wave_sync();
        for (;;) {
            if (cn == 0) break;
            // pop the minimum of C (hnsw.rs:648); ties: lowest position
            float bk = 0.f; uint32_t bi = NONE32;
            for (uint32_t b = 0; b < cn; b += WAVE) {
                const uint32_t i = b + lane;
                if (i < cn) { float k = C[i].sort_key; if (bi == NONE32 || k < bk) { bk = k; bi = i; } }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ok_ = __shfl_xor(bk, off); const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off);
                if (oi != NONE32 && (bi == NONE32 || ok_ < bk || (ok_ == bk && oi < bi))) { bk = ok_; bi = oi; }
            }
            const uint32_t cur = (uint32_t)C[bi].id;
            wave_sync();
            if (lane == 0) C[bi] = C[cn - 1];
            --cn;
            __threadfence_block();
            wave_sync();
            if (bk > bound) break;                                         // hnsw.rs:649-651
            ListRef L = get_list(gv, 0, cur);
            if (L.n == 0) continue;
            ++expanded; adj_entries += L.n;
            for (uint32_t b = 0; b < L.n; b += WAVE) {
                const uint32_t i = b + lane;
                const uint32_t nb = (i < L.n) ? L.p[i] : 0u;
                bool fresh = (i < L.n) && !is_deleted(nb);
                if (fresh) fresh = vis.template insert<GHASH>(nb);
                const uint64_t mk = ballot(fresh);
                const int m = popc64(mk);
                if (m == 0) continue;
                vis.count += m;
                if (fresh) cand[popc64(mk & lanemask_lt(lane))] = nb;
                wave_sync();
                score(m);
                if (cn + m > a.cand_cap) { flags |= QF_CAND_OVERFLOW; return; }
                const float rj = (lane < m) ? craw[lane] : 0.f;
                const float kj = key_of(rj);
                const uint32_t sj = (lane < m) ? cand[lane] : 0u;
                if (lane < m) { C[cn + lane].id = sj; C[cn + lane].raw_score = rj; C[cn + lane].sort_key = kj; }   // hnsw.rs:666
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
                __threadfence_block();
                wave_sync();
                if (vis.count > vis.limit) { flags |= QF_HASH_OVERFLOW; return; }
            }
        }
    }

    __device__ void emit(uint32_t q, const SortedList& L, bool negated) {
        const int lane = lane_id();
        const uint32_t n = L.n < a.top_k ? L.n : a.top_k;
        for (uint32_t i = lane; i < n; i += WAVE) {
            const uint32_t s = L.val[i] & ~FLAG_EXPANDED;
            const float k = negated ? -L.key[i] : L.key[i];
            vdb_scored_point sp; sp.id = gv.ids[s]; sp.sort_key = k; sp.raw_score = (gv.metric == VDB_DOT) ? -k : k;
            a.out[(size_t)q * a.top_k + i] = sp;
        }
        if (lane == 0) a.out_n[q] = n;
    }

    __device__ void run(uint32_t q, char* smem) {
        const int lane = lane_id();
        // ---- carve LDS (every offset a multiple of 16 bytes) ----
        const uint32_t ef_cap = (a.ef + 63u) & ~63u;
        size_t off = 0;
        R.key = reinterpret_cast<float*>(smem + off); off += ef_cap * 4;
        R.val = reinterpret_cast<uint32_t*>(smem + off); off += ef_cap * 4;
        if (a.mode == MODE_MASKED) {
            P.key = reinterpret_cast<float*>(smem + off); off += ef_cap * 4;
            P.val = reinterpret_cast<uint32_t*>(smem + off); off += ef_cap * 4;
        }
        cand = reinterpret_cast<uint32_t*>(smem + off); off += WAVE * 4;
        craw = reinterpret_cast<float*>(smem + off); off += WAVE * 4;
        qlds = reinterpret_cast<float*>(smem + off); off += (size_t)gv.stride * 4;
        vis.mask = (1u << a.hash_log2) - 1; vis.shift = 32 - a.hash_log2;
        vis.limit = (vis.mask + 1) - (vis.mask + 1) / 4 - WAVE;
        vis.tab = GHASH ? a.hash_global + ((size_t)q << a.hash_log2) : reinterpret_cast<uint32_t*>(smem + off);

        if (lane == 0) a.out_n[q] = 0;
        if (a.entry == NONE32) return;                                     // hnsw.rs:515-518

        // ---- query -> LDS (normalised for Cosine: hnsw.rs:526-536, 752-764) ----
        const float* qsrc = (a.mode == MODE_BUILD) ? gv.vecs + (size_t)a.query_slots[q] * gv.stride : a.queries + (size_t)q * gv.dim;
        for (uint32_t i = lane; i < gv.stride; i += WAVE) qlds[i] = (i < gv.dim) ? qsrc[i] : 0.f;
        wave_sync();
        if (gv.metric == VDB_COSINE && a.mode != MODE_BUILD && a.mode != MODE_LITERAL) {
            float s = -0.0f;                                               // every lane runs the same sequential sum
            for (uint32_t i = 0; i < gv.dim; ++i) { float p = qlds[i] * qlds[i]; s = s + p; }
            const float norm = __builtin_sqrtf(s);
            wave_sync();
            if (norm != 0.0f) for (uint32_t i = lane; i < gv.dim; i += WAVE) qlds[i] = qlds[i] / norm;
            wave_sync();
        }
        if (!STRICT) qr.load(qlds, gv.stride, gv.metric);

        if (a.mode == MODE_LITERAL) {
            uint32_t cur = a.q_entry ? a.q_entry[q] : a.entry;
            for (uint32_t l = a.max_level; l >= 1 && !(flags & QF_FILTER_ERROR); --l) cur = greedy_filtered(cur, l);   // hnsw.rs:628-630
            if (!(flags & QF_FILTER_ERROR)) literal_filtered(q, cur);
            if (!(flags & (QF_HASH_OVERFLOW | QF_CAND_OVERFLOW | QF_FILTER_ERROR))) emit(q, R, true);
        } else if (a.mode == MODE_BUILD) {
            const uint32_t lvl = a.q_levels[q];
            uint32_t cur = a.entry;
            for (uint32_t l = a.max_level; l > lvl; --l) cur = greedy(cur, l);          // hnsw.rs:169-173
            for (int l = (int)lvl; l >= 0; --l) {                                       // hnsw.rs:175-199
                beam<false>(cur, (uint32_t)l, a.ef);
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
            for (uint32_t l = a.max_level; l >= 1; --l) cur = greedy(cur, l);           // hnsw.rs:538-541
            if (a.mode == MODE_MASKED) { beam<true>(cur, 0, a.ef); if (!(flags & QF_HASH_OVERFLOW)) emit(q, P, false); }
            else { beam<false>(cur, 0, a.ef); if (!(flags & QF_HASH_OVERFLOW)) emit(q, R, false); }
        }
        // on a filter error `expanded` carries the slot whose evaluation failed
        if (a.stats && lane == 0) a.stats[q] = {rows_scored, (flags & QF_FILTER_ERROR) ? err_slot : expanded, adj_entries, flags};
    }
};

template <int G, int CH, int PASSES, bool STRICT, bool GHASH>
__global__ void __launch_bounds__(64) search_kernel(SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Searcher<G, CH, PASSES, STRICT, GHASH> s(a);
    s.run(blockIdx.x, smem);
}

template <int G, int CH, int PASSES, bool STRICT>
void launch_search_t(const SearchArgs& a, size_t lds, hipStream_t st) {
    if (a.hash_global) hipLaunchKernelGGL((search_kernel<G, CH, PASSES, STRICT, true>), dim3(a.nq), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((search_kernel<G, CH, PASSES, STRICT, false>), dim3(a.nq), dim3(64), lds, st, a);
}

}  // namespace

size_t search_lds_bytes(const SearchArgs& a) {
    const size_t ef_cap = (a.ef + 63u) & ~63u;
    size_t b = ef_cap * 8 + (a.mode == MODE_MASKED ? ef_cap * 8 : 0) + 64 * 8 + (size_t)a.g.stride * 4;
    if (!a.hash_global) b += (size_t)4 << a.hash_log2;
    return b;
}

void launch_search(const SearchArgs& a, hipStream_t st) {
    if (a.nq == 0) return;
    const size_t lds = search_lds_bytes(a);
    if (lds > 160 * 1024) throw Failure(err_arg("search working set exceeds the 160 KiB LDS of a CU"));
    if (a.arith == VDB_ARITH_STRICT) { launch_search_t<0, 0, 0, true>(a, lds, st); }
    else {
        const uint32_t chunks = a.g.stride / 4;
        if (chunks <= 8) launch_search_t<8, 1, 4, false>(a, lds, st);
        else if (chunks <= 16) launch_search_t<8, 2, 4, false>(a, lds, st);
        else if (chunks <= 24) launch_search_t<8, 3, 4, false>(a, lds, st);
        else if (chunks <= 32) launch_search_t<8, 4, 4, false>(a, lds, st);
        else if (chunks <= 64) launch_search_t<64, 1, 8, false>(a, lds, st);
        else if (chunks <= 128) launch_search_t<64, 2, 8, false>(a, lds, st);
        else if (chunks <= 192) launch_search_t<64, 3, 8, false>(a, lds, st);
        else if (chunks <= 256) launch_search_t<64, 4, 8, false>(a, lds, st);
        else if (chunks <= 512) launch_search_t<64, 8, 4, false>(a, lds, st);
        else throw Failure(err_arg("dimension above 2048 is not supported"));
    }
    VDB_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Streaming distance scan: every row of `base` against up to 8 queries held in LDS.
// One wave handles rows in a grid-stride loop; FAST uses G-lane groups as above, STRICT one
// lane per row.
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int SCAN_Q = 8;

__global__ void __launch_bounds__(256) scan_strict_kernel(const float* __restrict__ queries, uint32_t nq, const float* __restrict__ base,
                                                         uint32_t n, uint32_t dim, uint32_t stride, int metric, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* q = reinterpret_cast<float*>(smem);
    const uint32_t qpad = (dim + 3u) & ~3u;
    const uint32_t q0 = blockIdx.y * SCAN_Q;
    const uint32_t qn = min((uint32_t)SCAN_Q, nq - q0);
    for (uint32_t i = threadIdx.x; i < qn * qpad; i += blockDim.x) {
        const uint32_t t = i / qpad, d = i % qpad;
        q[i] = d < dim ? queries[(size_t)(q0 + t) * dim + d] : 0.f;
    }
    __syncthreads();
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const float* row = base + (size_t)r * stride;
        for (uint32_t t = 0; t < qn; ++t)
            out[(size_t)(q0 + t) * n + r] = strict_pair(q + t * qpad, row, dim, metric, true);
    }
}

template <int G, int CH>
__global__ void __launch_bounds__(256) scan_fast_kernel(const float* __restrict__ queries, uint32_t nq, const float* __restrict__ base,
                                                       uint32_t n, uint32_t dim, uint32_t stride, int metric, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* q = reinterpret_cast<float*>(smem);
    const uint32_t qpad = (dim + 3u) & ~3u;
    const uint32_t q0 = blockIdx.y * SCAN_Q;
    const uint32_t qn = min((uint32_t)SCAN_Q, nq - q0);
    for (uint32_t i = threadIdx.x; i < SCAN_Q * qpad; i += blockDim.x) {
        const uint32_t t = i / qpad, d = i % qpad;
        q[i] = (t < qn && d < dim) ? queries[(size_t)(q0 + t) * dim + d] : 0.f;
    }
    __syncthreads();
    constexpr int RPP = WAVE / G;
    const int lane = lane_id(), g = lane % G, rr = lane / G;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, nwaves = gridDim.x * blockDim.x / WAVE;
    const uint32_t nchunks = qpad >> 2;
    float qn2[SCAN_Q];
    if (metric == VDB_COSINE) {
#pragma unroll
        for (int t = 0; t < SCAN_Q; ++t) {
            float acc = 0.f;
            for (int j = 0; j < CH; ++j) {
                const uint32_t c = g + G * j;
                if (c < nchunks) { const float4 v = reinterpret_cast<const float4*>(q + t * qpad)[c];
                    acc = __builtin_fmaf(v.x, v.x, acc); acc = __builtin_fmaf(v.y, v.y, acc); acc = __builtin_fmaf(v.z, v.z, acc); acc = __builtin_fmaf(v.w, v.w, acc); }
            }
            qn2[t] = __builtin_sqrtf(group_sum<G>(acc));
        }
    }
    for (uint32_t r0 = wave * RPP; r0 < n; r0 += nwaves * RPP) {
        const uint32_t r = r0 + rr;
        const bool live = r < n;
        float4 x[CH];
        const float4* row = reinterpret_cast<const float4*>(base + (size_t)(live ? r : 0) * stride);
#pragma unroll
        for (int j = 0; j < CH; ++j) { const uint32_t c = g + G * j; x[j] = (live && c < nchunks) ? row[c] : make_float4(0.f, 0.f, 0.f, 0.f); }
        float xb = 0.f;
        if (metric == VDB_COSINE) {
#pragma unroll
            for (int j = 0; j < CH; ++j) { xb = __builtin_fmaf(x[j].x, x[j].x, xb); xb = __builtin_fmaf(x[j].y, x[j].y, xb); xb = __builtin_fmaf(x[j].z, x[j].z, xb); xb = __builtin_fmaf(x[j].w, x[j].w, xb); }
            xb = __builtin_sqrtf(group_sum<G>(xb));
        }
#pragma unroll
        for (int t = 0; t < SCAN_Q; ++t) {
            if (t >= (int)qn) break;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const uint32_t c = g + G * j;
                const float4 qq = (c < nchunks) ? reinterpret_cast<const float4*>(q + t * qpad)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 v = x[j];
                if (metric == VDB_EUCLIDEAN) {
                    float d0 = qq.x - v.x, d1 = qq.y - v.y, d2 = qq.z - v.z, d3 = qq.w - v.w;
                    acc = __builtin_fmaf(d0, d0, acc); acc = __builtin_fmaf(d1, d1, acc); acc = __builtin_fmaf(d2, d2, acc); acc = __builtin_fmaf(d3, d3, acc);
                } else {
                    acc = __builtin_fmaf(qq.x, v.x, acc); acc = __builtin_fmaf(qq.y, v.y, acc); acc = __builtin_fmaf(qq.z, v.z, acc); acc = __builtin_fmaf(qq.w, v.w, acc);
                }
            }
            acc = group_sum<G>(acc);
            float raw = metric == VDB_EUCLIDEAN ? __builtin_sqrtf(acc) : (metric == VDB_DOT ? acc : 1.0f - acc / (qn2[t] * xb + 1e-10f));
            if (g == 0 && live) out[(size_t)(q0 + t) * n + r] = raw;
        }
    }
}

__global__ void __launch_bounds__(64) score_rows_kernel(GraphView gv, uint32_t query_slot, const uint32_t* __restrict__ slots, uint32_t n, float* __restrict__ raw) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) raw[i] = strict_pair(gv.vecs + (size_t)query_slot * gv.stride, gv.vecs + (size_t)slots[i] * gv.stride, gv.dim, gv.metric, false);
}

// hnsw.rs:752-764: norm = sqrt(sequential sum of squares); zero norm leaves the row unchanged.
__global__ void __launch_bounds__(64) normalize_rows_kernel(float* __restrict__ vecs, uint32_t stride, uint32_t dim, uint32_t first, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* row = vecs + (size_t)(first + i) * stride;
    float s = -0.0f;
    for (uint32_t d = 0; d < dim; ++d) { float p = row[d] * row[d]; s = s + p; }
    const float norm = __builtin_sqrtf(s);
    if (norm == 0.0f) return;
    for (uint32_t d = 0; d < dim; ++d) row[d] = row[d] / norm;
}

}  // namespace

void launch_score_scan(const float* queries, uint32_t nq, const float* base, uint32_t n, uint32_t dim, uint32_t stride,
                       int metric, int arith, float* out, hipStream_t st) {
    if (nq == 0 || n == 0) return;
    const uint32_t qpad = (dim + 3u) & ~3u;
    const size_t lds = (size_t)SCAN_Q * qpad * 4;
    const uint32_t qblocks = (nq + SCAN_Q - 1) / SCAN_Q;
    const bool aligned = (stride % 4 == 0) && ((uintptr_t)base % 16 == 0);
    if (arith == VDB_ARITH_STRICT || !aligned || qpad / 4 > 512) {
        if (stride % 4 != 0 || (uintptr_t)base % 16 != 0) {
            // strict_pair reads rows with 16-byte loads: require the padded layout
            if (dim >= 4) throw Failure(err_arg("base rows must be 16-byte aligned with a stride that is a multiple of 4 floats"));
        }
        const uint32_t blocks = std::min<uint32_t>((n + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(scan_strict_kernel, dim3(blocks, qblocks), dim3(256), lds, st, queries, nq, base, n, dim, stride, metric, out);
    } else {
        const uint32_t chunks = qpad / 4;
        const uint32_t blocks = 256 * 8;
#define VDB_SCAN(G, CH) hipLaunchKernelGGL((scan_fast_kernel<G, CH>), dim3(blocks, qblocks), dim3(256), lds, st, queries, nq, base, n, dim, stride, metric, out)
        if (chunks <= 8) VDB_SCAN(8, 1); else if (chunks <= 16) VDB_SCAN(8, 2); else if (chunks <= 24) VDB_SCAN(8, 3);
        else if (chunks <= 32) VDB_SCAN(8, 4); else if (chunks <= 64) VDB_SCAN(64, 1); else if (chunks <= 128) VDB_SCAN(64, 2);
        else if (chunks <= 192) VDB_SCAN(64, 3); else if (chunks <= 256) VDB_SCAN(64, 4); else VDB_SCAN(64, 8);
#undef VDB_SCAN
    }
    VDB_HIP(hipGetLastError());
}

void launch_score_rows(const GraphView& g, uint32_t query_slot, const uint32_t* slots, uint32_t n, int, float* raw, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(score_rows_kernel, dim3((n + 63) / 64), dim3(64), 0, st, g, query_slot, slots, n, raw);
    VDB_HIP(hipGetLastError());
}

void launch_normalize_rows(float* vecs, uint32_t stride, uint32_t dim, uint32_t first, uint32_t n, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((n + 63) / 64), dim3(64), 0, st, vecs, stride, dim, first, n);
    VDB_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Exact k-NN (strict arithmetic) and the k-way merge.  Lists are ordered by (sort_key asc, id asc).
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int KNN_QT = 16;        // queries per wave
constexpr uint32_t KNN_MAX_K = 256;

struct TopK {        // wave-cooperative (key, id)-ordered list in LDS
    float* key; uint64_t* id; float* raw; uint32_t n, cap;
    __device__ bool admits(float k, uint64_t i) const {
        if (n < cap) return true;
        const float wk = key[cap - 1];
        return k < wk || (k == wk && i < id[cap - 1]);
    }
    __device__ void insert(float k, uint64_t idv, float rawv) {
        const int lane = lane_id();
        uint32_t pos = 0;
        for (uint32_t b = 0; b < n; b += WAVE) {
            const uint32_t i = b + lane;
            pos += popc64(ballot(i < n && (key[i] < k || (key[i] == k && id[i] < idv))));
        }
        const uint32_t last = (n < cap) ? n : cap - 1;
        for (int b = (int)(last & ~63u); b >= (int)(pos & ~63u); b -= WAVE) {
            const uint32_t i = (uint32_t)b + lane;
            const bool mv = i > pos && i <= last;
            float tk = 0.f, tr = 0.f; uint64_t ti = 0;
            if (mv) { tk = key[i - 1]; ti = id[i - 1]; tr = raw[i - 1]; }
            wave_sync();
            if (mv) { key[i] = tk; id[i] = ti; raw[i] = tr; }
            wave_sync();
        }
        if (lane == 0) { key[pos] = k; id[pos] = idv; raw[pos] = rawv; }
        if (n < cap) ++n;
        wave_sync();
    }
    // Offer one candidate per lane (valid lanes only), in lane order.
    __device__ void offer(bool valid, float k, uint64_t idv, float rawv) {
        uint64_t am = ballot(valid && admits(k, idv));
        while (am) {
            const int j = __ffsll((unsigned long long)am) - 1; am &= am - 1;
            const float kk = readlane_f(k, j), rr = readlane_f(rawv, j);
            const uint64_t ii = ((uint64_t)readlane_u((uint32_t)(idv >> 32), j) << 32) | readlane_u((uint32_t)idv, j);
            if (!admits(kk, ii)) continue;
            insert(kk, ii, rr);
        }
    }
};

// grid (n_chunks, ceil(nq / KNN_QT)); one wave.  Each lane owns one base row at a time and keeps
// KNN_QT sequential accumulators, one per query (queries broadcast from LDS).
__global__ void __launch_bounds__(64) exact_knn_kernel(GraphView gv, const float* __restrict__ queries, uint32_t nq, uint32_t top_k,
                                                      vdb_scored_point* __restrict__ scratch, uint32_t n_chunks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id();
    const uint32_t q0 = blockIdx.y * KNN_QT, qn = min((uint32_t)KNN_QT, nq - q0);
    const uint32_t dim = gv.dim, stride = gv.stride;
    float* q = reinterpret_cast<float*>(smem);                                 // KNN_QT x stride
    size_t off = (size_t)KNN_QT * stride * 4;
    uint64_t* lid = reinterpret_cast<uint64_t*>(smem + off); off += (size_t)KNN_QT * top_k * 8;
    float* lkey = reinterpret_cast<float*>(smem + off); off += (size_t)KNN_QT * top_k * 4;
    float* lraw = reinterpret_cast<float*>(smem + off); off += (size_t)KNN_QT * top_k * 4;
    float* qna = reinterpret_cast<float*>(smem + off);                        // KNN_QT: sqrt(sum q^2)
    for (uint32_t i = lane; i < KNN_QT * stride; i += WAVE) {
        const uint32_t t = i / stride, d = i % stride;
        q[i] = (t < qn && d < dim) ? queries[(size_t)(q0 + t) * dim + d] : 0.f;
    }
    wave_sync();
    if (gv.metric == VDB_COSINE) {                                            // hnsw.rs:532-536
        for (uint32_t t = 0; t < qn; ++t) {
            float s = -0.0f;
            for (uint32_t d = 0; d < dim; ++d) { float p = q[t * stride + d] * q[t * stride + d]; s = s + p; }
            const float norm = __builtin_sqrtf(s);
            wave_sync();
            if (norm != 0.0f) for (uint32_t d = lane; d < dim; d += WAVE) q[t * stride + d] = q[t * stride + d] / norm;
            wave_sync();
            float s2 = -0.0f;
            for (uint32_t d = 0; d < dim; ++d) { float p = q[t * stride + d] * q[t * stride + d]; s2 = s2 + p; }
            if (lane == 0) qna[t] = __builtin_sqrtf(s2);
        }
        wave_sync();
    }
    uint32_t ln[KNN_QT];
#pragma unroll
    for (int t = 0; t < KNN_QT; ++t) ln[t] = 0;

    const uint32_t per = (gv.n + n_chunks - 1) / n_chunks;
    const uint32_t r_begin = blockIdx.x * per, r_end = min(gv.n, r_begin + per);
    for (uint32_t rb = r_begin; rb < r_end; rb += WAVE) {
        const uint32_t r = rb + lane;
        const bool live = r < r_end && !(gv.deleted && bit_test(gv.deleted, r));
        const float* row = gv.vecs + (size_t)(r < r_end ? r : r_begin) * stride;
        float acc[KNN_QT], accb = -0.0f;
#pragma unroll
        for (int t = 0; t < KNN_QT; ++t) acc[t] = -0.0f;
        for (uint32_t d = 0; d < dim; d += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(row + d);
            const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int t = 0; t < KNN_QT; ++t) {
                const float4 qv = *reinterpret_cast<const float4*>(q + t * stride + d);
                const float qa[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (d + k < dim) {
                        if (gv.metric == VDB_EUCLIDEAN) { float df = qa[k] - xa[k]; float p = df * df; acc[t] = acc[t] + p; }
                        else { float p = qa[k] * xa[k]; acc[t] = acc[t] + p; }
                    }
                }
            }
            if (gv.metric == VDB_COSINE) {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (d + k < dim) { float p = xa[k] * xa[k]; accb = accb + p; }
            }
        }
        const uint64_t myid = live ? gv.ids[r] : 0ull;
#pragma unroll
        for (int t = 0; t < KNN_QT; ++t) {
            if (t >= (int)qn) break;
            float raw;
            if (gv.metric == VDB_EUCLIDEAN) raw = __builtin_sqrtf(acc[t]);
            else if (gv.metric == VDB_DOT) raw = acc[t];
            else { float den = qna[t] * __builtin_sqrtf(accb); den = den + 1e-10f; float qq = acc[t] / den; raw = 1.0f - qq; }
            const float key = gv.metric == VDB_DOT ? -raw : raw;
            TopK L{lkey + t * top_k, lid + t * top_k, lraw + t * top_k, ln[t], top_k};
            L.offer(live, key, myid, raw);
            ln[t] = L.n;
        }
    }
    // write this chunk's lists: scratch[(q * n_chunks + chunk) * top_k + i]; id = ~0 marks unused
#pragma unroll
    for (int t = 0; t < KNN_QT; ++t) {
        if (t >= (int)qn) break;
        vdb_scored_point* dst = scratch + ((size_t)(q0 + t) * n_chunks + blockIdx.x) * top_k;
        for (uint32_t i = lane; i < top_k; i += WAVE) {
            vdb_scored_point sp;
            if (i < ln[t]) { sp.id = lid[t * top_k + i]; sp.raw_score = lraw[t * top_k + i]; sp.sort_key = lkey[t * top_k + i]; }
            else { sp.id = ~0ull; sp.raw_score = 0.f; sp.sort_key = __int_as_float(0x7f800000); }
            dst[i] = sp;
        }
    }
}

// One wave per query: merge `n_lists` rows of `k` records each.  `counts` == nullptr: rows are
// padded with id == ~0 instead.
__global__ void __launch_bounds__(64) merge_kernel(const vdb_scored_point* __restrict__ lists, const uint32_t* __restrict__ counts,
                                                  uint32_t n_lists, uint32_t nq, uint32_t k, int shard_major,
                                                  vdb_scored_point* __restrict__ out, uint32_t* __restrict__ out_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id();
    const uint32_t q = blockIdx.x;
    uint64_t* lid = reinterpret_cast<uint64_t*>(smem);
    float* lkey = reinterpret_cast<float*>(smem + (size_t)k * 8);
    float* lraw = reinterpret_cast<float*>(smem + (size_t)k * 12);
    TopK L{lkey, lid, lraw, 0, k};
    for (uint32_t s = 0; s < n_lists; ++s) {
        const size_t rowi = shard_major ? ((size_t)s * nq + q) : ((size_t)q * n_lists + s);
        const vdb_scored_point* row = lists + rowi * k;
        const uint32_t cnt = counts ? counts[rowi] : k;
        for (uint32_t b = 0; b < cnt; b += WAVE) {
            const uint32_t i = b + lane;
            vdb_scored_point sp; sp.id = ~0ull; sp.raw_score = 0.f; sp.sort_key = 0.f;
            if (i < cnt) sp = row[i];
            L.offer(i < cnt && sp.id != ~0ull, sp.sort_key, sp.id, sp.raw_score);
        }
    }
    for (uint32_t i = lane; i < L.n; i += WAVE) { vdb_scored_point sp; sp.id = lid[i]; sp.raw_score = lraw[i]; sp.sort_key = lkey[i]; out[(size_t)q * k + i] = sp; }
    if (lane == 0) out_n[q] = L.n;
}

}  // namespace

uint32_t exact_knn_chunks(uint32_t n) { return std::max<uint32_t>(1, std::min<uint32_t>(64, (n + 4095) / 4096)); }

void launch_exact_knn(const GraphView& g, const float* queries, uint32_t nq, uint32_t top_k, vdb_scored_point* out, uint32_t* out_n,
                      vdb_scored_point* scratch, uint32_t n_chunks, hipStream_t st) {
    if (nq == 0) return;
    if (top_k == 0 || top_k > KNN_MAX_K) throw Failure(err_arg("exact k-NN supports 1 <= top_k <= 256"));
    const size_t lds = (size_t)KNN_QT * g.stride * 4 + (size_t)KNN_QT * top_k * 16 + KNN_QT * 4;
    if (lds > 160 * 1024) throw Failure(err_arg("exact k-NN working set exceeds LDS"));
    hipLaunchKernelGGL(exact_knn_kernel, dim3(n_chunks, (nq + KNN_QT - 1) / KNN_QT), dim3(64), lds, st, g, queries, nq, top_k, scratch, n_chunks);
    VDB_HIP(hipGetLastError());
    hipLaunchKernelGGL(merge_kernel, dim3(nq), dim3(64), (size_t)top_k * 16, st, scratch, (const uint32_t*)nullptr, n_chunks, nq, top_k, 0, out, out_n);
    VDB_HIP(hipGetLastError());
}

void launch_topk_merge(const vdb_scored_point* lists, const uint32_t* counts, uint32_t n_shards, uint32_t nq, uint32_t k,
                       vdb_scored_point* out, uint32_t* out_n, hipStream_t st) {
    if (nq == 0) return;
    if (k == 0 || k > 1024) throw Failure(err_arg("merge supports 1 <= k <= 1024"));
    hipLaunchKernelGGL(merge_kernel, dim3(nq), dim3(64), (size_t)k * 16, st, lists, counts, n_shards, nq, k, 1, out, out_n);
    VDB_HIP(hipGetLastError());
}

}  // namespace vdb
