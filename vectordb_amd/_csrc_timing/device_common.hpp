// Device-side building blocks shared by the gfx950 kernels of libvdbmi: wave64 helpers, the fused
// layer-0 adjacency reader, the distance arithmetic (STRICT and FAST), the wave-cooperative sorted
// list and the exact visited set.  Every search block is ONE wavefront.
//
// Translation units including this header are compiled with -ffp-contract=off: the STRICT paths rely
// on separately rounded multiply and add (metric.rs:15-35 never fuses); FAST paths ask for FMAs
// explicitly.
#pragma once
#include "kernels.hpp"

namespace vdb {
namespace {

constexpr int WAVE = 64;
constexpr uint32_t FLAG_EXPANDED = 0x80000000u;
constexpr uint32_t HASH_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int popc64(uint64_t m) { return __popcll(m); }
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (lane == 0) ? 0ull : (~0ull >> (64 - lane)); }
// Number of set bits of `m` below this lane (v_mbcnt pair).
__device__ __forceinline__ uint32_t rank_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
// Every search block is a single wavefront and the LDS executes one wavefront's DS instructions in
// issue order, so lanes exchange data through LDS with no hardware wait: only the compiler has to
// keep the accesses in program order.  (A __syncthreads() here would also drain vmcnt and expose
// the latency of every global load still in flight, the adjacency prefetch included.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

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

// One node's adjacency as a wave sees it.  Layer 0 lives in fixed 64-dword blocks: word 0 = degree,
// words 1..62 = the first 62 neighbours, word 63 = where the remainder starts in the overflow pool
// (degree > 62 only).  Lane i holds word i, so ONE coalesced 256-byte load brings header and
// neighbours together and the address depends on the slot alone (it can be issued ahead of time).
// Upper layers are small and keep (descriptor -> pool) lists.  Neighbours come out in list order:
// ascending chunk, ascending lane.
struct Adj {
    uint32_t deg; bool fused; uint32_t w; const uint32_t* rest;
    __device__ __forceinline__ uint32_t chunks() const {
        if (fused) return deg <= ADJ_INLINE ? (deg ? 1u : 0u) : 1u + (deg - ADJ_INLINE + 63u) / 64u;
        return (deg + 63u) / 64u;
    }
    __device__ __forceinline__ bool get(uint32_t c, uint32_t& nb) const {
        const uint32_t lane = (uint32_t)lane_id();
        if (fused) {
            if (c == 0) { nb = w; return lane >= 1 && lane <= ADJ_INLINE && lane - 1 < deg; }
            const uint32_t i = (c - 1) * 64u + lane;
            const bool v = ADJ_INLINE + i < deg;
            nb = v ? rest[i] : 0u;
            return v;
        }
        const uint32_t i = c * 64u + lane;
        const bool v = i < deg;
        nb = v ? rest[i] : 0u;
        return v;
    }
};
__device__ __forceinline__ uint32_t load_adj_word(const GraphView& g, uint32_t slot) { return g.adj0[(size_t)slot * ADJ_BLOCK + lane_id()]; }
__device__ __forceinline__ Adj open_adj0(const GraphView& g, uint32_t w) {
    Adj a; a.fused = true; a.w = w; a.deg = readlane_u(w, 0);
    a.rest = g.ovf + readlane_u(w, 63);
    return a;
}
__device__ __forceinline__ Adj open_adj(const GraphView& g, uint32_t level, uint32_t slot) {
    if (level == 0) return open_adj0(g, load_adj_word(g, slot));
    Adj a; a.fused = false; a.w = 0; a.deg = 0; a.rest = nullptr;
    const uint32_t base = g.upper_base[slot];
    if (base == NONE32 || level > g.upper_cnt[slot]) return a;
    const uint64_t d = g.udesc[base + level - 1];
    a.rest = g.pool + (d & 0xFFFFFFFFFFull); a.deg = (uint32_t)(d >> 40);
    return a;
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
    __device__ __forceinline__ void load(const float* qlds, uint32_t stride, int metric) {
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
// METRIC >= 0 fixes the metric at compile time (branch-free inner loops); -1 reads gv.metric.
template <int G, int CH, int PASSES, int METRIC>
__device__ __forceinline__ void score_cands_fast(const GraphView& gv, const QueryRegs<G, CH>& qr, const uint32_t* cand, int m, float* craw) {
    constexpr int RPP = WAVE / G;
    const int metric = METRIC >= 0 ? METRIC : gv.metric;
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
                if (metric == VDB_EUCLIDEAN) {
                    float d0 = q.x - v.x, d1 = q.y - v.y, d2 = q.z - v.z, d3 = q.w - v.w;
                    acc = __builtin_fmaf(d0, d0, acc); acc = __builtin_fmaf(d1, d1, acc);
                    acc = __builtin_fmaf(d2, d2, acc); acc = __builtin_fmaf(d3, d3, acc);
                } else {
                    acc = __builtin_fmaf(q.x, v.x, acc); acc = __builtin_fmaf(q.y, v.y, acc);
                    acc = __builtin_fmaf(q.z, v.z, acc); acc = __builtin_fmaf(q.w, v.w, acc);
                    if (metric == VDB_COSINE) {
                        accb = __builtin_fmaf(v.x, v.x, accb); accb = __builtin_fmaf(v.y, v.y, accb);
                        accb = __builtin_fmaf(v.z, v.z, accb); accb = __builtin_fmaf(v.w, v.w, accb);
                    }
                }
            }
            acc = group_sum<G>(acc);
            float raw;
            if (metric == VDB_EUCLIDEAN) raw = __builtin_sqrtf(acc);
            else if (metric == VDB_DOT) raw = acc;
            else {
                accb = group_sum<G>(accb);
                raw = 1.0f - acc / (qr.qnorm * __builtin_sqrtf(accb) + 1e-10f);
            }
            if (g == 0 && ci < m) craw[ci] = raw;
        }
    }
}

__device__ __forceinline__ void score_cands_strict(const GraphView& gv, const float* qlds, const uint32_t* cand, int m, float* craw) {
    const int lane = lane_id();
    if (lane < m) craw[lane] = strict_pair(qlds, gv.vecs + (size_t)cand[lane] * gv.stride, gv.dim, gv.metric, true);
}

// ---------------------------------------------------------------------------------------------
// Wave-cooperative sorted list in LDS (ascending key, arrival order among equal keys)
// ---------------------------------------------------------------------------------------------
struct SortedList {
    float* key; uint32_t* val;            // live buffers
    float* key2; uint32_t* val2;          // spare buffers the batched merge writes into
    uint32_t n, cap;
    __device__ __forceinline__ float worst() const { return key[cap - 1]; }
    // Caller guarantees n < cap or k < key[n-1].  All lanes call with the same (k, v).  Returns
    // the position taken.
    __device__ __forceinline__ uint32_t insert(float k, uint32_t v) {
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
        return pos;
    }
    // Offers one candidate per lane (lanes with `pass`), equivalent to calling insert() for each in
    // lane order with the acceptance test of hnsw.rs:486 (list not full or key < worst): the result
    // is the `cap` smallest of list + candidates, existing entries before candidates on equal
    // keys, candidates among themselves in lane order.  Lists up to 256 entries are merged by rank
    // in one pass; longer lists fall back to one insert() per candidate.  Returns the lowest
    // position a candidate took (NONE32 when none entered).
    __device__ __forceinline__ uint32_t offer(bool pass, float kj, uint32_t sj) {
        const int lane = lane_id();
        if (n >= cap) pass = pass && kj < worst();
        uint64_t pm = ballot(pass);
        const int mp = popc64(pm);
        uint32_t lowest = NONE32;
        if (mp == 0) return lowest;
        if (cap > 256) {
            while (pm) {
                const int j = __ffsll((unsigned long long)pm) - 1; pm &= pm - 1;
                const float k = readlane_f(kj, j);
                if (n >= cap && !(k < worst())) continue;
                const uint32_t p = insert(k, readlane_u(sj, j));
                lowest = p < lowest ? p : lowest;
            }
            return lowest;
        }
        const float INF = __int_as_float(0x7f800000);
        float ek[4]; uint32_t ev[4], sh[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t i = c * WAVE + lane;
            const bool have = i < n;
            ek[c] = have ? key[i] : INF; ev[c] = have ? val[i] : 0u; sh[c] = 0;
        }
        uint32_t mypos = 0;
        uint64_t it = pm;
        while (it) {
            const int j = __ffsll((unsigned long long)it) - 1; it &= it - 1;
            const float k = readlane_f(kj, j);
            uint32_t pos = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c * WAVE < (int)n) {
                    const bool le = ek[c] <= k;                       // existing entries that stay in front of k
                    pos += popc64(ballot(le));
                    sh[c] += le ? 0u : 1u;                            // the others move back by one
                }
            }
            pos += popc64(ballot(pass && (kj < k || (kj == k && lane < j))));
            if (lane == j) mypos = pos;
            lowest = pos < lowest ? pos : lowest;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t i = c * WAVE + lane;
            if (i < n) { const uint32_t np = i + sh[c]; if (np < cap) { key2[np] = ek[c]; val2[np] = ev[c]; } }
        }
        if (pass && mypos < cap) { key2[mypos] = kj; val2[mypos] = sj; }
        n = (n + mp < cap) ? n + mp : cap;
        float* tk = key; key = key2; key2 = tk;
        uint32_t* tv = val; val = val2; val2 = tv;
        wave_sync();
        return lowest < cap ? lowest : NONE32;
    }
};

// Visited set (the per-call HashSet of hnsw.rs:435): exact, open addressing over slot ids with
// bucketed probing (16-byte buckets of four ids, filled front to back, never emptied).  The primary
// table lives in LDS; once it holds `limit` ids it is frozen and further ids go to a spill table in
// HBM taken from a per-launch pool, so lookups stay exact at any search length.
struct Visited {
    uint32_t* tab; uint32_t mask, shift, count, limit;         // mask/shift address the buckets
    uint32_t* spill; uint32_t smask, sshift, scount, slimit;   // spill == nullptr until needed
    static constexpr uint32_t HASH_MUL = 2654435761u;

    // One bucket against id s: 0 = s is present, 1..4 = (first free slot + 1), 5 = full without s.
    static __device__ __forceinline__ uint32_t classify(const uint4 v, uint32_t s) {
        const uint32_t d = min(min(v.x ^ s, v.y ^ s), min(v.z ^ s, v.w ^ s));
        const uint32_t fr = v.x == HASH_EMPTY ? 1u : v.y == HASH_EMPTY ? 2u : v.z == HASH_EMPTY ? 3u : v.w == HASH_EMPTY ? 4u : 5u;
        return d == 0 ? 0u : fr;
    }
    // HBM buckets are read with device-scope atomic loads: the table is written by atomics that
    // bypass this CU's vector L1, so a plain load could keep returning a stale line.
    static __device__ __forceinline__ uint4 load_bucket_hbm(uint32_t* t, uint32_t b) {
        unsigned long long* p = reinterpret_cast<unsigned long long*>(t + 4 * b);
        const unsigned long long lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
    }
    // Generic probe: 0 = already present, 1 = inserted, 2 = absent (FROZEN: look only).
    template <bool FROZEN, bool HBM> static __device__ __forceinline__ int probe(uint32_t* t, uint32_t bmask, uint32_t b, uint32_t s) {
        for (;;) {
            const uint4 v = HBM ? load_bucket_hbm(t, b) : *reinterpret_cast<const uint4*>(t + 4 * b);
            const uint32_t c = classify(v, s);
            if (c == 0) return 0;
            if (c == 5) { b = (b + 1) & bmask; continue; }                         // bucket full: next bucket
            if (FROZEN) return 2;
            const uint32_t old = atomicCAS(&t[4 * b + c - 1], HASH_EMPTY, s);
            if (old == HASH_EMPTY) return 1;
            if (old == s) return 0;
            // another lane took the slot: look at the same bucket again
        }
    }
    __device__ __forceinline__ bool insert(uint32_t s) {
        const uint32_t hv = s * HASH_MUL;
        if (!spill) return probe<false, false>(tab, mask, hv >> shift, s) == 1;
        if (probe<true, false>(tab, mask, hv >> shift, s) == 0) return false;
        return probe<false, true>(spill, smask, hv >> sshift, s) == 1;
    }
    // E independent ids per lane (one per neighbour chunk): the first bucket read and the first
    // CAS of all E are in flight together, so a step costs one LDS (or HBM) round trip per phase
    // instead of one per chunk; lanes that need a second bucket or lost a slot finish in the
    // generic probe, chunk by chunk.  Two lanes holding the same id (duplicates across chunks)
    // aim at the same slot and exactly one of them reports `fresh`.
    template <int E> __device__ __forceinline__ void insert_many(const uint32_t (&nb)[E], const bool (&want)[E], bool (&fresh)[E]) {
        uint32_t hv[E], st[E], old[E];     // st: 0 settled (not fresh), 1..4 CAS slot + 1, 5 generic probe, 6 fresh, 7 look in HBM
        uint4 v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { hv[e] = nb[e] * HASH_MUL; v[e] = *reinterpret_cast<const uint4*>(tab + 4 * (hv[e] >> shift)); }
        if (!spill) {
#pragma unroll
            for (int e = 0; e < E; ++e) st[e] = want[e] ? classify(v[e], nb[e]) : 0u;
#pragma unroll
            for (int e = 0; e < E; ++e) { old[e] = 0; if (st[e] >= 1 && st[e] <= 4) old[e] = atomicCAS(&tab[4 * (hv[e] >> shift) + st[e] - 1], HASH_EMPTY, nb[e]); }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) { const uint32_t c = classify(v[e], nb[e]); st[e] = !want[e] ? 0u : c == 0 ? 0u : c == 5 ? 5u : 7u; }
#pragma unroll
            for (int e = 0; e < E; ++e) if (st[e] == 7) v[e] = load_bucket_hbm(spill, hv[e] >> sshift);
#pragma unroll
            for (int e = 0; e < E; ++e) if (st[e] == 7) st[e] = classify(v[e], nb[e]);
#pragma unroll
            for (int e = 0; e < E; ++e) { old[e] = 0; if (st[e] >= 1 && st[e] <= 4) old[e] = atomicCAS(&spill[4 * (hv[e] >> sshift) + st[e] - 1], HASH_EMPTY, nb[e]); }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (st[e] >= 1 && st[e] <= 4) st[e] = old[e] == HASH_EMPTY ? 6u : old[e] == nb[e] ? 0u : 5u;
            fresh[e] = st[e] == 6;
            if (__ballot(st[e] == 5)) { if (st[e] == 5) fresh[e] = insert(nb[e]); }
        }
    }
};

// Visited set of the register-resident beam, 16 bits per id.  An id below 2^(bbits + 15) is mapped
// one-to-one (odd multiplier modulo a power of two) onto a bucket number (bbits) and a 15-bit tag, so
// "present" is never wrong; a bucket is 8 bytes = four tags (bit 15 marks a used entry), filled front
// to back, and one ds_read_b64 shows whether the id is present and where the first free entry is.
// An id that meets a full bucket replaces one of its four tags: the set may FORGET an id, never
// invent one.  A forgotten id that turns up again is scored again and then either fails the
// admission test it failed before (the bound only tightens) or is found in the result list by its
// key and id (the merge drops it), so results do not depend on what was forgotten — only the work does.
struct TagSet {
    uint32_t* tab; uint32_t idmask;
    static constexpr uint32_t MUL = 2654435761u, USED = 0x80008000u, WAYS = 4;
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ uint32_t pkmin(uint32_t a, uint32_t b) {
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
    }
    __device__ __forceinline__ void locate(uint32_t id, uint32_t& word0, uint32_t& tag) const {
        const uint32_t h = (id * MUL) & idmask;
        word0 = (h >> 15) << 1; tag = (h & 0x7FFFu) | 0x8000u;
    }
    static __device__ __forceinline__ bool present(const uint2 v, uint32_t tag) {
        const uint32_t t2 = tag | (tag << 16);
        uint32_t m = pkmin(v.x ^ t2, v.y ^ t2);
        m = pkmin(m, __builtin_amdgcn_alignbit(m, m, 16));
        return (m & 0xFFFFu) == 0;
    }
    static __device__ __forceinline__ uint32_t used(const uint2 v) { return __popc(v.x & USED) + __popc(v.y & USED); }
    // Entry that takes `tag` in a bucket with u used entries: the first free one, or, in a full
    // bucket, the entry two bits of the tag pick (its id is forgotten).
    static __device__ __forceinline__ uint32_t entry_for(uint32_t u, uint32_t tag) { return u < WAYS ? u : ((tag >> 5) & (WAYS - 1u)); }
    // CAS operands that put `tag` into entry e (< 4) of bucket v
    static __device__ __forceinline__ void slot_of(const uint2 v, uint32_t e, uint32_t tag, uint32_t& widx, uint32_t& expect, uint32_t& desired) {
        widx = e >> 1;
        expect = widx == 0 ? v.x : v.y;
        const uint32_t sh = (e & 1u) * 16u;
        desired = (expect & ~(0xFFFFu << sh)) | (tag << sh);
    }
    // Generic insert: 0 = the id was already in the set, 1 = added, 2 = added in place of another id.
    __device__ __forceinline__ int insert(uint32_t id) const {
        uint32_t w0, tag;
        locate(id, w0, tag);
        for (;;) {
            const uint2 v = *reinterpret_cast<const uint2*>(tab + w0);
            if (present(v, tag)) return 0;
            const uint32_t u = used(v);
            uint32_t widx, expect, desired;
            slot_of(v, entry_for(u, tag), tag, widx, expect, desired);
            if (atomicCAS(&tab[w0 + widx], expect, desired) == expect) return u < WAYS ? 1 : 2;
            // the word changed under us (another lane wrote this or the neighbouring entry): look again
        }
    }
};

}  // namespace
}  // namespace vdb
