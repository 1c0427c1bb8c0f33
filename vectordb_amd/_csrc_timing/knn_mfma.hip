// Exact k-NN, FAST arithmetic: a dense fp32 MFMA contraction (queries x base rows) with a fused
// top-k' epilogue selects candidates, then every candidate is re-scored with the STRICT
// arithmetic (metric.rs:15-35, one lane per row, sequential sums) and ranked by (sort_key, id), so
// the returned keys are the strict ones bit for bit.  The candidate pass keeps, per query and per
// chunk of the base set, the KP (16 or 32) rows with the smallest *ranking key* (L2: |x|^2 - 2 q.x, dot:
// -q.x, cosine: -q.x / |x|); it misses a true neighbour only if more than KP - top_k rows of
// one chunk fall within the fp32 rounding error of the contraction (~1e-6 relative) of it.
#include "device_common.hpp"
#include <algorithm>
#include <cstdlib>
#include <mutex>

namespace vdb {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int MQ = 128;            // queries per block: 4 waves x 32 (the N side of the 32x32 tile)
constexpr int MT = 32;             // base rows per tile (the M side)
// Candidates kept per (query, chunk): KP = 16 for top_k <= 12, 32 for top_k <= 28.  A 32-row tile is
// pushed in two halves of at most 16 entries per query; the per-query buffer of CAP entries is
// compacted to KP once fewer than 16 entries are free and never holds more than 64 entries, one per
// lane of the compacting wave.
template <int KP> struct KnnBuf { static constexpr int CAP = KP == 16 ? 40 : 44, TRIG = KP == 16 ? 24 : 38; };
constexpr float INF_F = __builtin_inff();

// Keeps the KP smallest entries of query `qi`'s candidate buffer, ordered by (ranking key, PointId) — the order of
// the final result, so rows that tie exactly (repeated vectors) survive by id whatever their slots are.  Ids are read
// only when two entries of the buffer hold the same key.  All lanes of the wave take part; returns the new admission
// bound: key and row of the KP-th entry once KP are held, else (+inf, NONE32).
template <int KP, int CAP>
__device__ __forceinline__ void knn_compact(const GraphView& gv, uint32_t* cnt, uint2* buf, uint32_t qi, float& thr, uint32_t& thr_row) {
    const int lane = lane_id();
    const uint32_t n = __builtin_amdgcn_readfirstlane(cnt[qi]);
    uint2 e = make_uint2(0x7f800000u, 0xFFFFFFFFu);
    if ((uint32_t)lane < n) e = buf[qi * CAP + lane];
    const float k = __uint_as_float(e.x);
    uint32_t rank = 0, same = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const float ki = readlane_f(k, (int)i);
        rank += ki < k ? 1u : 0u; same += ki == k ? 1u : 0u;
    }
    if (__ballot((uint32_t)lane < n && same > 1u)) {           // equal keys: order those by id
        uint64_t id = ~0ull;
        if ((uint32_t)lane < n) id = gv.ids[gv.rowmap ? gv.rowmap[e.y] : e.y];
        const uint32_t id_lo = (uint32_t)id, id_hi = (uint32_t)(id >> 32);
        for (uint32_t i = 0; i < n; ++i) {
            const float ki = readlane_f(k, (int)i);
            const uint64_t ii = ((uint64_t)readlane_u(id_hi, (int)i) << 32) | readlane_u(id_lo, (int)i);
            rank += (ki == k && ii < id) ? 1u : 0u;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if ((uint32_t)lane < n && rank < KP) buf[qi * CAP + rank] = e;
    const uint32_t kept = n < KP ? n : KP;
    if (lane == 0) cnt[qi] = kept;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint64_t last = __ballot((uint32_t)lane < n && rank == KP - 1);
    thr = INF_F; thr_row = NONE32;
    if (last) {
        const int l = __ffsll((unsigned long long)last) - 1;
        thr = readlane_f(k, l);
        thr_row = readlane_u(e.y, l);
    }
}
// Whether row `row`, whose key equals the admission bound, precedes the bound's row by id.
__device__ __forceinline__ bool knn_tie_admits(const GraphView& gv, uint32_t row, uint32_t thr_row) {
    if (thr_row == NONE32) return false;
    return gv.ids[gv.rowmap ? gv.rowmap[row] : row] < gv.ids[gv.rowmap ? gv.rowmap[thr_row] : thr_row];
}

// Epilogue of one 32-row tile: lane (j, h) holds the ranking keys of query j of its wave against rows
// row0 + (v / 4) * 8 + h * 4 + v % 4, v = 0..15.  Keys under the query's admission bound go to its candidate buffer.
// On average about one key of a tile passes somewhere in the wave, so this path runs for most tiles and is kept
// short: one compare per key, one LDS atomic per lane that has keys to add; keys equal to the bound (repeated
// vectors) and buffers that fill up before all keys of the tile are in take the branches that are rarely entered.
template <int KP>
__device__ __forceinline__ void knn_admit_tile(const GraphView& gv, const float (&key)[16], uint32_t row0, bool qvalid, uint32_t qbase,
                                               uint32_t* cnt, uint2* buf, float& thr, uint32_t& thr_row) {
    constexpr uint32_t CAP = KnnBuf<KP>::CAP, TRIG = KnnBuf<KP>::TRIG;
    const int lane = lane_id(), j = lane & 31, h = lane >> 5;
    const uint32_t ql = qbase + j;                             // this lane's query within the block; qbase: the wave's first
    // fp32 MFMAs and vector instructions share a SIMD's issue slot (tools/mfma_interleave.hip), so every instruction
    // here is paid in full: the common case — nothing of the tile passes anywhere in the wave — costs the minimum of
    // the 16 keys and one compare
    float mn = key[0];
#pragma unroll
    for (int v = 1; v < 16; ++v) mn = __builtin_fminf(mn, key[v]);
    if (!__ballot(qvalid && mn <= thr)) return;
    // The rest holds up the block's next barrier, and while the SIMD's other wave streams MFMAs this wave's vector
    // instructions would only issue in its gaps: ask for the issue slots until the keys are in.
    __builtin_amdgcn_s_setprio(3);
    uint32_t m = 0, eq = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) { m |= (key[v] < thr ? 1u : 0u) << v; eq |= (key[v] == thr ? 1u : 0u) << v; }
    if (!qvalid) { m = 0; eq = 0; }
    if (thr_row == NONE32) eq = 0;                             // no bound yet (+inf): nothing ties with it
    if (__ballot(eq != 0)) {                                   // a key equal to the bound passes only ahead of the bound's row by id
#pragma unroll
        for (int v = 0; v < 16; ++v)
            if (((eq >> v) & 1u) && knn_tie_admits(gv, row0 + (v / 4) * 8 + h * 4 + (v % 4), thr_row)) m |= 1u << v;
    }
    while (__ballot(m != 0)) {                                 // one pass, unless a buffer fills before all keys of the tile are in
        uint32_t held = 0;
        if (m != 0) {
            const uint32_t want = (uint32_t)__popc(m);
            uint32_t pos = atomicAdd(&cnt[ql], want);
            held = min(pos + want, CAP);
#pragma unroll
            for (int v = 0; v < 16; ++v)
                if (((m >> v) & 1u) && pos < CAP) {
                    buf[ql * CAP + pos++] = make_uint2(__float_as_uint(key[v]), row0 + (v / 4) * 8 + h * 4 + (v % 4));
                    m &= ~(1u << v);
                }
            if (m != 0) cnt[ql] = CAP;                         // places past the end were counted but not written
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // the later of a query's two lanes saw the earlier one's entries, so the larger `held` is the buffer's fill
        const uint64_t over = __ballot(held > TRIG);
        uint32_t need = (uint32_t)over | (uint32_t)(over >> 32);
        while (need) {
            const int jj = __ffs(need) - 1; need &= need - 1;
            float nt; uint32_t ni;
            knn_compact<KP, (int)CAP>(gv, cnt, buf, qbase + jj, nt, ni);
            if (j == jj && nt <= thr) { thr = nt; thr_row = ni; }      // (fewer than KP held: +inf, the bound stays)
        }
        if (m != 0) {                                          // keys still waiting face the renewed bound
#pragma unroll
            for (int v = 0; v < 16; ++v)
                if (((m >> v) & 1u) && !(key[v] < thr || (key[v] == thr && knn_tie_admits(gv, row0 + (v / 4) * 8 + h * 4 + (v % 4), thr_row)))) m &= ~(1u << v);
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

// scale / bias turn a dot product into the ranking key: key = dot * scale[row] + bias[row].
// centre[d] = mean of `sample` evenly spaced rows for the Euclidean metric, 0 otherwise (kq entries).  The bf16
// kernels contract q - centre with x - centre: Euclidean keys do not change with a common shift, but the products
// they are built from shrink from the rows' distance to the origin to their spread, and the split error with them.
__global__ void __launch_bounds__(256) knn_centre_kernel(GraphView gv, float* __restrict__ centre, uint32_t kq, uint32_t sample) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= kq) return;
    float s = 0.f;
    if (gv.metric == VDB_EUCLIDEAN && d < gv.dim && sample > 0) {
        const uint32_t step = gv.n / sample;
        for (uint32_t i = 0; i < sample; ++i) {
            const uint32_t r = i * step;
            s += gv.vecs[(size_t)(gv.rowmap ? gv.rowmap[r] : r) * gv.stride + d];
        }
        s /= (float)sample;
    }
    centre[d] = s;
}

__global__ void __launch_bounds__(256) knn_prepare_kernel(GraphView gv, const float* __restrict__ centre, float* __restrict__ scale, float* __restrict__ bias,
                                                         uint32_t* __restrict__ max_norm2) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = t >> 3, part = t & 7;                   // eight lanes share a row, 16 bytes each per step
    const bool in = r < gv.n;
    const uint32_t slot = in ? (gv.rowmap ? gv.rowmap[r] : r) : 0u;
    const float4* row = reinterpret_cast<const float4*>(gv.vecs + (size_t)slot * gv.stride);
    float s = 0.f;
    for (uint32_t c = part; c < (gv.stride >> 2); c += 8) {
        float4 v = row[c];
        if (centre) { const float4 m = reinterpret_cast<const float4*>(centre)[c]; v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w; }
        s = __builtin_fmaf(v.x, v.x, s); s = __builtin_fmaf(v.y, v.y, s); s = __builtin_fmaf(v.z, v.z, s); s = __builtin_fmaf(v.w, v.w, s);
    }
    s = group_sum<8>(s);
    const bool dead = !in || (gv.deleted && bit_test(gv.deleted, slot));
    // largest squared norm of the block's live rows, one atomic per block (s >= 0: its bit pattern orders like the value)
    __shared__ uint32_t wmax[4];
    uint32_t mx = dead ? 0u : __float_as_uint(s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(max_norm2, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
    if (!in || part != 0) return;
    float sc, bi = 0.f;
    if (gv.metric == VDB_EUCLIDEAN) { sc = -2.f; bi = s; }
    else if (gv.metric == VDB_DOT) sc = -1.f;
    else sc = s > 0.f ? -1.0f / __builtin_sqrtf(s) : 0.f;
    if (dead) { sc = 0.f; bi = INF_F; }
    scale[r] = sc; bias[r] = bi;
}

// grid (n_chunks, ceil(nq / MQ)), 256 threads.  KH = k-steps of the 32x32x2 MFMA = padded row / 2;
// k-step s of lane half h multiplies dimension h*KH + s, so a lane reads KH contiguous floats.
template <int KH, int KNN_KP>
__global__ void __launch_bounds__(256) knn_mfma_kernel(GraphView gv, const float* __restrict__ queries, uint32_t nq,
                                                      const float* __restrict__ scale, const float* __restrict__ bias,
                                                      uint32_t* __restrict__ out_slots, float* __restrict__ out_thr, uint32_t n_chunks) {
    constexpr int K = 2 * KH, LDA = K + 4;                     // padded tile row: conflict-free 16-byte reads
    constexpr int KNN_CAP = KnnBuf<KNN_KP>::CAP;
    // 32-row tiles written to LDS per pair of block barriers.  Two (64-row steps, half the barriers) measured the same
    // 28.4 ms per 10k x 1M x 128 batch as one: the barriers are not where this kernel waits.
    constexpr int NS = 1;
    constexpr int ROWS = NS * MT;
    constexpr int VPT = ROWS * K / 4 / 256;                    // float4 per thread per step (256 threads)
    constexpr int TPR = 256 / ROWS;                            // threads per tile row
    constexpr int CPT = K / TPR;                               // floats per thread: a run of CPT / 4 float4
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* At = reinterpret_cast<float*>(smem);                                  // ROWS x LDA
    float* sc_t = At + ROWS * LDA;                                               // ROWS
    float* bi_t = sc_t + ROWS;                                                   // ROWS
    uint32_t* cnt = reinterpret_cast<uint32_t*>(bi_t + ROWS);                    // MQ
    uint2* buf = reinterpret_cast<uint2*>(cnt + MQ);                             // MQ x KNN_CAP {key bits, row}

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t stride = gv.stride, dim = gv.dim;
    const uint32_t q = blockIdx.y * MQ + wave * 32 + j;        // this lane's query (two lanes per query)

    // B operand: the whole query stays in registers
    float b[KH];
#pragma unroll
    for (int s = 0; s < KH; ++s) {
        const uint32_t d = h * KH + s;
        b[s] = (q < nq && d < dim) ? queries[(size_t)q * dim + d] : 0.f;
    }
    if (tid < MQ) cnt[tid] = 0;
    float thr = INF_F; uint32_t thr_row = NONE32;  // admission bound of this lane's query: key and row of its KP-th candidate

    const uint32_t per = (((gv.n + n_chunks - 1) / n_chunks) + ROWS - 1) / ROWS * ROWS;
    const uint32_t r_begin = blockIdx.x * per, r_end = min(gv.n, r_begin + per);
    // loader: thread t brings columns [part * CPT, + CPT) of step row t / TPR
    const int lrow = tid / TPR, lpart = tid % TPR;
    float4 stage[VPT];
    float st_sc = 0.f, st_bi = INF_F;
    auto fetch = [&](uint32_t step_base) {
        const uint32_t r = step_base + lrow;
        const bool in = r < r_end;
        const uint32_t rp = in ? r : r_begin;
        const float* row = gv.vecs + (size_t)(gv.rowmap ? gv.rowmap[rp] : rp) * stride;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const uint32_t c = lpart * CPT + v * 4;
            stage[v] = (in && c < stride) ? *reinterpret_cast<const float4*>(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < ROWS) { const uint32_t rr = step_base + tid; st_sc = rr < r_end ? scale[rr] : 0.f; st_bi = rr < r_end ? bias[rr] : INF_F; }
    };
    auto stash = [&]() {
#pragma unroll
        for (int v = 0; v < VPT; ++v) *reinterpret_cast<float4*>(At + lrow * LDA + lpart * CPT + v * 4) = stage[v];
        if (tid < ROWS) { sc_t[tid] = st_sc; bi_t[tid] = st_bi; }
    };

    if (r_begin < r_end) fetch(r_begin);
    for (uint32_t tb = r_begin; tb < r_end; tb += ROWS) {
        __syncthreads();                               // the previous step's tiles have been consumed
        stash();
        __syncthreads();
        if (tb + ROWS < r_end) fetch(tb + ROWS);       // in flight during the contraction
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) {
            if (sub > 0 && tb + sub * MT >= r_end) break;
            floatx16 acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
            const float* arow = At + (sub * MT + j) * LDA + h * KH;
            float4 afrag[KH / 4];                 // the lane's whole slice of the tile: reads run ahead of the MFMAs
#pragma unroll
            for (int c = 0; c < KH / 4; ++c) afrag[c] = *reinterpret_cast<const float4*>(arow + 4 * c);
#pragma unroll
            for (int c = 0; c < KH / 4; ++c) {
                const float4 a4 = afrag[c];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b[4 * c + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b[4 * c + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b[4 * c + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b[4 * c + 3], acc, 0, 0, 0);
            }
            // epilogue: acc[v] = q_j . x_i with i = (v/4)*8 + h*4 + v%4
            float key[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 s4 = *reinterpret_cast<const float4*>(sc_t + sub * MT + g4 * 8 + h * 4);
                const float4 b4 = *reinterpret_cast<const float4*>(bi_t + sub * MT + g4 * 8 + h * 4);
                key[g4 * 4 + 0] = __builtin_fmaf(acc[g4 * 4 + 0], s4.x, b4.x);
                key[g4 * 4 + 1] = __builtin_fmaf(acc[g4 * 4 + 1], s4.y, b4.y);
                key[g4 * 4 + 2] = __builtin_fmaf(acc[g4 * 4 + 2], s4.z, b4.z);
                key[g4 * 4 + 3] = __builtin_fmaf(acc[g4 * 4 + 3], s4.w, b4.w);
            }
            knn_admit_tile<KNN_KP>(gv, key, tb + sub * MT, q < nq, (uint32_t)wave * 32, cnt, buf, thr, thr_row);
        }
    }
    // hand over: KNN_KP slots per (query, chunk), NONE32 marks unused entries
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int jj = 0; jj < 32; ++jj) {
        const uint32_t qi = wave * 32 + jj, qg = blockIdx.y * MQ + qi;
        if (qg >= nq) break;
        // the bound every row this chunk dropped for the query was at or above: the KP-th key held at the end, +inf if
        // fewer than KP rows were ever held (then nothing was dropped)
        float fin = readlane_f(thr, jj);
        if (cnt[qi] > KNN_KP) { uint32_t ni; knn_compact<KNN_KP, KNN_CAP>(gv, cnt, buf, qi, fin, ni); }
        const uint32_t n = cnt[qi];
        if (lane == 0) out_thr[(size_t)qg * n_chunks + blockIdx.x] = fin;
        if (lane < KNN_KP) out_slots[((size_t)qg * n_chunks + blockIdx.x) * KNN_KP + lane] = (uint32_t)lane < n ? buf[qi * KNN_CAP + lane].y : NONE32;
    }
}

// ---- candidate pass on the bf16 matrix pipe -------------------------------------------------------------------
// Every fp32 value is split into a bf16 high part and a bf16 remainder, x = xh + xl + O(2^-17 |x|), and a product
// q.x is taken as qh.xh + qh.xl + ql.xh (v_mfma_f32_32x32x16_bf16, fp32 accumulation): the dropped ql.xl and the
// split residuals leave an error below 2^-14 of sum |q_i x_i| in the worst case, about 2^-17 typically, where three
// bf16 MFMAs take 98 clocks per 16 dimensions against 512 for the eight fp32 MFMAs they replace.  The keys only
// select candidates: the strict re-rank scores whatever is kept, so an error moves the boundary of a chunk's KP
// candidates, not a result — a true neighbour is lost only if more than KP - top_k rows of one chunk lie within that
// error of it.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bf16_split(float x, __bf16& h, __bf16& l) { h = (__bf16)x; l = (__bf16)(x - (float)h); }

// queries minus the centre -> two planes (high parts, remainders) of nq x kq bf16, zero beyond dim
__global__ void __launch_bounds__(256) knn_split_queries_kernel(const float* __restrict__ queries, const float* __restrict__ centre, uint32_t nq, uint32_t dim, uint32_t kq,
                                                               __bf16* __restrict__ qh, __bf16* __restrict__ ql) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nq * kq) return;
    const uint32_t q = (uint32_t)(i / kq), d = (uint32_t)(i % kq);
    const float x = d < dim ? queries[(size_t)q * dim + d] - centre[d] : 0.f;
    __bf16 h, l; bf16_split(x, h, l);
    qh[i] = h; ql[i] = l;
}

// grid (n_chunks, ceil(nq / (32 WPB))), WPB waves per block.  KS = 16-dimension steps of the row (padded row = 16 KS floats).  Lane
// (j, h) of a wave holds, per step, dimensions 16 s + 8 h .. + 8 of query j (B operand, registers) and of tile row j
// (A operand, read from the tile's two bf16 planes in LDS); the accumulator layout is that of the fp32 32x32 MFMA.
// Waves per block of the bf16 kernel (32 queries each): eight share a tile's loads, split and barriers; measured per
// 10k x 1M batch — 128-float rows: 14.2 ms with eight, 13.9 with four (top-20: 19.3 against 16.6); 256-float rows: 24.4
// against 37 ms (four waves with 64 query registers each leave room for one wave per SIMD).  Eight for 256, four below.
// Tile buffers of the bf16 kernel: 2 (the next tile is written while this one is contracted, one block barrier per
// tile) measured the same 12.55 ms per 10k x 1M x 128 batch as 1 (two barriers per tile) and takes 17 KiB more LDS.
#ifndef KNN_BF16_BUFFERS
#define KNN_BF16_BUFFERS 1
#endif
template <int KS, int KNN_KP, int WPB>
__global__ void __launch_bounds__(64 * WPB) knn_bf16_kernel(GraphView gv, const __bf16* __restrict__ qh, const __bf16* __restrict__ ql, uint32_t nq,
                                                      const float* __restrict__ centre, const float* __restrict__ scale, const float* __restrict__ bias,
                                                      uint32_t* __restrict__ out_slots, float* __restrict__ out_thr, uint32_t n_chunks, const float* __restrict__ init_thr) {
    constexpr int K = 16 * KS, LDB = K + 8;                    // bf16 per tile row, padded by 16 bytes: conflict-free 16-byte reads
    constexpr int KNN_CAP = KnnBuf<KNN_KP>::CAP;
    constexpr int MQB = 32 * WPB;                              // queries per block: 32 per wave
    constexpr int TPR = 2 * WPB;                               // loading threads per tile row (all threads load)
    constexpr int CPT = K / TPR;                               // floats per loading thread
    constexpr int VPT = CPT / 4;                               // float4 per thread per tile
    static_assert(VPT >= 2 && VPT % 2 == 0, "a loading thread writes 16 bytes per plane at a time");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NB = KNN_BF16_BUFFERS;                       // tile buffers
    __bf16* Ah = reinterpret_cast<__bf16*>(smem);                                // NB x MT x LDB high parts
    __bf16* Al = Ah + NB * MT * LDB;                                             // NB x MT x LDB remainders
    float* sc_t = reinterpret_cast<float*>(Al + NB * MT * LDB);                  // NB x MT
    float* bi_t = sc_t + NB * MT;                                                // NB x MT
    uint32_t* cnt = reinterpret_cast<uint32_t*>(bi_t + NB * MT);                 // MQB
    uint2* buf = reinterpret_cast<uint2*>(cnt + MQB);                            // MQB x KNN_CAP {key bits, row}

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t stride = gv.stride;
    const uint32_t q = blockIdx.y * MQB + wave * 32 + j;       // this lane's query (two lanes per query)

    bf16x8 bh[KS], bl[KS];                                     // B operand: the whole query stays in registers
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const size_t off = (size_t)(q < nq ? q : 0) * K + 16 * s + 8 * h;
        bh[s] = *reinterpret_cast<const bf16x8*>(qh + off);
        bl[s] = *reinterpret_cast<const bf16x8*>(ql + off);
    }
    if (tid < MQB) cnt[tid] = 0;
    float thr = INF_F; uint32_t thr_row = NONE32;  // admission bound of this lane's query: key and row of its KP-th candidate
    if (init_thr && q < nq) thr = init_thr[q];     // a bound known beforehand (the graph's answer)

    const uint32_t per = (((gv.n + n_chunks - 1) / n_chunks) + MT - 1) / MT * MT;
    const uint32_t r_begin = blockIdx.x * per, r_end = min(gv.n, r_begin + per);
    // loader: thread t brings columns [part * CPT, + CPT) of tile row t / TPR
    const int lrow = tid / TPR, lpart = tid % TPR;
    float4 stage[VPT], mid[VPT];                               // mid: the centre's values of this thread's columns
#pragma unroll
    for (int v = 0; v < VPT; ++v) mid[v] = *reinterpret_cast<const float4*>(centre + lpart * CPT + v * 4);
    float st_sc = 0.f, st_bi = INF_F;
    auto fetch = [&](uint32_t tile_base) {
        const uint32_t r = tile_base + lrow;
        const bool in = r < r_end;
        const uint32_t rp = in ? r : r_begin;
        const float* row = gv.vecs + (size_t)(gv.rowmap ? gv.rowmap[rp] : rp) * stride;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const uint32_t c = lpart * CPT + v * 4;
            stage[v] = (in && c < stride) ? *reinterpret_cast<const float4*>(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < MT) { const uint32_t rr = tile_base + tid; st_sc = rr < r_end ? scale[rr] : 0.f; st_bi = rr < r_end ? bias[rr] : INF_F; }
    };
    auto stash = [&](uint32_t ib) {
#pragma unroll
        for (int v = 0; v < VPT; v += 2) {
            const float x[8] = {stage[v].x - mid[v].x, stage[v].y - mid[v].y, stage[v].z - mid[v].z, stage[v].w - mid[v].w,
                                stage[v + 1].x - mid[v + 1].x, stage[v + 1].y - mid[v + 1].y, stage[v + 1].z - mid[v + 1].z, stage[v + 1].w - mid[v + 1].w};
            bf16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) { __bf16 a, b; bf16_split(x[e], a, b); hi[e] = a; lo[e] = b; }
            const int at = (int)ib * MT * LDB + lrow * LDB + lpart * CPT + v * 4;
            *reinterpret_cast<bf16x8*>(Ah + at) = hi;
            *reinterpret_cast<bf16x8*>(Al + at) = lo;
        }
        if (tid < MT) { sc_t[ib * MT + tid] = st_sc; bi_t[ib * MT + tid] = st_bi; }
    };

    if (r_begin < r_end) fetch(r_begin);
    if (NB == 2) {
        if (r_begin < r_end) stash(0);
        __syncthreads();
        if (r_begin + MT < r_end) fetch(r_begin + MT);
    }
    uint32_t ib = 0;
    for (uint32_t tb = r_begin; tb < r_end; tb += MT) {
        if (NB == 1) {
            __syncthreads();                           // the previous tile has been consumed
            stash(0);
            __syncthreads();
            if (tb + MT < r_end) fetch(tb + MT);       // in flight during the contraction
        } else {
            // the other buffer was last read during the previous tile, which every wave left through the barrier below
            if (tb + MT < r_end) stash(ib ^ 1u);
            if (tb + 2 * MT < r_end) fetch(tb + 2 * MT);
        }
        floatx16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        const __bf16* arh = Ah + ib * (MT * LDB) + j * LDB + 8 * h;
        const __bf16* arl = Al + ib * (MT * LDB) + j * LDB + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(arh + 16 * s);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(arl + 16 * s);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[s], acc, 0, 0, 0);
        }
        // epilogue: acc[v] = q_j . x_i with i = (v/4)*8 + h*4 + v%4
        float key[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 s4 = *reinterpret_cast<const float4*>(sc_t + ib * MT + g4 * 8 + h * 4);
            const float4 b4 = *reinterpret_cast<const float4*>(bi_t + ib * MT + g4 * 8 + h * 4);
            key[g4 * 4 + 0] = __builtin_fmaf(acc[g4 * 4 + 0], s4.x, b4.x);
            key[g4 * 4 + 1] = __builtin_fmaf(acc[g4 * 4 + 1], s4.y, b4.y);
            key[g4 * 4 + 2] = __builtin_fmaf(acc[g4 * 4 + 2], s4.z, b4.z);
            key[g4 * 4 + 3] = __builtin_fmaf(acc[g4 * 4 + 3], s4.w, b4.w);
        }
        knn_admit_tile<KNN_KP>(gv, key, tb, q < nq, (uint32_t)wave * 32, cnt, buf, thr, thr_row);
        if (NB == 2) { __syncthreads(); ib ^= 1u; }    // this tile is consumed, the next one is complete
    }
    // hand over: KNN_KP slots per (query, chunk), NONE32 marks unused entries
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int jj = 0; jj < 32; ++jj) {
        const uint32_t qi = wave * 32 + jj, qg = blockIdx.y * MQB + qi;
        if (qg >= nq) break;
        // the bound every row this chunk dropped for the query was at or above: the KP-th key held at the end, +inf if
        // fewer than KP rows were ever held (then nothing was dropped)
        float fin = readlane_f(thr, jj);
        if (cnt[qi] > KNN_KP) { uint32_t ni; knn_compact<KNN_KP, KNN_CAP>(gv, cnt, buf, qi, fin, ni); }
        const uint32_t n = cnt[qi];
        if (lane == 0) out_thr[(size_t)qg * n_chunks + blockIdx.x] = fin;
        if (lane < KNN_KP) out_slots[((size_t)qg * n_chunks + blockIdx.x) * KNN_KP + lane] = (uint32_t)lane < n ? buf[qi * KNN_CAP + lane].y : NONE32;
    }
}

// Rows longer than 256 floats: the contraction runs over the row in chunks of KC floats with the
// accumulators kept in registers.  A block handles LT = 128 base rows (four 32x32 tiles per wave,
// i.e. four independent MFMA chains) against MQ = 128 queries; per chunk the A slice (LT x KC) goes
// through LDS (double buffered, one barrier per chunk) and every lane reads its 16 query values of
// the chunk straight from global memory (the query block stays in L2).  Same epilogue and hand-over
// as knn_mfma_kernel.  grid (n_chunks, ceil(nq / MQ)), 256 threads.
constexpr int LT = 128, KC = 32, LDK = KC + 4;
template <int KNN_KP>
__global__ void __launch_bounds__(256) knn_mfma_long_kernel(GraphView gv, const float* __restrict__ queries, uint32_t nq,
                                                           const float* __restrict__ scale, const float* __restrict__ bias,
                                                           uint32_t* __restrict__ out_slots, float* __restrict__ out_thr, uint32_t n_chunks) {
    constexpr int KNN_CAP = KnnBuf<KNN_KP>::CAP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* At = reinterpret_cast<float*>(smem);                                  // 2 x LT x LDK
    float* sc_t = At + 2 * LT * LDK;                                             // LT
    float* bi_t = sc_t + LT;                                                     // LT
    uint32_t* cnt = reinterpret_cast<uint32_t*>(bi_t + LT);                      // MQ
    uint2* buf = reinterpret_cast<uint2*>(cnt + MQ);                             // MQ x KNN_CAP {key bits, row}

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t stride = gv.stride, dim = gv.dim;
    const uint32_t q = blockIdx.y * MQ + wave * 32 + j;
    const bool qin = q < nq;
    const float* qrow = queries + (size_t)(qin ? q : 0) * dim;
    const bool qvec = (dim & 3u) == 0 && (reinterpret_cast<uintptr_t>(queries) & 15u) == 0;
    const uint32_t nkc = (stride + KC - 1) / KC;

    if (tid < MQ) cnt[tid] = 0;
    float thr = INF_F; uint32_t thr_row = NONE32;  // admission bound of this lane's query: key and row of its KP-th candidate
    const uint32_t per = (((gv.n + n_chunks - 1) / n_chunks) + LT - 1) / LT * LT;
    const uint32_t r_begin = blockIdx.x * per, r_end = min(gv.n, r_begin + per);
    const int lrow = tid >> 1, lhalf = tid & 1;          // loader: thread t brings 16 floats of tile row t / 2


    for (uint32_t tb = r_begin; tb < r_end; tb += LT) {
        float4 stage[4]; float bq[16], bn[16];
        auto fetch_a = [&](uint32_t kc) {
            const uint32_t r = tb + lrow;
            const bool in = r < r_end;
            const uint32_t rp = in ? r : r_begin;
            const float* row = gv.vecs + (size_t)(gv.rowmap ? gv.rowmap[rp] : rp) * stride;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const uint32_t c = kc * KC + lhalf * 16 + v * 4;
                stage[v] = (in && c < stride) ? *reinterpret_cast<const float4*>(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto fetch_b = [&](uint32_t kc, float* dst) {
            const uint32_t d0 = kc * KC + h * 16;
            if (qvec) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const uint32_t d = d0 + v * 4;
                    const float4 x = (qin && d < dim) ? *reinterpret_cast<const float4*>(qrow + d) : make_float4(0.f, 0.f, 0.f, 0.f);
                    dst[4 * v] = x.x; dst[4 * v + 1] = x.y; dst[4 * v + 2] = x.z; dst[4 * v + 3] = x.w;
                }
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v) dst[v] = (qin && d0 + v < dim) ? qrow[d0 + v] : 0.f;
            }
        };
        floatx16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
        __syncthreads();                                   // the previous tile's scale / bias and A slices are consumed
        if (tid < LT) { const uint32_t rr = tb + tid; sc_t[tid] = rr < r_end ? scale[rr] : 0.f; bi_t[tid] = rr < r_end ? bias[rr] : INF_F; }
        fetch_a(0); fetch_b(0, bq);
        for (uint32_t kc = 0; kc < nkc; ++kc) {
            float* Ab = At + (kc & 1u) * (LT * LDK);
#pragma unroll
            for (int v = 0; v < 4; ++v) *reinterpret_cast<float4*>(Ab + lrow * LDK + lhalf * 16 + v * 4) = stage[v];
            __syncthreads();
            if (kc + 1 < nkc) { fetch_a(kc + 1); fetch_b(kc + 1, bn); }     // in flight during the contraction
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float* arow = Ab + (t * 32 + j) * LDK + h * 16;
                float4 af[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) af[c] = *reinterpret_cast<const float4*>(arow + 4 * c);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c].x, bq[4 * c + 0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c].y, bq[4 * c + 1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c].z, bq[4 * c + 2], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c].w, bq[4 * c + 3], acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) bq[v] = bn[v];
        }
        // epilogue, one 32-row sub-tile at a time: acc[t][v] = q_j . x_i with i = t*32 + (v/4)*8 + h*4 + v%4
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float key[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 s4 = *reinterpret_cast<const float4*>(sc_t + t * 32 + g4 * 8 + h * 4);
                const float4 b4 = *reinterpret_cast<const float4*>(bi_t + t * 32 + g4 * 8 + h * 4);
                key[g4 * 4 + 0] = __builtin_fmaf(acc[t][g4 * 4 + 0], s4.x, b4.x);
                key[g4 * 4 + 1] = __builtin_fmaf(acc[t][g4 * 4 + 1], s4.y, b4.y);
                key[g4 * 4 + 2] = __builtin_fmaf(acc[t][g4 * 4 + 2], s4.z, b4.z);
                key[g4 * 4 + 3] = __builtin_fmaf(acc[t][g4 * 4 + 3], s4.w, b4.w);
            }
            knn_admit_tile<KNN_KP>(gv, key, tb + t * 32, qin, (uint32_t)wave * 32, cnt, buf, thr, thr_row);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int jj = 0; jj < 32; ++jj) {
        const uint32_t qi = wave * 32 + jj, qg = blockIdx.y * MQ + qi;
        if (qg >= nq) break;
        // the bound every row this chunk dropped for the query was at or above: the KP-th key held at the end, +inf if
        // fewer than KP rows were ever held (then nothing was dropped)
        float fin = readlane_f(thr, jj);
        if (cnt[qi] > KNN_KP) { uint32_t ni; knn_compact<KNN_KP, KNN_CAP>(gv, cnt, buf, qi, fin, ni); }
        const uint32_t n = cnt[qi];
        if (lane == 0) out_thr[(size_t)qg * n_chunks + blockIdx.x] = fin;
        if (lane < KNN_KP) out_slots[((size_t)qg * n_chunks + blockIdx.x) * KNN_KP + lane] = (uint32_t)lane < n ? buf[qi * KNN_CAP + lane].y : NONE32;
    }
}

// Rows longer than 256 floats on the bf16 matrix pipe: knn_mfma_long_kernel's shape (128 rows x 128 queries per block,
// the row contracted in chunks of 32 floats, accumulators of four 32-row sub-tiles in registers, A slices double
// buffered in LDS with one block barrier per chunk) with the arithmetic of knn_bf16_kernel: the loader splits the
// A slice into two bf16 planes, every lane reads its query's two 8-value runs per 16-dimension step from the
// pre-split planes in global memory (they stay in L2), three MFMAs per step and sub-tile.  One A slice in LDS and two
// block barriers per chunk, so that two blocks fit a CU (a second slice buffer leaves room for one: 119 against 107 ms
// per 10k x 1M x 768 batch); the centre sits in LDS.
constexpr int KC2 = 32, LDK2 = KC2 + 8;
template <int KNN_KP>
__global__ void __launch_bounds__(256, 2) knn_bf16_long_kernel(GraphView gv, const __bf16* __restrict__ qh, const __bf16* __restrict__ ql, uint32_t kq, uint32_t nq,
                                                           const float* __restrict__ centre, const float* __restrict__ scale, const float* __restrict__ bias,
                                                           uint32_t* __restrict__ out_slots, float* __restrict__ out_thr, uint32_t n_chunks, const float* __restrict__ init_thr) {
    constexpr int KNN_CAP = KnnBuf<KNN_KP>::CAP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Ah = reinterpret_cast<__bf16*>(smem);                                // LT x LDK2 high parts
    __bf16* Al = Ah + LT * LDK2;                                                 // LT x LDK2 remainders
    float* sc_t = reinterpret_cast<float*>(Al + LT * LDK2);                      // LT
    float* bi_t = sc_t + LT;                                                     // LT
    uint32_t* cnt = reinterpret_cast<uint32_t*>(bi_t + LT);                      // MQ
    uint2* buf = reinterpret_cast<uint2*>(cnt + MQ);                             // MQ x KNN_CAP {key bits, row}
    float* mid_l = reinterpret_cast<float*>(buf + MQ * KNN_CAP);                 // kq: the centre

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t stride = gv.stride;
    const uint32_t q = blockIdx.y * MQ + wave * 32 + j;
    const bool qin = q < nq;
    for (uint32_t i = tid; i < kq; i += 256) mid_l[i] = centre[i];
    const __bf16* qhr = qh + (size_t)(qin ? q : 0) * kq + 8 * h;
    const __bf16* qlr = ql + (size_t)(qin ? q : 0) * kq + 8 * h;
    const uint32_t nkc = kq / KC2;

    if (tid < MQ) cnt[tid] = 0;
    float thr = INF_F; uint32_t thr_row = NONE32;  // admission bound of this lane's query: key and row of its KP-th candidate
    if (init_thr && qin) thr = init_thr[q];        // a bound known beforehand (the graph's answer)
    const uint32_t per = (((gv.n + n_chunks - 1) / n_chunks) + LT - 1) / LT * LT;
    const uint32_t r_begin = blockIdx.x * per, r_end = min(gv.n, r_begin + per);
    const int lrow = tid >> 1, lhalf = tid & 1;          // loader: thread t brings 16 floats of tile row t / 2

    for (uint32_t tb = r_begin; tb < r_end; tb += LT) {
        // the A slices of the next two chunks are in flight (st0: even chunks, st1: odd ones) while a chunk is contracted:
        // with one chunk ahead an iteration lasts as long as a memory round trip, whatever it computes
        float4 st0[4], st1[4]; bf16x8 bhq[2], blq[2], bhn[2], bln[2];
        const uint32_t lr = tb + lrow;
        const bool lin = lr < r_end;
        const uint32_t lrp = lin ? lr : r_begin;
        const float* lsrc = gv.vecs + (size_t)(gv.rowmap ? gv.rowmap[lrp] : lrp) * stride;      // one look-up per tile, ahead of all its loads
        auto fetch_a = [&](uint32_t kc, float4 (&st)[4]) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const uint32_t c = kc * KC2 + lhalf * 16 + v * 4;
                st[v] = (lin && c < stride) ? *reinterpret_cast<const float4*>(lsrc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto fetch_b = [&](uint32_t kc, bf16x8 (&dh)[2], bf16x8 (&dl)[2]) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                dh[s] = *reinterpret_cast<const bf16x8*>(qhr + kc * KC2 + 16 * s);
                dl[s] = *reinterpret_cast<const bf16x8*>(qlr + kc * KC2 + 16 * s);
            }
        };
        floatx16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
        auto chunk = [&](uint32_t kc, float4 (&st)[4]) {
            if (kc + 1 < nkc) fetch_b(kc + 1, bhn, bln);   // ahead of the barriers: its registers were freed by the last copy
            if (kc > 0) __syncthreads();                   // the previous slice has been consumed (the first follows the tile's barrier)
#pragma unroll
            for (int v = 0; v < 4; v += 2) {
                const float4 m0 = *reinterpret_cast<const float4*>(mid_l + kc * KC2 + lhalf * 16 + v * 4);
                const float4 m1 = *reinterpret_cast<const float4*>(mid_l + kc * KC2 + lhalf * 16 + v * 4 + 4);
                const float x[8] = {st[v].x - m0.x, st[v].y - m0.y, st[v].z - m0.z, st[v].w - m0.w,
                                    st[v + 1].x - m1.x, st[v + 1].y - m1.y, st[v + 1].z - m1.z, st[v + 1].w - m1.w};
                bf16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) { __bf16 a, b; bf16_split(x[e], a, b); hi[e] = a; lo[e] = b; }
                const int at = lrow * LDK2 + lhalf * 16 + v * 4;
                *reinterpret_cast<bf16x8*>(Ah + at) = hi;
                *reinterpret_cast<bf16x8*>(Al + at) = lo;
            }
            __syncthreads();
            if (kc + 2 < nkc) fetch_a(kc + 2, st);         // into the registers the split has just emptied
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const __bf16* arh = Ah + (t * 32 + j) * LDK2 + 8 * h;
                const __bf16* arl = Al + (t * 32 + j) * LDK2 + 8 * h;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(arh + 16 * s);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(arl + 16 * s);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhq[s], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhq[s], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blq[s], acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) { bhq[s] = bhn[s]; blq[s] = bln[s]; }
        };
        __syncthreads();                                   // the previous tile's scale / bias and A slices are consumed
        if (tid < LT) { const uint32_t rr = tb + tid; sc_t[tid] = rr < r_end ? scale[rr] : 0.f; bi_t[tid] = rr < r_end ? bias[rr] : INF_F; }
        fetch_a(0, st0); fetch_b(0, bhq, blq);
        if (nkc > 1) fetch_a(1, st1);
        for (uint32_t kc = 0; kc < nkc; kc += 2) {         // nkc is even (the queries' planes are padded to 64 dimensions)
            chunk(kc, st0);
            chunk(kc + 1, st1);
        }
        // epilogue, one 32-row sub-tile at a time: acc[t][v] = q_j . x_i with i = t*32 + (v/4)*8 + h*4 + v%4
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float key[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 s4 = *reinterpret_cast<const float4*>(sc_t + t * 32 + g4 * 8 + h * 4);
                const float4 b4 = *reinterpret_cast<const float4*>(bi_t + t * 32 + g4 * 8 + h * 4);
                key[g4 * 4 + 0] = __builtin_fmaf(acc[t][g4 * 4 + 0], s4.x, b4.x);
                key[g4 * 4 + 1] = __builtin_fmaf(acc[t][g4 * 4 + 1], s4.y, b4.y);
                key[g4 * 4 + 2] = __builtin_fmaf(acc[t][g4 * 4 + 2], s4.z, b4.z);
                key[g4 * 4 + 3] = __builtin_fmaf(acc[t][g4 * 4 + 3], s4.w, b4.w);
            }
            knn_admit_tile<KNN_KP>(gv, key, tb + t * 32, qin, (uint32_t)wave * 32, cnt, buf, thr, thr_row);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int jj = 0; jj < 32; ++jj) {
        const uint32_t qi = wave * 32 + jj, qg = blockIdx.y * MQ + qi;
        if (qg >= nq) break;
        // the bound every row this chunk dropped for the query was at or above: the KP-th key held at the end, +inf if
        // fewer than KP rows were ever held (then nothing was dropped)
        float fin = readlane_f(thr, jj);
        if (cnt[qi] > KNN_KP) { uint32_t ni; knn_compact<KNN_KP, KNN_CAP>(gv, cnt, buf, qi, fin, ni); }
        const uint32_t n = cnt[qi];
        if (lane == 0) out_thr[(size_t)qg * n_chunks + blockIdx.x] = fin;
        if (lane < KNN_KP) out_slots[((size_t)qg * n_chunks + blockIdx.x) * KNN_KP + lane] = (uint32_t)lane < n ? buf[qi * KNN_CAP + lane].y : NONE32;
    }
}

// A bound every query can start its candidate pass from when `out` already holds an approximate answer (the graph's):
// top_k actual rows, so the true k-th key is at or below their k-th key.  In the candidate pass's key space, widened by
// three times the error budget of the verification (the rows found must pass under it by their approximate keys, and the
// verification must still see a gap of one budget above the strict k-th key); +inf where the answer holds fewer than top_k
// rows.  ORACLE (diagnostic VDB_KNN_ORACLE_BOUND): the k-th key widened by 1e-3 only.
template <bool ORACLE>
__global__ void __launch_bounds__(64) knn_bound_kernel(GraphView gv, const float* __restrict__ queries, const float* __restrict__ centre, uint32_t top_k,
                                                      const uint32_t* __restrict__ max_norm2,
                                                      const vdb_scored_point* __restrict__ out, const uint32_t* __restrict__ out_n, float* __restrict__ init_thr) {
    const uint32_t q = blockIdx.x; const int lane = lane_id();
    float qn2 = 0.f;
    for (uint32_t i = lane; i < gv.dim; i += WAVE) { const float d = queries[(size_t)q * gv.dim + i] - (gv.metric == VDB_EUCLIDEAN ? centre[i] : 0.f); qn2 = __builtin_fmaf(d, d, qn2); }
    qn2 = group_sum<64>(qn2);
    if (lane != 0) return;
    float b = INF_F;
    if (out_n[q] >= top_k) {
        const float kth = out[(size_t)q * top_k + top_k - 1].sort_key;
        const float E = 1.220703125e-4f + (float)gv.dim * 1.1920929e-7f + 1e-5f;
        const float xm = __builtin_sqrtf(__uint_as_float(*max_norm2)), qn = __builtin_sqrtf(qn2);
        if (ORACLE) {
            if (gv.metric == VDB_EUCLIDEAN) b = kth * kth * 1.001f - qn2;
            else if (gv.metric == VDB_DOT) b = kth + 1e-3f * __builtin_fabsf(kth);
            else b = (kth - 1.0f) * qn * (kth < 1.0f ? 0.999f : 1.001f);
        } else {
            if (gv.metric == VDB_EUCLIDEAN) b = kth * kth * 1.00003f - qn2 + 3.0f * E * (qn + xm) * (qn + xm);
            else if (gv.metric == VDB_DOT) b = kth + 3.0f * E * qn * xm;
            else b = (kth - 1.0f) * qn + 3.0f * E * qn;
        }
    }
    init_thr[q] = b;
}

// One wave per query: strict scores of the candidate slots, ranked by (sort_key, id).
struct TopKList {
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
};

// The candidate pass ranks rows by approximate keys; a row it dropped for (query, chunk) had an approximate key at or
// above that chunk's final bound thr.  With err = the most an approximate key and a strict key can differ (both against
// the real value: the bf16 split leaves at most 2^-14 of sum |q_i x_i| <= |q||x|, the strict fp32 sum dim * 2^-24 of it; the
// budget used is 2^-13 + dim * 2^-23), a dropped row can precede the strict k-th result only if thr - err <= k-th key.
// Where that cannot be excluded the query is flagged and knn_fallback_kernel scores every row for it.
//   dot:     approximate key = -q.x = sort key;                         err = E |q| max|x|
//   cosine:  approximate key = -q.x / |x|, sort key = 1 + that / |q|;    err = E (relative to 1)
//   L2:      approximate key = |x-c|^2 - 2 (q-c).(x-c) = dist^2 - |q-c|^2; err = E (|q-c| + max|x-c|)^2, compared in dist^2
__global__ void __launch_bounds__(64) knn_rerank_kernel(GraphView gv, const float* __restrict__ queries, uint32_t top_k,
                                                       const uint32_t* __restrict__ slots, uint32_t per_query,
                                                       const float* __restrict__ thr, uint32_t n_chunks, const float* __restrict__ centre,
                                                       const uint32_t* __restrict__ max_norm2, uint32_t* __restrict__ flags, uint32_t* __restrict__ flag_count,
                                                       vdb_scored_point* __restrict__ out, uint32_t* __restrict__ out_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id();
    const uint32_t q = blockIdx.x, dim = gv.dim, stride = gv.stride;
    float* ql = reinterpret_cast<float*>(smem);
    uint64_t* lid = reinterpret_cast<uint64_t*>(smem + (size_t)stride * 4);
    float* lkey = reinterpret_cast<float*>(lid + top_k);
    float* lraw = lkey + top_k;
    for (uint32_t i = lane; i < stride; i += WAVE) ql[i] = i < dim ? queries[(size_t)q * dim + i] : 0.f;
    wave_sync();
    float qn2 = 0.f;                                                          // |q - c|^2 (c = 0 unless Euclidean), for the verification below
    for (uint32_t i = lane; i < dim; i += WAVE) { const float d = ql[i] - (centre ? centre[i] : 0.f); qn2 = __builtin_fmaf(d, d, qn2); }
    qn2 = group_sum<64>(qn2);
    if (gv.metric == VDB_COSINE) {                                            // hnsw.rs:526-536, 752-764
        float s = -0.0f;
        for (uint32_t i = 0; i < dim; ++i) { float p = ql[i] * ql[i]; s = s + p; }
        const float norm = __builtin_sqrtf(s);
        wave_sync();
        if (norm != 0.0f) for (uint32_t i = lane; i < dim; i += WAVE) ql[i] = ql[i] / norm;
        wave_sync();
    }
    TopKList L{lkey, lid, lraw, 0, top_k};
    for (uint32_t b = 0; b < per_query; b += WAVE) {
        const uint32_t i = b + lane;
        uint32_t s = i < per_query ? slots[(size_t)q * per_query + i] : NONE32;      // a position of the scan
        const bool valid = s != NONE32;
        if (valid && gv.rowmap) s = gv.rowmap[s];
        float raw = 0.f; uint64_t idv = 0;
        if (valid) { raw = strict_pair(ql, gv.vecs + (size_t)s * stride, dim, gv.metric, true); idv = gv.ids[s]; }
        const float k = gv.metric == VDB_DOT ? -raw : raw;
        uint64_t am = ballot(valid && L.admits(k, idv));
        while (am) {
            const int jj = __ffsll((unsigned long long)am) - 1; am &= am - 1;
            const float kk = readlane_f(k, jj), rr = readlane_f(raw, jj);
            const uint64_t ii = ((uint64_t)readlane_u((uint32_t)(idv >> 32), jj) << 32) | readlane_u((uint32_t)idv, jj);
            if (!L.admits(kk, ii)) continue;
            L.insert(kk, ii, rr);
        }
    }
    for (uint32_t i = lane; i < L.n; i += WAVE) { vdb_scored_point sp; sp.id = lid[i]; sp.raw_score = lraw[i]; sp.sort_key = lkey[i]; out[(size_t)q * top_k + i] = sp; }
    if (lane == 0) out_n[q] = L.n;
    if (flags) {
        // smallest bound over the chunks (chunks that dropped nothing report +inf)
        float lo = INF_F;
        for (uint32_t c = lane; c < n_chunks; c += WAVE) lo = __builtin_fminf(lo, thr[(size_t)q * n_chunks + c]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) lo = __builtin_fminf(lo, __shfl_xor(lo, off));
        bool unsure = false;
        if (lo < INF_F) {
            if (L.n < top_k) unsure = true;                                   // rows were dropped although the list is not full
            else {
                const float kth = lkey[top_k - 1];
                const float E = 1.220703125e-4f + (float)dim * 1.1920929e-7f;
                const float xm = __builtin_sqrtf(__uint_as_float(*max_norm2)), qn = __builtin_sqrtf(qn2);
                if (gv.metric == VDB_DOT) unsure = !(lo - E * qn * xm > kth);
                else if (gv.metric == VDB_COSINE) unsure = !(qn > 0.f && 1.0f + lo / qn - E > kth);
                else unsure = !(lo + qn2 - E * (qn + xm) * (qn + xm) > kth * kth * (1.0f + 2.4e-7f));
            }
        }
        if (lane == 0 && unsure) flags[atomicAdd(flag_count, 1u)] = q;          // the list of queries to score in full
    }
}

// Every live row scored in strict arithmetic for the queries the verification listed (a block of 16 wavefronts takes
// one listed query at a time): each wavefront keeps the top_k of its share of the rows, wavefront 0 merges them by
// (sort_key, id).  Replaces the query's results.
constexpr int FB_WAVES = 16;
__global__ void __launch_bounds__(64 * FB_WAVES) knn_fallback_kernel(GraphView gv, const float* __restrict__ queries, uint32_t top_k,
                                                                   const uint32_t* __restrict__ flags, const uint32_t* __restrict__ flag_count,
                                                                   vdb_scored_point* __restrict__ out, uint32_t* __restrict__ out_n) {
    const uint32_t listed = *flag_count;
    for (uint32_t fi = blockIdx.x; fi < listed; fi += gridDim.x) {
    const uint32_t q = flags[fi];
    __syncthreads();                                                          // the previous query's lists are done with
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t dim = gv.dim, stride = gv.stride;
    float* ql = reinterpret_cast<float*>(smem);
    uint64_t* lid = reinterpret_cast<uint64_t*>(smem + (size_t)stride * 4) + (size_t)wave * top_k;
    float* lkey = reinterpret_cast<float*>(smem + (size_t)stride * 4 + (size_t)FB_WAVES * top_k * 8) + (size_t)wave * top_k;
    float* lraw = lkey + (size_t)FB_WAVES * top_k;
    uint32_t* ln = reinterpret_cast<uint32_t*>(smem + (size_t)stride * 4 + (size_t)FB_WAVES * top_k * 16);
    for (uint32_t i = threadIdx.x; i < stride; i += blockDim.x) ql[i] = i < dim ? queries[(size_t)q * dim + i] : 0.f;
    __syncthreads();
    if (gv.metric == VDB_COSINE) {                                            // hnsw.rs:526-536, 752-764
        float s = -0.0f;
        for (uint32_t i = 0; i < dim; ++i) { float p = ql[i] * ql[i]; s = s + p; }
        const float norm = __builtin_sqrtf(s);
        __syncthreads();
        if (norm != 0.0f) for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) ql[i] = ql[i] / norm;
        __syncthreads();
    }
    TopKList L{lkey, lid, lraw, 0, top_k};
    for (uint32_t b = wave * WAVE; b < gv.n; b += FB_WAVES * WAVE) {
        const uint32_t pos = b + lane;
        uint32_t s = pos < gv.n ? (gv.rowmap ? gv.rowmap[pos] : pos) : 0u;
        const bool valid = pos < gv.n && !(gv.deleted && bit_test(gv.deleted, s));
        float raw = 0.f; uint64_t idv = 0;
        if (valid) { raw = strict_pair(ql, gv.vecs + (size_t)s * stride, dim, gv.metric, true); idv = gv.ids[s]; }
        const float k = gv.metric == VDB_DOT ? -raw : raw;
        uint64_t am = ballot(valid && L.admits(k, idv));
        while (am) {
            const int jj = __ffsll((unsigned long long)am) - 1; am &= am - 1;
            const float kk = readlane_f(k, jj), rr = readlane_f(raw, jj);
            const uint64_t ii = ((uint64_t)readlane_u((uint32_t)(idv >> 32), jj) << 32) | readlane_u((uint32_t)idv, jj);
            if (!L.admits(kk, ii)) continue;
            L.insert(kk, ii, rr);
        }
    }
    if (lane == 0) ln[wave] = L.n;
    __syncthreads();
    if (wave != 0) continue;
    for (int w = 1; w < FB_WAVES; ++w) {                                       // wavefront 0 takes in the other lists
        const uint32_t nw = ln[w];
        for (uint32_t i = 0; i < nw; ++i) {
            const float kk = lkey[(size_t)w * top_k + i]; const uint64_t ii = lid[(size_t)w * top_k + i]; const float rr = lraw[(size_t)w * top_k + i];
            if (L.admits(kk, ii)) L.insert(kk, ii, rr);
        }
    }
    for (uint32_t i = lane; i < L.n; i += WAVE) { vdb_scored_point sp; sp.id = lid[i]; sp.raw_score = lraw[i]; sp.sort_key = lkey[i]; out[(size_t)q * top_k + i] = sp; }
    if (lane == 0) out_n[q] = L.n;
    }
}

// Blocks of one launch do equal work, so the launch lasts ceil(blocks / resident blocks) rounds: the base set is cut
// into the number of chunks that wastes least of the last round (more chunks also mean more candidates to re-rank).
inline uint32_t pick_chunks(uint32_t n, uint32_t nq, uint32_t qblock, uint32_t resident, uint32_t max_chunks) {
    const uint32_t qblocks = (nq + qblock - 1) / qblock;
    const uint32_t cmax = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(64, max_chunks), n / 4096));
    uint32_t best = 1; double best_cost = 1e30;
    for (uint32_t c = 1; c <= cmax; ++c) {
        const uint32_t rounds = (c * qblocks + resident - 1) / resident;
        const double cost = (double)rounds / c + 0.0008 * c;
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}
template <class Kernel>
uint32_t resident_blocks(Kernel kernel, size_t lds) {
    int dev = 0, cus = 256, per_cu = 1;
    VDB_HIP(hipGetDevice(&dev));
    VDB_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    VDB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds));
    return (uint32_t)std::max(1, cus) * (uint32_t)std::max(1, per_cu);
}
template <int KH, int KP>
uint32_t launch_mfma(const GraphView& g, const float* queries, uint32_t nq, const float* scale, const float* bias, uint32_t* slots, float* thr, uint32_t max_chunks, hipStream_t st) {
    constexpr int K = 2 * KH;
    constexpr int ROWS = MT;
    const size_t lds = ((size_t)ROWS * (K + 4) * 4 + 2 * ROWS * 4) + MQ * 4 + (size_t)MQ * KnnBuf<KP>::CAP * 8;
    const uint32_t n_chunks = pick_chunks(g.n, nq, MQ, resident_blocks(knn_mfma_kernel<KH, KP>, lds), max_chunks);
    hipLaunchKernelGGL((knn_mfma_kernel<KH, KP>), dim3(n_chunks, (nq + MQ - 1) / MQ), dim3(256), lds, st, g, queries, nq, scale, bias, slots, thr, n_chunks);
    return n_chunks;
}
// padded row length of the bf16 planes of the queries
inline uint32_t knn_bf16_k(uint32_t stride) { return stride <= 64 ? 64u : stride <= 128 ? 128u : stride <= 256 ? 256u : (stride + 63u) / 64u * 64u; }
template <int KS, int KP, int WPB>
uint32_t launch_bf16(const GraphView& g, const __bf16* qh, const __bf16* ql, uint32_t nq, const float* centre, const float* scale, const float* bias, uint32_t* slots, float* thr, const float* init, uint32_t max_chunks, hipStream_t st) {
    constexpr int K = 16 * KS, MQB = 32 * WPB;
    const size_t lds = (size_t)KNN_BF16_BUFFERS * ((size_t)2 * MT * (K + 8) * 2 + 2 * MT * 4) + MQB * 4 + (size_t)MQB * KnnBuf<KP>::CAP * 8;
    int dev = 0, cus = 256, per_cu = 1;
    VDB_HIP(hipGetDevice(&dev));
    VDB_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    VDB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, knn_bf16_kernel<KS, KP, WPB>, 64 * WPB, lds));
    const uint32_t n_chunks = pick_chunks(g.n, nq, MQB, (uint32_t)std::max(1, cus) * (uint32_t)std::max(1, per_cu), max_chunks);
    hipLaunchKernelGGL((knn_bf16_kernel<KS, KP, WPB>), dim3(n_chunks, (nq + MQB - 1) / MQB), dim3(64 * WPB), lds, st, g, qh, ql, nq, centre, scale, bias, slots, thr, n_chunks, init);
    return n_chunks;
}
template <int KP>
uint32_t launch_bf16_any(const GraphView& g, const __bf16* qh, const __bf16* ql, uint32_t nq, const float* centre, const float* scale, const float* bias, uint32_t* slots, float* thr, const float* init, uint32_t max_chunks, hipStream_t st) {
    const bool force_long = std::getenv("VDB_KNN_LONG") != nullptr;   // diagnostic: the chunked kernel on short rows
    if (!force_long && g.stride <= 64) return launch_bf16<4, KP, 4>(g, qh, ql, nq, centre, scale, bias, slots, thr, init, max_chunks, st);
    if (!force_long && g.stride <= 128) return launch_bf16<8, KP, 4>(g, qh, ql, nq, centre, scale, bias, slots, thr, init, max_chunks, st);
    if (!force_long && g.stride <= 256) return launch_bf16<16, KP, 8>(g, qh, ql, nq, centre, scale, bias, slots, thr, init, max_chunks, st);
    const size_t lds = (size_t)2 * LT * LDK2 * 2 + 2 * LT * 4 + MQ * 4 + (size_t)MQ * KnnBuf<KP>::CAP * 8 + (size_t)knn_bf16_k(g.stride) * 4;
    const uint32_t n_chunks = pick_chunks(g.n, nq, MQ, resident_blocks(knn_bf16_long_kernel<KP>, lds), max_chunks);
    hipLaunchKernelGGL((knn_bf16_long_kernel<KP>), dim3(n_chunks, (nq + MQ - 1) / MQ), dim3(256), lds, st, g, qh, ql, knn_bf16_k(g.stride), nq, centre, scale, bias, slots, thr, n_chunks, init);
    return n_chunks;
}
template <int KP>
uint32_t launch_mfma_any(const GraphView& g, const float* queries, uint32_t nq, const float* scale, const float* bias, uint32_t* slots, float* thr, uint32_t max_chunks, hipStream_t st) {
    if (g.stride <= 64) return launch_mfma<32, KP>(g, queries, nq, scale, bias, slots, thr, max_chunks, st);
    if (g.stride <= 128) return launch_mfma<64, KP>(g, queries, nq, scale, bias, slots, thr, max_chunks, st);
    if (g.stride <= 256) return launch_mfma<128, KP>(g, queries, nq, scale, bias, slots, thr, max_chunks, st);
    const size_t lds = (size_t)2 * LT * LDK * 4 + 2 * LT * 4 + MQ * 4 + (size_t)MQ * KnnBuf<KP>::CAP * 8;
    const uint32_t n_chunks = pick_chunks(g.n, nq, MQ, resident_blocks(knn_mfma_long_kernel<KP>, lds), max_chunks);
    hipLaunchKernelGGL((knn_mfma_long_kernel<KP>), dim3(n_chunks, (nq + MQ - 1) / MQ), dim3(256), lds, st, g, queries, nq, scale, bias, slots, thr, n_chunks);
    return n_chunks;
}
inline uint32_t knn_kp(uint32_t top_k) { return top_k + 4 <= 16 ? 16u : 32u; }

}  // namespace

bool knn_mfma_supported(const GraphView& g, uint32_t top_k) { return top_k >= 1 && top_k + 4 <= 32 && g.stride <= MAX_DIM && g.n > 0; }
// room for the largest chunk count and candidate width the launch may pick
// words of the fixed part of the scratch: the centre, row scale and bias (rounded up to 16 bytes) and the two bf16 planes of the queries
// (and, for the verification of the candidate pass, the largest squared row norm — four words — and one flag per query)
static size_t knn_fixed_words(uint32_t n, uint32_t nq, uint32_t stride) {
    return knn_bf16_k(stride) + ((2 * (size_t)n + 3) & ~size_t(3)) + (size_t)nq * knn_bf16_k(stride) + 4 + (((size_t)nq + 3) & ~size_t(3));
}
size_t knn_mfma_scratch_words(uint32_t n, uint32_t nq, uint32_t stride) {
    return knn_fixed_words(n, nq, stride) + (size_t)nq * std::max<uint32_t>(1, std::min<uint32_t>(64, n / 4096)) * 33;
}

// scratch: [centre K][scale n][bias n][query planes: 2 x nq x K bf16][largest squared row norm: 4][flags nq]
//          [candidate slots: nq x max_chunks x KP][final bounds: nq x max_chunks]
// hint: `out` / `out_n` hold an approximate answer (top_k actual rows per query, the graph's): every query's candidate
// pass then starts from the bound that answer gives instead of +inf (knn_bound_kernel), which spares most admissions.
static void run_exact_knn_mfma(const GraphView& g, const float* queries, uint32_t nq, uint32_t top_k, vdb_scored_point* out, uint32_t* out_n,
                               uint32_t* scratch, uint32_t max_chunks, hipStream_t st, bool hint) {
    const uint32_t kpad = knn_bf16_k(g.stride);
    const bool bf16 = !std::getenv("VDB_KNN_FP32");            // diagnostic: the fp32 MFMA kernels the bf16 ones are measured against
    float* centre = reinterpret_cast<float*>(scratch);
    float* scale = centre + kpad; float* bias = scale + g.n;
    __bf16* qh = reinterpret_cast<__bf16*>(scratch + kpad + ((2 * (size_t)g.n + 3) & ~size_t(3))); __bf16* ql = qh + (size_t)nq * kpad;
    uint32_t* max_norm2 = scratch + kpad + ((2 * (size_t)g.n + 3) & ~size_t(3)) + (size_t)nq * kpad;
    uint32_t* flags = max_norm2 + 4;
    uint32_t* slots = scratch + knn_fixed_words(g.n, nq, g.stride);
    // the launch takes at most this many chunks (pick_chunks), and the scratch is sized for them
    const uint32_t cap_chunks = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(64, max_chunks), g.n / 4096));
    float* thr = reinterpret_cast<float*>(slots + (size_t)nq * cap_chunks * knn_kp(top_k));
    VDB_HIP(hipMemsetAsync(max_norm2, 0, 16, st));
    if (bf16) {
        hipLaunchKernelGGL(knn_centre_kernel, dim3((kpad + 255) / 256), dim3(256), 0, st, g, centre, kpad,
                           std::getenv("VDB_KNN_NO_CENTRE") ? 0u : std::min<uint32_t>(g.n, 2048u));   // (diagnostic: what the centring buys)
        VDB_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(knn_prepare_kernel, dim3((g.n + 31) / 32), dim3(256), 0, st, g, bf16 ? centre : nullptr, scale, bias, max_norm2);
    VDB_HIP(hipGetLastError());
    const uint32_t kp = knn_kp(top_k);
    uint32_t n_chunks;
    // the starting bounds share the words of the flag list, which is written only after the candidate pass has run
    const float* init = nullptr;
    if (bf16 && hint) {
        hipLaunchKernelGGL(knn_bound_kernel<false>, dim3(nq), dim3(64), 0, st, g, queries, centre, top_k, max_norm2, out, out_n, reinterpret_cast<float*>(flags));
        VDB_HIP(hipGetLastError());
        init = reinterpret_cast<const float*>(flags);
    }
    if (bf16) {
        const size_t cells = (size_t)nq * kpad;
        hipLaunchKernelGGL(knn_split_queries_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, queries, centre, nq, g.dim, kpad, qh, ql);
        VDB_HIP(hipGetLastError());
        n_chunks = kp == 16 ? launch_bf16_any<16>(g, qh, ql, nq, centre, scale, bias, slots, thr, init, max_chunks, st)
                            : launch_bf16_any<32>(g, qh, ql, nq, centre, scale, bias, slots, thr, init, max_chunks, st);
    } else {
        n_chunks = kp == 16 ? launch_mfma_any<16>(g, queries, nq, scale, bias, slots, thr, max_chunks, st)
                            : launch_mfma_any<32>(g, queries, nq, scale, bias, slots, thr, max_chunks, st);
    }
    VDB_HIP(hipGetLastError());
    const size_t lds = (size_t)g.stride * 4 + (size_t)top_k * 16;
    // (the fp32 MFMA pass, kept as a diagnostic, is not verified: its keys carry a different error)
    const bool verify = bf16 && !std::getenv("VDB_KNN_NO_VERIFY");
    hipLaunchKernelGGL(knn_rerank_kernel, dim3(nq), dim3(64), lds, st, g, queries, top_k, slots, n_chunks * kp, thr, n_chunks,
                       g.metric == VDB_EUCLIDEAN ? centre : nullptr, max_norm2, verify ? flags : nullptr, max_norm2 + 1, out, out_n);
    VDB_HIP(hipGetLastError());
    if (bf16 && std::getenv("VDB_KNN_ORACLE_BOUND") && g.stride > 64 && g.stride <= 128 && kp == 16) {
        // diagnostic: the same candidate pass again, every query starting from the bound an oracle would give it; timed
        float* oinit = nullptr; VDB_HIP(hipMalloc(&oinit, (size_t)nq * 4));
        hipLaunchKernelGGL(knn_bound_kernel<true>, dim3(nq), dim3(64), 0, st, g, queries, centre, top_k, max_norm2, out, out_n, oinit);
        constexpr int KS = 8, WPB = 4, K = 16 * KS, MQB = 32 * WPB;
        const size_t l2 = (size_t)KNN_BF16_BUFFERS * ((size_t)2 * MT * (K + 8) * 2 + 2 * MT * 4) + MQB * 4 + (size_t)MQB * KnnBuf<16>::CAP * 8;
        hipEvent_t e0, e1, e2; VDB_HIP(hipEventCreate(&e0)); VDB_HIP(hipEventCreate(&e1)); VDB_HIP(hipEventCreate(&e2));
        VDB_HIP(hipEventRecord(e0, st));
        hipLaunchKernelGGL((knn_bf16_kernel<KS, 16, WPB>), dim3(n_chunks, (nq + MQB - 1) / MQB), dim3(64 * WPB), l2, st, g, qh, ql, nq, centre, scale, bias, slots, thr, n_chunks, (const float*)nullptr);
        VDB_HIP(hipEventRecord(e1, st));
        hipLaunchKernelGGL((knn_bf16_kernel<KS, 16, WPB>), dim3(n_chunks, (nq + MQB - 1) / MQB), dim3(64 * WPB), l2, st, g, qh, ql, nq, centre, scale, bias, slots, thr, n_chunks, (const float*)oinit);
        VDB_HIP(hipEventRecord(e2, st)); VDB_HIP(hipEventSynchronize(e2));
        float a = 0, b = 0; VDB_HIP(hipEventElapsedTime(&a, e0, e1)); VDB_HIP(hipEventElapsedTime(&b, e1, e2));
        fprintf(stderr, "[knn] candidate pass %u chunks: %.3f ms from +inf bounds, %.3f ms from oracle bounds\n", n_chunks, a, b);
        VDB_HIP(hipFree(oinit));
        hipLaunchKernelGGL(knn_rerank_kernel, dim3(nq), dim3(64), lds, st, g, queries, top_k, slots, n_chunks * kp, thr, n_chunks,
                           g.metric == VDB_EUCLIDEAN ? centre : nullptr, max_norm2, (uint32_t*)nullptr, max_norm2 + 1, out, out_n);
    }
    if (verify) {
        const size_t fl = (size_t)g.stride * 4 + (size_t)FB_WAVES * top_k * 16 + FB_WAVES * 4;
        hipLaunchKernelGGL(knn_fallback_kernel, dim3(std::min<uint32_t>(nq, 1024u)), dim3(64 * FB_WAVES), fl, st, g, queries, top_k, flags, max_norm2 + 1, out, out_n);
        VDB_HIP(hipGetLastError());
    }
}

// scratch: knn_mfma_scratch_words() dwords
void launch_exact_knn_mfma(const GraphView& g, const float* queries, uint32_t nq, uint32_t top_k, vdb_scored_point* out, uint32_t* out_n,
                           uint32_t* scratch, hipStream_t st, bool hint) {
    if (nq == 0) return;
    run_exact_knn_mfma(g, queries, nq, top_k, out, out_n, scratch, 64, st, hint);
}

// ---- the same pass on a scratch area that is part of the library's code object: nothing is allocated ----
// One array per GPU (each device that loads the library holds its own copy until the process ends); every call on a
// device shares it, so calls that overlap on one GPU race.  All of it is rewritten by the kernels before it is read
// (scale / bias by knn_prepare_kernel, every candidate slot of every (query, chunk) by the contraction), so nothing
// has to be cleared between calls.
constexpr size_t KNN_STATIC_WORDS = size_t(4) << 20;                       // 16 MiB
static __device__ __align__(256) uint32_t knn_static_scratch[KNN_STATIC_WORDS];

size_t knn_mfma_static_words() { return KNN_STATIC_WORDS; }

// Device address of this GPU's copy of knn_static_scratch (the array's name is a host-side stand-in, not an address
// a kernel can use); looked up once per device with that device current.
static uint32_t* knn_static_scratch_ptr() {
    constexpr int MAX_DEV = 64;
    static std::mutex mu;
    static uint32_t* ptr[MAX_DEV] = {nullptr};
    int dev = 0;
    VDB_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) throw Failure(err_device("device ordinal beyond the static scratch table"));
    std::lock_guard<std::mutex> lk(mu);
    if (!ptr[dev]) {
        void* p = nullptr;
        VDB_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(knn_static_scratch)));
        if (!p) throw Failure(err_device("hipGetSymbolAddress returned no address for the static scratch"));
        ptr[dev] = static_cast<uint32_t*>(p);
    }
    return ptr[dev];
}

void launch_exact_knn_mfma_static(const GraphView& g, const float* queries, uint32_t nq, uint32_t top_k, vdb_scored_point* out, uint32_t* out_n,
                                  hipStream_t st) {
    if (nq == 0) return;
    const uint32_t kp = knn_kp(top_k);
    const size_t fixed = knn_fixed_words(g.n, nq, g.stride), per_chunk = (size_t)nq * (kp + 1);
    if (fixed + per_chunk > KNN_STATIC_WORDS)
        throw Failure(err_arg("exact k-NN without allocation: 2 * rows + queries * (" + std::to_string(kp) + " + padded row length up to 256) = " + std::to_string(fixed + per_chunk) +
                              " words exceed the static scratch of " + std::to_string(KNN_STATIC_WORDS) + " words (16 MiB); use vdb_index_exact_knn_batch_dev"));
    const uint32_t max_chunks = (uint32_t)std::min<size_t>(64, (KNN_STATIC_WORDS - fixed) / per_chunk);
    run_exact_knn_mfma(g, queries, nq, top_k, out, out_n, knn_static_scratch_ptr(), max_chunks, st, false);
}

}  // namespace vdb
