// gfx950 (MI355X) kernels of libvdbmi other than the graph search: the streaming distance scan, row
// normalisation, the exact k-NN and the k-way merge; plus the dispatcher of the search launchers.
// Compiled with -ffp-contract=off (see device_common.hpp).
#include <type_traits>
#include "device_common.hpp"

#include <algorithm>

namespace vdb {

void launch_search_plain(const SearchArgs& a, size_t lds, hipStream_t st);
void launch_search_build(const SearchArgs& a, size_t lds, hipStream_t st);
void launch_search_masked(const SearchArgs& a, size_t lds, hipStream_t st);
void launch_search_literal(const SearchArgs& a, size_t lds, hipStream_t st);
void launch_search_team_cosine(const SearchArgs& a, hipStream_t st);
void launch_search_team_dot(const SearchArgs& a, hipStream_t st);
void launch_search_team_euclidean(const SearchArgs& a, hipStream_t st);

uint32_t team_size_for(const SearchArgs& a) {
    if (a.mode != MODE_SEARCH || reg_list_chunks(a) == 0 || !a.g.ublk) return 0;
    if (const char* e = std::getenv("VDB_TEAM")) { const int t = std::atoi(e); return t == 2 || t == 4 ? (uint32_t)t : 0u; }
    // 1024 SIMDs hold four wavefronts each at the register budget of these kernels
    return a.nq <= 1024 ? 4u : a.nq <= 2048 ? 2u : 0u;
}

size_t search_lds_bytes(const SearchArgs& a) {
    const size_t ef_cap = (a.ef + 63u) & ~63u;
    const size_t nc = reg_list_chunks(a), hash = (size_t)4 << a.hash_log2, row = (size_t)a.g.stride * 4;
    const size_t staging = (nc ? 256 : 64) * 8;      // cand + craw
    const size_t tags = nc ? ((size_t)2 << a.tag_log2) : 0;   // 16-bit visited table
    if (nc && a.mode == MODE_SEARCH) return ((std::max(std::max(nc * 512, row), staging) + 15) & ~(size_t)15) + tags;   // one shared staging area
    if (nc && a.mode == MODE_MASKED) return ((std::max(std::max((nc + 1) * 512, row), staging) + 15) & ~(size_t)15) + tags;   // + the passing list
    return ef_cap * 16 + (a.mode == MODE_MASKED ? ef_cap * 16 : 0) + nc * 512 + tags + staging + row + hash;
}

void launch_search(const SearchArgs& a, hipStream_t st) {
    if (a.nq == 0) return;
    if (a.team) {
        switch (a.g.metric) {
            case VDB_COSINE: launch_search_team_cosine(a, st); break;
            case VDB_DOT: launch_search_team_dot(a, st); break;
            default: launch_search_team_euclidean(a, st); break;
        }
        VDB_HIP(hipGetLastError());
        return;
    }
    const size_t lds = search_lds_bytes(a);
    if (lds > 160 * 1024) throw Failure(err_arg("search working set exceeds the 160 KiB LDS of a CU"));
    switch (a.mode) {
        case MODE_BUILD: launch_search_build(a, lds, st); break;
        case MODE_MASKED: launch_search_masked(a, lds, st); break;
        case MODE_LITERAL: launch_search_literal(a, lds, st); break;
        default: launch_search_plain(a, lds, st); break;
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

// Shape of the FAST scan, from measurements on 1M x 32 .. 1536 rows against 8 queries: four rows per lane group and
// step for most row lengths (the queries are read from LDS once per step: one row per step is 8 % slower than two,
// two 4-6 % slower than four, eight slower again), the step's rows fetched and scored in one buffer (with four rows
// per step a second buffer in flight adds nothing: the other wavefronts of the CU cover the fetch), all queries of
// the block in one pass over the rows' registers.
constexpr int SCAN_QH = 8, SCAN_QB = 1, SCAN_NB = 2;
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// V partial sums per lane, to be summed over W adjacent lanes.  While more than one value is left a step halves the
// values instead of repeating the exchange: the lane whose bit W/2 is set keeps the upper half and hands the lower half to
// its partner, so V values cost about V exchanges instead of V log2(W).  Afterwards, with g the lane's position among
// the G lanes the first call was made for: V >= G leaves a[0 .. V/G) = totals of values g * (V/G) + i; V < G leaves
// a[0] = total of value g / (G/V) in every lane of that run.
template <int V, int W, int STRIDE = 1>                  // STRIDE: distance in lanes between neighbours of the group
__device__ __forceinline__ void fold_lanes(float* a, int g) {
    if constexpr (W > 1) {
        constexpr int M = W / 2;
        if constexpr (V > 1) {
            const bool up = (g & M) != 0;
#pragma unroll
            for (int i = 0; i < V / 2; ++i) {
                const float send = up ? a[i] : a[i + V / 2];
                const float keep = up ? a[i + V / 2] : a[i];
                a[i] = keep + __shfl_xor(send, M * STRIDE, WAVE);
            }
            fold_lanes<V / 2, M, STRIDE>(a, g);
        } else {
            a[0] += __shfl_xor(a[0], M * STRIDE, WAVE);
            fold_lanes<1, M, STRIDE>(a, g);
        }
    }
}

// FAST scan.  G lanes share a row (CH float4 each), a wave step covers R rows per group = R * 64 / G rows, and the rows
// of the next step are in flight while this one is scored against the QT queries of the block (read from LDS once per
// R rows).  Sums run as packed pairs (v_pk_fma_f32) and are folded across the group by fold_lanes.
template <int G, int CH, int R, int QT, int NB>
__global__ void __launch_bounds__(256) scan_fast_kernel(const float* __restrict__ queries, uint32_t nq, const float* __restrict__ base,
                                                       uint32_t n, uint32_t dim, uint32_t stride, int metric, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* q = reinterpret_cast<float*>(smem);
    const uint32_t qpad = (dim + 3u) & ~3u;
    const uint32_t q0 = blockIdx.y * SCAN_Q;
    const uint32_t qn = min((uint32_t)QT, nq - q0);
    constexpr int QS = G * CH;                                 // float4 per query in LDS: one per (lane of the group, step), zero beyond the row
    for (uint32_t i = threadIdx.x; i < QT * QS * 4; i += blockDim.x) {
        const uint32_t t = i / (QS * 4), d = i % (QS * 4);
        q[i] = (t < qn && d < dim) ? queries[(size_t)(q0 + t) * dim + d] : 0.f;
    }
    __syncthreads();
    constexpr int RPW = (WAVE / G) * R;                        // rows per wave step
    constexpr int QH = QT < SCAN_QH ? QT : SCAN_QH;                        // queries scored per pass over the registers holding the rows
    constexpr int V = QH * R;                                  // partial sums per lane and pass
    const int lane = lane_id(), g = lane % G, rr = lane / G;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, nwaves = gridDim.x * blockDim.x / WAVE;
    const uint32_t nchunks = qpad >> 2;
    const float4* q4 = reinterpret_cast<const float4*>(q);
    float qn2[QT];
    if (metric == VDB_COSINE) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float acc = 0.f;
            for (int j = 0; j < CH; ++j) {
                const uint32_t c = g + G * j;
                const float4 v = q4[t * QS + c];
                acc = __builtin_fmaf(v.x, v.x, acc); acc = __builtin_fmaf(v.y, v.y, acc); acc = __builtin_fmaf(v.z, v.z, acc); acc = __builtin_fmaf(v.w, v.w, acc);
            }
            qn2[t] = __builtin_sqrtf(group_sum<G>(acc));
        }
    }
    constexpr int VO = V >= G ? V / G : 1;                     // totals a lane holds after the fold
    constexpr int RUN = V >= G ? 1 : G / V;                    // lanes that end up with the same total
    const bool writer = g % RUN == 0;
    uint32_t my_t[VO], my_i[VO];              // which (query, row of the step) each total belongs to
#pragma unroll
    for (int o = 0; o < VO; ++o) {
        const uint32_t idx = V >= G ? (uint32_t)g * VO + o : (uint32_t)g / RUN;
        my_t[o] = idx / R; my_i[o] = idx % R;                  // query within the pass
    }
    auto fetch = [&](float4 (&dst)[R][CH], uint32_t r0) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const uint32_t r = min(r0 + rr * R + i, n - 1);   // rows past the end repeat the last one and are not written
            const floatx4* row = reinterpret_cast<const floatx4*>(base + (size_t)r * stride);
#pragma unroll
            for (int j = 0; j < CH; ++j) {                     // unconditional: no predicated loads; streamed: a row is read once
                const floatx4 v = __builtin_nontemporal_load(row + min((uint32_t)(g + G * j), nchunks - 1));
                dst[i][j] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    auto score = [&](float4 (&x)[R][CH], uint32_t r0) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if ((uint32_t)(G * (j + 1)) > nchunks) {           // uniform: only the last float4 steps of a row can run past its end
                const bool past = (uint32_t)(g + G * j) >= nchunks;
#pragma unroll
                for (int i = 0; i < R; ++i) if (past) x[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float xnorm[R];
        if (metric == VDB_COSINE) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                floatx2 s{0.f, 0.f};
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const floatx2 va{x[i][j].x, x[i][j].y}, vb{x[i][j].z, x[i][j].w};
                    s = __builtin_elementwise_fma(va, va, s); s = __builtin_elementwise_fma(vb, vb, s);
                }
                xnorm[i] = __builtin_sqrtf(group_sum<G>(s.x + s.y));
            }
        }
#pragma unroll
        for (int h = 0; h < QT; h += QH) {
            floatx2 acc[QH][R];
#pragma unroll
            for (int t = 0; t < QH; ++t)
#pragma unroll
                for (int i = 0; i < R; ++i) acc[t][i] = floatx2{0.f, 0.f};
            auto accumulate = [&](auto l2) {
                constexpr int QB = CH <= 4 ? SCAN_QB : 1;            // the next query's slice is read while this one is used (short slices)
                float4 qq[QB][CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) qq[0][j] = q4[h * QS + g + G * j];
#pragma unroll
                for (int t = 0; t < QH; ++t) {
                    if (QB == 2 && t + 1 < QH) {
#pragma unroll
                        for (int j = 0; j < CH; ++j) qq[(t + 1) & 1][j] = q4[(h + t + 1) * QS + g + G * j];
                    }
                    if (QB == 1 && t > 0) {
#pragma unroll
                        for (int j = 0; j < CH; ++j) qq[0][j] = q4[(h + t) * QS + g + G * j];
                    }
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const float4 qv = qq[QB == 2 ? (t & 1) : 0][j];
                        const floatx2 qa{qv.x, qv.y}, qb{qv.z, qv.w};
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            const floatx2 va{x[i][j].x, x[i][j].y}, vb{x[i][j].z, x[i][j].w};
                            if constexpr (decltype(l2)::value) {
                                const floatx2 da = qa - va, db = qb - vb;
                                acc[t][i] = __builtin_elementwise_fma(da, da, acc[t][i]);
                                acc[t][i] = __builtin_elementwise_fma(db, db, acc[t][i]);
                            } else {
                                acc[t][i] = __builtin_elementwise_fma(qa, va, acc[t][i]);
                                acc[t][i] = __builtin_elementwise_fma(qb, vb, acc[t][i]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);         // keeps the LDS reads two queries deep: the rows hold the registers
                }
            };
            if (metric == VDB_EUCLIDEAN) accumulate(std::true_type{}); else accumulate(std::false_type{});
            float a[V];
#pragma unroll
            for (int t = 0; t < QH; ++t)
#pragma unroll
                for (int i = 0; i < R; ++i) a[t * R + i] = acc[t][i].x + acc[t][i].y;
            fold_lanes<V, G>(a, g);
#pragma unroll
            for (int o = 0; o < VO; ++o) {
                const uint32_t r = r0 + rr * R + my_i[o], t = h + my_t[o];
                float raw = a[o];
                if (metric == VDB_EUCLIDEAN) raw = __builtin_sqrtf(raw);
                else if (metric == VDB_COSINE) {
                    float xv = 0.f, qv = 0.f;
#pragma unroll
                    for (int ii = 0; ii < R; ++ii) xv = my_i[o] == (uint32_t)ii ? xnorm[ii] : xv;
#pragma unroll
                    for (int tt = 0; tt < QH; ++tt) qv = my_t[o] == (uint32_t)tt ? qn2[h + tt] : qv;
                    raw = 1.0f - raw / (qv * xv + 1e-10f);
                }
                if (writer && r < n && t < qn) out[(size_t)(q0 + t) * n + r] = raw;
            }
        }
    };
    // NB row buffers in turn: the rows of the next NB - 1 steps are in flight while this step is scored (the bytes in
    // flight per CU, not the arithmetic, set the speed of this kernel)
    float4 xb[NB][R][CH];
    const uint32_t step = nwaves * RPW;
    uint32_t r0 = wave * RPW;
#pragma unroll
    for (int b = 0; b < NB - 1; ++b) if (r0 + b * step < n) fetch(xb[b], r0 + b * step);
    while (true) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (r0 >= n) return;
            if (r0 + (NB - 1) * step < n) fetch(xb[(b + NB - 1) % NB], r0 + (NB - 1) * step);
            score(xb[b], r0);
            r0 += step;
        }
    }
}

// FAST scan of dot / cosine rows against a block of 3 .. 8 queries on the matrix cores: v_mfma_f32_4x4x1_16B_f32 is 16
// independent outer products of 4 x 4 (full fp32 products and sums), here 4 rows x 4 queries at 16 places of the row.
// A wavefront takes 4 rows per step: lane (b = lane / 4, i = lane % 4) loads the float4 at float4 position 16 p + b of
// row i (256 contiguous bytes per row and load) for the PC positions p of a tile; each of its four floats is the A
// operand of two MFMAs (queries 0..3 and 4..7), whose B operand -- lane (b, j): the same float of query j -- comes from
// LDS, laid out so that a wavefront reads 1 KiB in lane order.  The 4 x 8 sums of a step take 8 accumulator registers
// (64 in the vector kernel), which leaves room for the next tile's rows in flight while this one is multiplied; the
// vector units only square the rows for the cosine norm and fold the 16 partial products of a sum at the end of a row.
template <int PC, int RQ, bool COS>
__global__ void __launch_bounds__(256) scan_mfma_kernel(const float* __restrict__ queries, uint32_t nq, const float* __restrict__ base,
                                                       uint32_t n, uint32_t dim, uint32_t stride, uint32_t ntile, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qf = reinterpret_cast<float*>(smem);                // [position][half][b][j] float4: query 4 half + j at float4 16 position + b; zero beyond the row and the block's queries
    const float4* qb = reinterpret_cast<const float4*>(smem);
    const uint32_t q0 = blockIdx.y * SCAN_Q;
    const uint32_t qn = min((uint32_t)SCAN_Q, nq - q0);
    const uint32_t nchunks = (dim + 3u) >> 2;                  // float4 per row
    const uint32_t npos = ntile * PC;
    for (uint32_t i = threadIdx.x; i < npos * 512; i += blockDim.x) {
        const uint32_t comp = i & 3, j = (i >> 2) & 3, b = (i >> 4) & 15, half = (i >> 8) & 1, p = i >> 9;
        const uint32_t d = (p * 16 + b) * 4 + comp, t = half * 4 + j;
        qf[i] = (t < qn && d < dim) ? queries[(size_t)(q0 + t) * dim + d] : 0.f;
    }
    __syncthreads();
    const int lane = lane_id(), b = lane >> 2, i4 = lane & 3;
    // after the fold lane (b, j) holds the sum of row (b / 2) % 4 of its quad and query j + 4 (b / 8)
    const int my_row = (b >> 1) & 3, my_q = i4 + 4 * (b >> 3);
    float qnorm = 0.f;
    if (COS) {                                                 // |query my_q|: each of the 8 lanes that share it sums 1/8
        float a = 0.f;
        for (uint32_t f = b & 7; f < npos * 16; f += 8) {
            const float4 v = qb[((f >> 4) * 2 + (b >> 3)) * 64 + (f & 15) * 4 + i4];
            a = __builtin_fmaf(v.x, v.x, a); a = __builtin_fmaf(v.y, v.y, a); a = __builtin_fmaf(v.z, v.z, a); a = __builtin_fmaf(v.w, v.w, a);
        }
        a += __shfl_xor(a, 4, WAVE); a += __shfl_xor(a, 8, WAVE); a += __shfl_xor(a, 16, WAVE);
        qnorm = __builtin_sqrtf(a);
    }
    constexpr uint32_t RS = 4 * RQ;                            // rows per step: RQ quads of 4 rows share every read of the queries
    constexpr uint32_t SPAN = 32, SPS = SPAN / RS;             // a wavefront takes spans of 32 consecutive rows: its results leave as whole 128-byte lines per query
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, nwaves = gridDim.x * blockDim.x / WAVE;
    const uint32_t nspans = (n + SPAN - 1) / SPAN;
    if (wave >= nspans) return;
    const uint32_t my_steps = (nspans - wave + nwaves - 1) / nwaves * SPS;
    const uint32_t tiles = my_steps * ntile;
    float* stage = qf + (size_t)npos * 512 + (threadIdx.x / WAVE) * (SCAN_Q * SPAN);      // [query][row of the span], this wavefront's own
    const bool wide = (n % 4 == 0) && ((uintptr_t)out % 16 == 0);
    auto row_of = [&](uint32_t step) { return (wave + (step / SPS) * nwaves) * SPAN + (step % SPS) * RS; };

    auto fetch = [&](float4 (&a)[RQ][PC], uint32_t step, uint32_t c) {
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
            const uint32_t row = min(row_of(step) + 4u * r + (uint32_t)i4, n - 1);   // rows past the end repeat the last one and are not written
            const floatx4* p = reinterpret_cast<const floatx4*>(base + (size_t)row * stride);
#pragma unroll
            for (int g = 0; g < PC; ++g) {                     // unconditional: no predicated loads; streamed: a row is read once
                const floatx4 v = __builtin_nontemporal_load(p + min((c * PC + g) * 16u + (uint32_t)b, nchunks - 1));
                a[r][g] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    // The results of a span leave one tile late, right after the next tile's loads were issued: the memory counter of a
    // wavefront retires in order, so a store issued before those loads would have to be acknowledged before the tile
    // after it can be multiplied.
    float4 pend_v = make_float4(0.f, 0.f, 0.f, 0.f); uint32_t pend_row = 0; bool pending = false;
    auto flush = [&]() {
        if (!pending) return;                                  // uniform
        pending = false;
        const uint32_t t = (uint32_t)lane >> 3, row = pend_row;
        if (t < qn && row < n) {
            float* o = out + (size_t)(q0 + t) * n + row;
            if (wide) __builtin_nontemporal_store(floatx4{pend_v.x, pend_v.y, pend_v.z, pend_v.w}, reinterpret_cast<floatx4*>(o));
            else { o[0] = pend_v.x; if (row + 1 < n) o[1] = pend_v.y; if (row + 2 < n) o[2] = pend_v.z; if (row + 3 < n) o[3] = pend_v.w; }
        }
    };
    floatx4 lo[RQ], hi[RQ];                                    // [quad][row] of query i4 / i4 + 4, the part of place b
    floatx2 ss[RQ];
#pragma unroll
    for (int r = 0; r < RQ; ++r) { lo[r] = floatx4{0.f, 0.f, 0.f, 0.f}; hi[r] = floatx4{0.f, 0.f, 0.f, 0.f}; ss[r] = floatx2{0.f, 0.f}; }
    auto multiply = [&](float4 (&a)[RQ][PC], uint32_t step, uint32_t c) {
        if ((c + 1) * PC * 16u > nchunks) {                    // uniform: only the last tile of a row can run past its end
#pragma unroll
            for (int g = 0; g < PC; ++g)
                if ((c * PC + g) * 16u + (uint32_t)b >= nchunks) {
#pragma unroll
                    for (int r = 0; r < RQ; ++r) a[r][g] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
        const float4* bq = qb + (size_t)c * PC * 128 + lane;
#pragma unroll
        for (int g = 0; g < PC; ++g) {
            const float4 bl = bq[g * 128], bh = bq[g * 128 + 64];
#pragma unroll
            for (int r = 0; r < RQ; ++r) {
                const float4 x = a[r][g];
                lo[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, bl.x, lo[r], 0, 0, 0); hi[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, bh.x, hi[r], 0, 0, 0);
                lo[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, bl.y, lo[r], 0, 0, 0); hi[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, bh.y, hi[r], 0, 0, 0);
                lo[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.z, bl.z, lo[r], 0, 0, 0); hi[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.z, bh.z, hi[r], 0, 0, 0);
                lo[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.w, bl.w, lo[r], 0, 0, 0); hi[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.w, bh.w, hi[r], 0, 0, 0);
                if (COS) {
                    const floatx2 va{x.x, x.y}, vb{x.z, x.w};
                    ss[r] = __builtin_elementwise_fma(va, va, ss[r]); ss[r] = __builtin_elementwise_fma(vb, vb, ss[r]);
                }
            }
        }
        if (c + 1 != ntile) return;
        // the rows' sums are complete in 16 parts (one per b): fold them over b
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
            float v[8] = {lo[r][0], lo[r][1], lo[r][2], lo[r][3], hi[r][0], hi[r][1], hi[r][2], hi[r][3]};
            fold_lanes<8, 16, 4>(v, b);                        // every lane: the sum of value b / 2
            float raw = v[0];
            if (COS) {
                float t = ss[r].x + ss[r].y;                   // |row i4 of the quad|^2, the part of place b
                t += __shfl_xor(t, 4, WAVE); t += __shfl_xor(t, 8, WAVE); t += __shfl_xor(t, 16, WAVE); t += __shfl_xor(t, 32, WAVE);
                const float xv = __builtin_sqrtf(__shfl(t, my_row, WAVE));
                raw = 1.0f - raw / (qnorm * xv + 1e-10f);
            }
            if ((b & 1) == 0) stage[my_q * SPAN + (step % SPS) * RS + 4 * r + my_row] = raw;
            lo[r] = floatx4{0.f, 0.f, 0.f, 0.f}; hi[r] = floatx4{0.f, 0.f, 0.f, 0.f}; ss[r] = floatx2{0.f, 0.f};
        }
        if (step % SPS != SPS - 1) return;
        // the span is complete: lane l writes rows 4 (l % 8) .. + 3 of query l / 8
        wave_sync();
        const float4 v4 = reinterpret_cast<const float4*>(stage)[lane];
        wave_sync();
        pend_v = v4; pend_row = row_of(step - (SPS - 1)) + 4u * ((uint32_t)lane & 7); pending = true;
    };
    // two tile buffers in turn: the next tile's rows are in flight while this one is multiplied
    float4 xa[RQ][PC], xb[RQ][PC];
    uint32_t step = 0, c = 0;
    auto next = [&](uint32_t& s2, uint32_t& c2) { c2 = c + 1; s2 = step; if (c2 == ntile) { c2 = 0; s2 = min(step + 1, my_steps - 1); } };
    fetch(xa, 0, 0);
    for (uint32_t t = 0; t < tiles; t += 2) {
        uint32_t s2, c2;
        next(s2, c2);
        fetch(xb, s2, c2);                                     // past the last tile: a repeat of valid rows, never multiplied
        flush();
        multiply(xa, step, c);
        if (t + 1 >= tiles) break;
        step = c2 == 0 ? step + 1 : step; c = c2;
        next(s2, c2);
        fetch(xa, s2, c2);
        flush();
        multiply(xb, step, c);
        step = c2 == 0 ? step + 1 : step; c = c2;
    }
    flush();
}

__global__ void __launch_bounds__(64) score_rows_kernel(GraphView gv, uint32_t query_slot, const uint32_t* __restrict__ slots, uint32_t n, float* __restrict__ raw) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) raw[i] = strict_pair(gv.vecs + (size_t)query_slot * gv.stride, gv.vecs + (size_t)slots[i] * gv.stride, gv.dim, gv.metric, false);
}

__global__ void __launch_bounds__(64) score_pairs_kernel(GraphView gv, const uint32_t* __restrict__ qslots, const uint32_t* __restrict__ slots, uint32_t n, float* __restrict__ raw) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) raw[i] = strict_pair(gv.vecs + (size_t)qslots[i] * gv.stride, gv.vecs + (size_t)slots[i] * gv.stride, gv.dim, gv.metric, false);
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
    size_t lds = (size_t)SCAN_Q * qpad * 4;
    const uint32_t qblocks = (nq + SCAN_Q - 1) / SCAN_Q;
    const bool aligned = (stride % 4 == 0) && ((uintptr_t)base % 16 == 0);
    if (arith == VDB_ARITH_STRICT || !aligned || qpad / 4 > 512) {
        if (stride % 4 != 0 || (uintptr_t)base % 16 != 0) {
            // strict_pair reads rows with 16-byte loads: require the padded layout
            if (dim >= 4) throw Failure(err_arg("base rows must be 16-byte aligned with a stride that is a multiple of 4 floats"));
        }
        const uint32_t blocks = std::min<uint32_t>((n + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(scan_strict_kernel, dim3(blocks, qblocks), dim3(256), lds, st, queries, nq, base, n, dim, stride, metric, out);
    } else if (metric != VDB_EUCLIDEAN && nq >= 5 && qpad >= 320 && qpad <= 1920 && !std::getenv("VDB_SCAN_NO_MFMA")) {
        // dot / cosine rows of more than 256 floats against a block of 5 .. 8 queries: the matrix-core kernel (measured
        // against the vector kernel on 1M rows: faster from 5 queries and from rows of 384 floats on, equal at 256).
        // Up to 1920 floats per row the eight queries and the staging area fit 64 KiB of LDS; longer rows stay on the vector kernel.
        // Tiles of PC places of 16 float4, PC chosen for the least padding of the row.
        const uint32_t npos = (qpad / 4 + 15) / 16;
        const uint32_t pc = (npos + 2) / 3 * 3 - npos <= (npos + 1) / 2 * 2 - npos ? 3u : 2u;
        const uint32_t ntile = (npos + pc - 1) / pc;
        lds = (size_t)ntile * pc * 2048 + 4 * SCAN_Q * 32 * sizeof(float);          // the queries, then a staging area of the results per wavefront
        int dev = 0, cus = 0;
        VDB_HIP(hipGetDevice(&dev));
        VDB_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        // one launch of resident blocks: every wavefront walks its share of the rows
#define VDB_SCAN_M(PC, COS) do { int per_cu = 0; VDB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, scan_mfma_kernel<PC, 4, COS>, 256, lds)); \
                                 const dim3 grid((uint32_t)(cus * std::max(per_cu, 1)), qblocks); \
                                 hipLaunchKernelGGL((scan_mfma_kernel<PC, 4, COS>), grid, dim3(256), lds, st, queries, nq, base, n, dim, stride, ntile, out); } while (0)
#define VDB_SCAN_MC(PC) do { if (metric == VDB_COSINE) VDB_SCAN_M(PC, true); else VDB_SCAN_M(PC, false); } while (0)
        if (pc == 3) VDB_SCAN_MC(3); else VDB_SCAN_MC(2);
#undef VDB_SCAN_MC
#undef VDB_SCAN_M
    } else {
        const uint32_t chunks = qpad / 4;
        const uint32_t blocks = 256 * 8;
        const uint32_t qt = nq >= 5 ? 8u : nq >= 3 ? 4u : nq;          // queries a block scores (the last block may hold fewer)
#define VDB_SCAN_Q(G, CH, R, NB, QT) lds = (size_t)QT * G * CH * 16; hipLaunchKernelGGL((scan_fast_kernel<G, CH, R, QT, NB>), dim3(blocks, qblocks), dim3(256), lds, st, queries, nq, base, n, dim, stride, metric, out)
#define VDB_SCAN(G, CH, R, NB) do { if (qt == 8) { VDB_SCAN_Q(G, CH, R, NB, 8); } else if (qt == 4) { VDB_SCAN_Q(G, CH, R, NB, 4); } else if (qt == 2) { VDB_SCAN_Q(G, CH, R, NB, 2); } else { VDB_SCAN_Q(G, CH, R, NB, 1); } } while (0)
        // rows of up to 32 and of 97..128 floats: two row buffers in turn; the other shapes: twice the rows per step in one
        // buffer (half the LDS reads of the queries per row; measured 2-9 % faster there, slower on the two shapes above)
        if (chunks <= 8) VDB_SCAN(8, 1, 4, SCAN_NB); else if (chunks <= 16) VDB_SCAN(8, 2, 4, 1); else if (chunks <= 24) VDB_SCAN(8, 3, 4, 1);
        else if (chunks <= 32) VDB_SCAN(8, 4, 2, SCAN_NB); else if (chunks <= 64) VDB_SCAN(16, 4, 4, 1); else if (chunks <= 128) VDB_SCAN(32, 4, 4, 1);
        else if (chunks <= 192) VDB_SCAN(64, 3, 4, 1); else if (chunks <= 256) VDB_SCAN(64, 4, 4, 1); else VDB_SCAN(64, 8, 2, 1);
#undef VDB_SCAN_Q
#undef VDB_SCAN
    }
    VDB_HIP(hipGetLastError());
}

void launch_score_rows(const GraphView& g, uint32_t query_slot, const uint32_t* slots, uint32_t n, int, float* raw, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(score_rows_kernel, dim3((n + 63) / 64), dim3(64), 0, st, g, query_slot, slots, n, raw);
    VDB_HIP(hipGetLastError());
}

void launch_score_pairs(const GraphView& g, const uint32_t* qslots, const uint32_t* slots, uint32_t n, float* raw, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(score_pairs_kernel, dim3((n + 63) / 64), dim3(64), 0, st, g, qslots, slots, n, raw);
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
        const uint32_t rpos = rb + lane < r_end ? rb + lane : r_begin;
        const uint32_t r = gv.rowmap ? gv.rowmap[rpos] : rpos;                      // the slot behind this position
        const bool live = rb + lane < r_end && !(gv.deleted && bit_test(gv.deleted, r));
        const float* row = gv.vecs + (size_t)r * stride;
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

namespace {
__global__ void __launch_bounds__(64) patch_blocks_kernel(uint32_t* __restrict__ adj0, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ blocks) {
    adj0[(size_t)idx[blockIdx.x] * ADJ_BLOCK + threadIdx.x] = blocks[(size_t)blockIdx.x * ADJ_BLOCK + threadIdx.x];
}
}  // namespace
namespace {
__global__ void __launch_bounds__(256) patch_entries_kernel(uint32_t* __restrict__ adj0, const uint32_t* __restrict__ slot, const uint32_t* __restrict__ pos,
                                                            const uint32_t* __restrict__ val, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* blk = adj0 + (size_t)slot[i] * ADJ_BLOCK;
    blk[1 + pos[i]] = val[i];
    atomicMax(blk, pos[i] + 1u);
}
}  // namespace
void launch_patch_entries(uint32_t* adj0, const uint32_t* slot, const uint32_t* pos, const uint32_t* val, uint32_t n, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(patch_entries_kernel, dim3((n + 255) / 256), dim3(256), 0, st, adj0, slot, pos, val, n);
    VDB_HIP(hipGetLastError());
}

namespace {
__global__ void __launch_bounds__(256) patch_words_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ val, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = val[i];
}
}  // namespace
void launch_patch_words(uint32_t* dst, const uint32_t* idx, const uint32_t* val, uint32_t n, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(patch_words_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dst, idx, val, n);
    VDB_HIP(hipGetLastError());
}

void launch_patch_blocks(uint32_t* adj0, const uint32_t* idx, const uint32_t* blocks, uint32_t n, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(patch_blocks_kernel, dim3(n), dim3(64), 0, st, adj0, idx, blocks);
    VDB_HIP(hipGetLastError());
}

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
