// Payload filters on the device (src/payload_storage/filters.rs:23-75 over src/utils/payload.rs:36-186): a Filter
// tree is evaluated for every slot of a segment at once.  Leaves read the segment's payload columns (one kind byte
// and one 8-byte value per slot and key) or scatter a posting list of the inverted index (stores.rs:55-61) into a
// bitmap; And / Or / Not combine word-wise with the reference's short-circuit and error propagation.  Every node
// yields two bitmaps over the slots: pass (the filter is true) and err (evaluate_filter returned Err).  All of it is
// HBM-streaming byte / integer work: 9 bytes per slot and leaf, 1/8 byte per slot and bitmap.
#include "device_common.hpp"

namespace vdb {
namespace {

// OrderedFloat<f64> order (ordered-float 4.6): NaN equals NaN and sorts last, -0.0 equals 0.0.
__device__ __forceinline__ int of64_cmp(double a, double b) {
    const bool an = a != a, bn = b != b;
    if (an | bn) return (int)an - (int)bn;
    return (a > b) - (a < b);
}
__device__ __forceinline__ bool cmp_from_order(int op, int c) {            // ScalarComparisonOp, payload.rs:131-139
    switch (op) {
        case VDB_OP_EQ: return c == 0;
        case VDB_OP_NEQ: return c != 0;
        case VDB_OP_LT: return c < 0;
        case VDB_OP_LTE: return c <= 0;
        case VDB_OP_GT: return c > 0;
        default: return c >= 0;
    }
}

// One thread per slot; a wavefront's 64 outcomes become two bitmap words.
__global__ void __launch_bounds__(256) filter_leaf_kernel(FilterLeafArgs a) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    bool pass = false, err = false;
    if (s < a.n && bit_test(a.has, s)) {
        const uint32_t k = s < a.n_col ? a.kind[s] : 0u;              // 0: the payload lacks the key
        const int64_t v = k ? a.bits[s] : 0;
        const int have = (int)k - 1;
        if (a.leaf == FILTER_LEAF_MATCH) {                            // filters.rs:25-33: present and equal, never an error
            if (k && have == a.want_kind) {
                switch (have) {
                    case VDB_T_INT: pass = v == a.want_bits; break;
                    case VDB_T_FLOAT: pass = of64_cmp(__longlong_as_double(v), __longlong_as_double(a.want_bits)) == 0; break;
                    case VDB_T_BOOL: pass = (v != 0) == (a.want_bits != 0); break;
                    case VDB_T_STR: pass = (uint64_t)v < a.table_n && a.table[v] != 0; break;
                    default: break;                                   // list values: patched from the host
                }
            }
        } else {                                                      // payload.rs:36-114
            if (!k) err = true;                                       // "Missing field"
            else if (have == VDB_T_LIST_STR && (a.want_kind == VDB_T_LIST_STR || a.want_kind == VDB_T_STR)) { }   // payload.rs:50-90: patched from the host
            else if (have != a.want_kind) err = true;                 // "Type mismatch" (Int vs Float included)
            else {
                switch (have) {
                    case VDB_T_INT: pass = cmp_from_order(a.op, (v > a.want_bits) - (v < a.want_bits)); break;
                    case VDB_T_FLOAT: pass = cmp_from_order(a.op, of64_cmp(__longlong_as_double(v), __longlong_as_double(a.want_bits))); break;
                    case VDB_T_STR: pass = (uint64_t)v < a.table_n && a.table[v] != 0; break;
                    case VDB_T_BOOL:
                        if (a.op == VDB_OP_EQ || a.op == VDB_OP_NEQ) pass = cmp_from_order(a.op, (int)(v != 0) - (int)(a.want_bits != 0));
                        else err = true;                              // payload.rs:176-182
                        break;
                    default: err = true;                              // lists of equal kind: compare_scalar has no arm
                }
            }
        }
    }
    const uint64_t pm = __ballot(pass), em = __ballot(err);
    const uint32_t lane = threadIdx.x & 63u, w = (s >> 6) * 2;
    if (lane == 0 && s < a.n) {
        a.pass[w] = (uint32_t)pm; a.err[w] = (uint32_t)em;
        if (w + 1 < a.words) { a.pass[w + 1] = (uint32_t)(pm >> 32); a.err[w + 1] = (uint32_t)(em >> 32); }
    }
}

// Match through the inverted index: every slot of the posting list sets its bit (the bitmap starts at zero).
__global__ void __launch_bounds__(256) filter_scatter_kernel(const uint32_t* __restrict__ slots, uint32_t m, uint32_t* __restrict__ pass) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) atomicOr(&pass[slots[i] >> 5], 1u << (slots[i] & 31u));
}

// Outcomes computed on the host for a few slots (list-valued payload fields): 0 false, 1 true, 2 Err.
__global__ void __launch_bounds__(256) filter_patch_kernel(const uint32_t* __restrict__ slots, const uint8_t* __restrict__ tri, uint32_t m,
                                                           uint32_t* __restrict__ pass, uint32_t* __restrict__ err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t s = slots[i], bit = 1u << (s & 31u);
    if (tri[i] == 1) atomicOr(&pass[s >> 5], bit); else atomicAnd(&pass[s >> 5], ~bit);
    if (tri[i] == 2) atomicOr(&err[s >> 5], bit); else atomicAnd(&err[s >> 5], ~bit);
}

// acc <- acc OP b, word by word (filters.rs:46-73).  And: an Err or a false on the left decides, otherwise the right
// side; Or: an Err or a true on the left decides; Not keeps an Err and negates the rest.
__global__ void __launch_bounds__(256) filter_combine_kernel(int op, uint32_t* __restrict__ ap, uint32_t* __restrict__ ae, const uint32_t* __restrict__ bp,
                                                             const uint32_t* __restrict__ be, const uint32_t* __restrict__ has, uint32_t words) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    const uint32_t h = has[w];
    uint32_t p = ap[w], e = ae[w];
    switch (op) {
        case FILTER_AND: { const uint32_t q = bp[w], f = be[w]; e = e | (p & f); p = p & q & ~e; break; }
        case FILTER_OR: { const uint32_t q = bp[w], f = be[w]; const uint32_t open = h & ~p & ~e; p = p | (open & q); e = e | (open & f); break; }
        case FILTER_NOT: p = h & ~p & ~e; break;
        case FILTER_TRUE: p = h; e = 0; break;
        default: p = 0; e = 0; break;                                  // FILTER_FALSE
    }
    ap[w] = p; ae[w] = e;
}

// counts[0] += live passing slots, counts[1] += live slots whose evaluation is an Err; bits beyond n are cleared.
__global__ void __launch_bounds__(256) filter_count_kernel(uint32_t* __restrict__ pass, uint32_t* __restrict__ err, const uint32_t* __restrict__ deleted,
                                                           uint32_t n, uint32_t words, uint32_t* __restrict__ counts) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cp = 0, ce = 0;
    if (w < words) {
        const uint32_t tail = (w == words - 1 && (n & 31u)) ? ((1u << (n & 31u)) - 1u) : 0xFFFFFFFFu;
        const uint32_t p = pass[w] & tail, e = err[w] & tail, live = deleted ? ~deleted[w] : 0xFFFFFFFFu;
        pass[w] = p; err[w] = e;
        cp = __popc(p & live); ce = __popc(e & live);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { cp += __shfl_xor(cp, off); ce += __shfl_xor(ce, off); }
    if ((threadIdx.x & 63u) == 0 && (cp | ce)) { if (cp) atomicAdd(&counts[0], cp); if (ce) atomicAdd(&counts[1], ce); }
}

// ---- bitmap -> ascending list of the live passing slots (three passes: block sums, scan of the sums, expansion) ----
constexpr uint32_t CB = 256;                       // words per block
__device__ __forceinline__ uint32_t live_word(const uint32_t* pass, const uint32_t* deleted, uint32_t w, uint32_t words) {
    return w < words ? (pass[w] & (deleted ? ~deleted[w] : 0xFFFFFFFFu)) : 0u;
}
__global__ void __launch_bounds__(CB) compact_sums_kernel(const uint32_t* __restrict__ pass, const uint32_t* __restrict__ deleted, uint32_t words, uint32_t* __restrict__ sums) {
    __shared__ uint32_t part[CB / 64];
    uint32_t c = __popc(live_word(pass, deleted, blockIdx.x * CB + threadIdx.x, words));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// exclusive scan of `sums` in place by one block (the carry runs through chunks of 1024)
__global__ void __launch_bounds__(1024) compact_scan_kernel(uint32_t* __restrict__ sums, uint32_t nb) {
    __shared__ uint32_t buf[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? sums[i] : 0u;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0u;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nb) sums[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(CB) compact_expand_kernel(const uint32_t* __restrict__ pass, const uint32_t* __restrict__ deleted, uint32_t words,
                                                            const uint32_t* __restrict__ offsets, uint32_t* __restrict__ rows) {
    __shared__ uint32_t buf[CB];
    const uint32_t w = blockIdx.x * CB + threadIdx.x;
    uint32_t word = live_word(pass, deleted, w, words);
    const uint32_t c = __popc(word);
    buf[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t off = 1; off < CB; off <<= 1) {
        const uint32_t t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0u;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t pos = offsets[blockIdx.x] + buf[threadIdx.x] - c;
    while (word) { const uint32_t b = __ffs(word) - 1; word &= word - 1; rows[pos++] = w * 32u + b; }
}

}  // namespace

void launch_filter_leaf(const FilterLeafArgs& a, hipStream_t st) {
    if (a.n == 0) return;
    hipLaunchKernelGGL(filter_leaf_kernel, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
    VDB_HIP(hipGetLastError());
}
void launch_filter_scatter(const uint32_t* slots, uint32_t m, uint32_t* pass, hipStream_t st) {
    if (m == 0) return;
    hipLaunchKernelGGL(filter_scatter_kernel, dim3((m + 255) / 256), dim3(256), 0, st, slots, m, pass);
    VDB_HIP(hipGetLastError());
}
void launch_filter_patch(const uint32_t* slots, const uint8_t* tri, uint32_t m, uint32_t* pass, uint32_t* err, hipStream_t st) {
    if (m == 0) return;
    hipLaunchKernelGGL(filter_patch_kernel, dim3((m + 255) / 256), dim3(256), 0, st, slots, tri, m, pass, err);
    VDB_HIP(hipGetLastError());
}
void launch_filter_combine(int op, uint32_t* ap, uint32_t* ae, const uint32_t* bp, const uint32_t* be, const uint32_t* has, uint32_t words, hipStream_t st) {
    if (words == 0) return;
    hipLaunchKernelGGL(filter_combine_kernel, dim3((words + 255) / 256), dim3(256), 0, st, op, ap, ae, bp, be, has, words);
    VDB_HIP(hipGetLastError());
}
void launch_filter_count(uint32_t* pass, uint32_t* err, const uint32_t* deleted, uint32_t n, uint32_t* counts, hipStream_t st) {
    const uint32_t words = (n + 31) / 32;
    if (words == 0) return;
    hipLaunchKernelGGL(filter_count_kernel, dim3((words + 255) / 256), dim3(256), 0, st, pass, err, deleted, n, words, counts);
    VDB_HIP(hipGetLastError());
}
uint32_t filter_compact_blocks(uint32_t n) { return ((n + 31) / 32 + CB - 1) / CB; }
void launch_filter_compact(const uint32_t* pass, const uint32_t* deleted, uint32_t n, uint32_t* block_sums, uint32_t* rows, hipStream_t st) {
    const uint32_t words = (n + 31) / 32, nb = filter_compact_blocks(n);
    if (nb == 0) return;
    hipLaunchKernelGGL(compact_sums_kernel, dim3(nb), dim3(CB), 0, st, pass, deleted, words, block_sums);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, block_sums, nb);
    hipLaunchKernelGGL(compact_expand_kernel, dim3(nb), dim3(CB), 0, st, pass, deleted, words, block_sums, rows);
    VDB_HIP(hipGetLastError());
}

}  // namespace vdb
