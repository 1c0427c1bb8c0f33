// Instantiations of the search kernel for the unfiltered batched search (HNSWIndex::search, hnsw.rs:513-554), metric fixed at compile time.
#include "search_kernel.hpp"

namespace vdb {

void launch_search_plain(const SearchArgs& a, size_t lds, hipStream_t st) {
    if (a.arith == VDB_ARITH_STRICT) { launch_search_strict<MODE_SEARCH>(a, lds, st); return; }
    switch (a.g.metric) {
        case VDB_COSINE: launch_search_fast<MODE_SEARCH, VDB_COSINE>(a, lds, st); break;
        case VDB_DOT: launch_search_fast<MODE_SEARCH, VDB_DOT>(a, lds, st); break;
        default: launch_search_fast<MODE_SEARCH, VDB_EUCLIDEAN>(a, lds, st); break;
    }
}

namespace {
// Descent-only launch: a few hundred bytes of LDS and few registers per wavefront, so many more
// wavefronts are resident per CU than the beam kernel allows and the dependent round trips of the
// upper layers overlap across queries.
template <int METRIC>
void launch_descend_shape(const SearchArgs& a, size_t lds, hipStream_t st) {
    const uint32_t chunks = a.g.stride / 4;
    if (chunks <= 32) hipLaunchKernelGGL((search_kernel<MODE_DESCEND, 8, 4, 2, false, METRIC, 0>), dim3(a.nq), dim3(64), lds, st, a);
    else if (chunks <= 256) hipLaunchKernelGGL((search_kernel<MODE_DESCEND, 64, 4, 8, false, METRIC, 0>), dim3(a.nq), dim3(64), lds, st, a);
    else if (chunks <= 512) hipLaunchKernelGGL((search_kernel<MODE_DESCEND, 64, 8, 4, false, METRIC, 0>), dim3(a.nq), dim3(64), lds, st, a);
    else throw Failure(err_arg("dimension above 2048 is not supported"));
}
}  // namespace

void launch_search_descend(const SearchArgs& a, size_t lds, hipStream_t st) {
    if (a.arith != VDB_ARITH_FAST || !a.g.ublk || !a.entry_out) throw Failure(err_arg("descent launch needs FAST arithmetic and the fused upper-layer blocks"));
    switch (a.g.metric) {
        case VDB_COSINE: launch_descend_shape<VDB_COSINE>(a, lds, st); break;
        case VDB_DOT: launch_descend_shape<VDB_DOT>(a, lds, st); break;
        default: launch_descend_shape<VDB_EUCLIDEAN>(a, lds, st); break;
    }
}

}  // namespace vdb
