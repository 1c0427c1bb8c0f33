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

}  // namespace vdb
