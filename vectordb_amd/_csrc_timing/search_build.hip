// Instantiations of the search kernel for graph construction (the searches of HNSWIndex::insert, hnsw.rs:169-199).
#include "search_kernel.hpp"

namespace vdb {

void launch_search_build(const SearchArgs& a, size_t lds, hipStream_t st) {
    if (a.arith == VDB_ARITH_STRICT) { launch_search_strict<MODE_BUILD>(a, lds, st); return; }
    launch_search_fast<MODE_BUILD, -1>(a, lds, st);
}

}  // namespace vdb
