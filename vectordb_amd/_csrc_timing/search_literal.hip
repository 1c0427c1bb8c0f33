// Instantiations of the search kernel for the literal in-place filtered search (hnsw.rs:585-689).
#include "search_kernel.hpp"

namespace vdb {

void launch_search_literal(const SearchArgs& a, size_t lds, hipStream_t st) {
    if (a.arith == VDB_ARITH_STRICT) { launch_search_strict<MODE_LITERAL>(a, lds, st); return; }
    launch_search_fast<MODE_LITERAL, -1>(a, lds, st);
}

}  // namespace vdb
