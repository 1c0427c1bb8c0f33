// Instantiations of the search kernel for the bitmap-masked filtered search (extension).
#include "search_kernel.hpp"

namespace vdb {

void launch_search_masked(const SearchArgs& a, size_t lds, hipStream_t st) {
    if (a.arith == VDB_ARITH_STRICT) { launch_search_strict<MODE_MASKED>(a, lds, st); return; }
    launch_search_fast<MODE_MASKED, -1>(a, lds, st);
}

}  // namespace vdb
