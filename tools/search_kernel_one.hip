// development aid: only the benchmark's instantiation of the search kernel
#include "search_kernel.hpp"
namespace vdb {
void launch_one(const SearchArgs& a, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((search_kernel<MODE_SEARCH, 8, 4, 2, false, VDB_EUCLIDEAN, 3>), dim3(a.nq), dim3(64), lds, st, a);
}
}
