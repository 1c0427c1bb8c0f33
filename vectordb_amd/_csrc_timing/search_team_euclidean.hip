// Team-of-wavefronts search kernels (search_team.hpp) for the euclidean metric.
#include "search_team.hpp"

namespace vdb {

void launch_search_team_euclidean(const SearchArgs& a, hipStream_t st) { launch_team_metric<VDB_EUCLIDEAN>(a, st); }

}  // namespace vdb
