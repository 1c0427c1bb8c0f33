// Team-of-wavefronts search kernels (search_team.hpp) for the dot metric.
#include "search_team.hpp"

namespace vdb {

void launch_search_team_dot(const SearchArgs& a, hipStream_t st) { launch_team_metric<VDB_DOT>(a, st); }

}  // namespace vdb
