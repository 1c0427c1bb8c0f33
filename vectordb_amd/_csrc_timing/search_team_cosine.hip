// Team-of-wavefronts search kernels (search_team.hpp) for the cosine metric.
#include "search_team.hpp"

namespace vdb {

void launch_search_team_cosine(const SearchArgs& a, hipStream_t st) { launch_team_metric<VDB_COSINE>(a, st); }

}  // namespace vdb
