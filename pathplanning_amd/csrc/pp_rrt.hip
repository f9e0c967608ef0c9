// RRT / RRT* entry points (placeholder until the persistent tree kernel lands in this file).
#include "pp_internal.hpp"

struct pp_rrt {
	std::vector<double> nodes, costs, path;
	std::vector<int32_t> parents;
};

extern "C" {
int pp_rrt_run(pp_ctx*, pp_map*, const double*, const double*, const double*, const double*, const double*, uint64_t, int32_t, pp_rrt**, pp_rrt_result*)
{
	pph::set_error("pp_rrt_run: not built yet");
	return PP_ERR_INVALID;
}
int pp_rrt_get(pp_rrt*, double*, int32_t*, double*, double*)
{
	pph::set_error("pp_rrt_get: not built yet");
	return PP_ERR_INVALID;
}
int pp_rrt_destroy(pp_rrt* r)
{
	delete r;
	return PP_OK;
}
}
