// Test shim (CPU): exposes the product's host brushfire (pathplanning_amd/csrc/pp_brushfire_host.hpp, the reference-order mode
// of pp_map_update_gvd_ex) through a C interface so that tests/test_brushfire_host.py can compare it with the oracle without a GPU.
#include "../../pathplanning_amd/csrc/pp_brushfire_host.hpp"

extern "C" {
void* bf_create(int rows, int cols) { return new pph::GvdReference(rows, cols); }
void bf_destroy(void* h) { delete (pph::GvdReference*)h; }
/// n (cell, value) pairs in order: AddObstacle's / RemoveObstacle's loop bodies
void bf_edit(void* h, int n, const int* cellValue)
{
	auto* g = (pph::GvdReference*)h;
	for (int i = 0; i < n; i++)
		g->edit(cellValue[2 * i], cellValue[2 * i + 1]);
}
long long bf_update(void* h)
{
	auto* g = (pph::GvdReference*)h;
	g->update();
	return g->pops;
}
void bf_get(void* h, int* d2, int* source, int* edgeD2, int* edgeSource)
{
	auto* g = (pph::GvdReference*)h;
	const size_t n = g->occ.size();
	for (size_t i = 0; i < n; i++) {
		d2[i] = g->obstacles.dist[i];
		source[i] = g->obstacles.source[i];
		edgeD2[i] = g->edges.dist[i];
		edgeSource[i] = g->edges.source[i];
	}
}
}
