// Mesh refinement by point density (reference: include/opencalibration/surface/refine_mesh.hpp,
// src/surface/refine_mesh.cpp:15-909) and the MESH_REFINEMENT state that alternates it with the
// ground-mesh relax (src/pipeline/pipeline.cpp:666-819).
//
// The reference's mesh is a DirectedGraph over ankerl::unordered_dense maps / sets: iteration runs in insertion order,
// and erasing an element moves the LAST element into its place.  Which sub-triangle a re-located triangle becomes
// (findTriangleNearVertices walks a vertex's edge set and the centroid of a split triangle lies on the splitting median)
// and the order in which triangles are refined depend on exactly that order, so it is kept: a vertex's edge list
// (MeshGraph::node_edges) and the edge iteration order (kept here while edges are removed, written back by compaction:
// the edge ids a refinement leaves behind are 0 .. n-1 in iteration order; ids are opaque in the reference).
#pragma once

#include "relax_mesh.hpp"
#include "relax_stage.hpp"

namespace opencalibration_amd
{

struct TriangleId // refine_mesh.hpp:30-39: one of the triangle's edges and the side of it (which triangleOppositeNode)
{
    size_t edgeId = MeshEdge::NONE; // the reference's "0 = no triangle" is NONE here (0 is a valid index)
    int side = 0;
    bool operator==(const TriangleId &o) const
    {
        return edgeId == o.edgeId && side == o.side;
    }
};
struct TrianglePointStats // refine_mesh.hpp
{
    size_t count = 0;
    double distanceVariance = 0;
};

// TriangleLocator (refine_mesh.cpp:572-711): the triangle under a point - from the triangle with the nearest centroid
// across the most violated edge, at most 100 steps, then the exhaustive scan.
class TriangleLocator
{
  public:
    explicit TriangleLocator(const MeshGraph &m, const std::vector<size_t> *edge_order = nullptr);
    TriangleId find(double x, double y) const;
    bool vertices(const TriangleId &t, size_t v[3]) const;

  private:
    TriangleId brute_force(double x, double y) const;
    const MeshGraph &_m;
    std::vector<size_t> _order;
    std::vector<TriangleId> _tri;
    std::vector<double> _cx, _cy;
    // bucket grid over the centroids
    double _x0 = 0, _y0 = 0, _cell = 1;
    int _nx = 1, _ny = 1;
    std::vector<uint32_t> _start, _items;
};

// countPointsPerTriangle (:713-825): per triangle that holds points, their number and the variance of their signed
// distance to the triangle's plane.  Triangles in the order their first point appears; sums in point order (what the
// reference does with one thread - with more its accumulation order is whatever the threads' finishing order is).
std::vector<std::pair<TriangleId, TrianglePointStats>> countPointsPerTriangle(const MeshGraph &mesh, const std::vector<point_cloud> &points);

// refineByPointDensity (:827-909): bisects (longest edge, neighbour first) every triangle with more than
// maxPointsPerTriangle points whose distance variance exceeds minDistanceVariance and whose longest side is at least
// minTriangleSizeMeters.  Returns the number of triangles created.
size_t refineByPointDensity(MeshGraph &mesh, const std::vector<point_cloud> &points, size_t maxPointsPerTriangle,
                            double minDistanceVariance = 0.0, int maxIterations = 10, double minTriangleSizeMeters = 0.0);
// refineAtPoint (:452-473): `levels` rounds of refining the triangle under (x, y)
size_t refineAtPoint(MeshGraph &mesh, double x, double y, int levels = 1);

// Pipeline::Impl::mesh_refinement (pipeline.cpp:666-819) as a step function over the state the pipeline keeps.
struct MeshRefinementState
{
    uint64_t run_count = 0; // stateRunCount()
    int grid_level = 0;
    size_t level_triangles = 0;
    // diagnostics of the last step (not in the reference)
    double gsd = 0, grid_fraction = 0, reduced_gsd = 0;
    size_t triangles_above_threshold = 0, max_points = 0, refined = 0;
};
enum class Transition
{
    REPEAT,
    NEXT
};
// One run of the state: (first run: minimal mesh from the camera positions) -> RelaxStage over all cameras with
// {ORIENTATION, GROUND_MESH} and grid fraction 0.1 / 2^level -> count -> refine or advance the grid level.  `surfaces`
// is the pipeline's surface list; `stage` its RelaxStage (keeps the previous surfaces).  false + error on a device error.
bool mesh_refinement_step(ochip_ctx *ctx, MeasurementGraph &graph, std::vector<surface_model> &surfaces, RelaxStage &stage,
                          MeshRefinementState &state, Transition *transition, std::string *error);

} // namespace opencalibration_amd
