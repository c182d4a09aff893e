// Host side of the relax step, general form (reference: include/opencalibration/relax/relax.hpp:12-15,
// src/relax/relax.cpp:89-134, src/relax/relax_problem.cpp:83-120): the {ORIENTATION, GROUND_MESH} flavour every pipeline
// state after INITIAL_PROCESSING runs - surface mesh with per-vertex heights, 3..5-ray track blocks, 2-ray blocks for the
// image cells the tracks leave uncovered, mesh priors - and its intrinsics variants.  Problem assembly stays on the host
// as in the reference (pose lookup, union-find tracks, grid filters, mesh walk); the Ceres solve is ochip_relaxg_solve.
#pragma once

#include "relax.hpp"

#include <array>
#include <unordered_map>

namespace opencalibration_amd
{

// include/opencalibration/types/mesh_graph.hpp, as index-addressed arrays (node / edge ids = insertion indices; the
// reference iterates its mesh in insertion order)
struct MeshNode
{
    double location[3];
};
struct MeshEdge
{
    static constexpr size_t NONE = (size_t)-1;
    size_t source = NONE, dest = NONE;
    bool border = false;
    size_t triangleOppositeNodes[2] = {NONE, NONE};
};
class MeshGraph
{
  public:
    size_t addNode(double x, double y, double z)
    {
        nodes.push_back(MeshNode{{x, y, z}});
        node_edges.emplace_back();
        return nodes.size() - 1;
    }
    size_t addEdge(MeshEdge e, size_t source, size_t dest)
    {
        e.source = source;
        e.dest = dest;
        edges.push_back(e);
        _lookup.emplace(key(source, dest), edges.size() - 1);
        if (source < node_edges.size())
            node_edges[source].push_back(edges.size() - 1);
        if (dest < node_edges.size() && dest != source)
            node_edges[dest].push_back(edges.size() - 1);
        return edges.size() - 1;
    }
    // mesh refinement (refine_mesh.cpp) removes and adds edges: it rewrites edges / node_edges and then calls this
    void rebuild_lookup()
    {
        _lookup.clear();
        for (size_t i = 0; i < edges.size(); i++)
            if (edges[i].source != MeshEdge::NONE)
                _lookup.emplace(key(edges[i].source, edges[i].dest), i);
    }
    void forget_edge(size_t source, size_t dest)
    {
        _lookup.erase(key(source, dest));
    }
    const MeshEdge *getEdge(size_t source, size_t dest) const
    {
        auto it = _lookup.find(key(source, dest));
        return it == _lookup.end() ? nullptr : &edges[it->second];
    }
    MeshEdge *getEdge(size_t source, size_t dest)
    {
        auto it = _lookup.find(key(source, dest));
        return it == _lookup.end() ? nullptr : &edges[it->second];
    }
    size_t size_nodes() const
    {
        return nodes.size();
    }
    size_t size_edges() const
    {
        return edges.size();
    }
    std::vector<MeshNode> nodes;
    std::vector<MeshEdge> edges;
    // Node::getEdges(): the ids of the edges at a vertex in the order the reference's ankerl::unordered_dense set holds
    // them (insertion order; an erase moves the last element into the hole)
    std::vector<std::vector<size_t>> node_edges;

  private:
    static uint64_t key(size_t s, size_t d)
    {
        return ((uint64_t)s << 32) ^ (uint64_t)d;
    }
    std::unordered_map<uint64_t, size_t> _lookup;
};

using point_cloud = std::vector<std::array<double, 3>>;
struct surface_model // include/opencalibration/types/surface_model.hpp
{
    std::vector<point_cloud> cloud;
    MeshGraph mesh;
};

// include/opencalibration/types/relax_options.hpp:9-33 as bits
enum RelaxOption : uint32_t
{
    OPT_ORIENTATION = 1u << 0,
    OPT_POSITION = 1u << 1,
    OPT_GROUND_PLANE = 1u << 2,
    OPT_GROUND_MESH = 1u << 3,
    OPT_POINTS_3D = 1u << 4,
    OPT_FOCAL_LENGTH = 1u << 5,
    OPT_PRINCIPAL_POINT = 1u << 6,
    OPT_LENS_DISTORTIONS_RADIAL = 1u << 7,
    OPT_LENS_DISTORTIONS_RADIAL_BROWN2_PARAMETERIZATION = 1u << 8,
    OPT_LENS_DISTORTIONS_RADIAL_BROWN24_PARAMETERIZATION = 1u << 9,
    OPT_LENS_DISTORTIONS_RADIAL_BROWN246_PARAMETERIZATION = 1u << 10,
    OPT_LENS_DISTORTIONS_TANGENTIAL = 1u << 11,
    OPT_MINIMAL_MESH = 1u << 12,
};
struct RelaxConfig
{
    uint32_t options = 0;
    double ground_mesh_grid_fraction = 0.1;
};

struct RelaxMeshStats // what the last problem looked like (not in the reference)
{
    int track_blocks = 0, two_ray_blocks = 0, mesh_vertices = 0, unknowns = 0;
};

// src/surface/expand_mesh.cpp:17-380.  Nearest-neighbour queries are exact; among exactly equidistant points the one
// inserted first wins.
MeshGraph rebuildMesh(const point_cloud &cameraLocations, const std::vector<surface_model> &previousSurfaces);
MeshGraph buildMinimalMesh(const point_cloud &cameraLocations, const std::vector<surface_model> &previousSurfaces);

// relax(graph, nodes, cam_models, edges_to_optimize, config, previousSurfaces) (relax.hpp:12-15).  cam_models: the group's
// copies of the camera models by id, in insertion order.  Flavours: GROUND_MESH (this file) and GROUND_PLANE (relax.cpp).
// Returns false with `error` set when the device reports an error; poses are then untouched.
// shard: one process per GPU, every rank calling relax() on the same graph with the same arguments - the evaluation of
// the residual blocks is cut over the ranks and `shard->exchange` all-gathers the records (ochip_relaxg_set_exchange,
// ochip_relax_set_shard); every rank gets the same, bit-identical result.
bool relax(ochip_ctx *ctx, const MeasurementGraph &graph, std::vector<NodePose> &nodes,
           std::vector<std::pair<size_t, CameraModel>> &cam_models, const std::vector<size_t> &edges_to_optimize,
           const RelaxConfig &config, const std::vector<surface_model> &previousSurfaces, surface_model *surface,
           RelaxTimers *timers, RelaxMeshStats *stats, std::string *error, const RelaxShard *shard = nullptr);

// runPoints (src/relax/relax.cpp:103-115) on the device: setup3dPointProblem, relaxObservedModelOnly, solve
// (host/relax_points.cpp over ochip_relaxp_*).  mode < 0: that driver; 0: set-up only, 1: solve, 2: relaxObservedModelOnly
// (TestRelaxProblem of test/test_relax.cpp:470-483).  points_before / points_after: the tracks' 3-D points (xyz) after the
// set-up / at the end.  The surface is the point cloud.
bool relax_points(ochip_ctx *ctx, const MeasurementGraph &graph, std::vector<NodePose> &nodes,
                  std::vector<std::pair<size_t, CameraModel>> &cam_models, const std::vector<size_t> &edges_to_optimize,
                  uint32_t options, surface_model *surface, RelaxTimers *timers, std::string *error, int mode = -1,
                  std::vector<double> *points_before = nullptr, std::vector<double> *points_after = nullptr);

// runRelativeOrientation (src/relax/relax.cpp:14-42) on the device: setupDecompositionProblem with the NaN-orientation
// bootstrap (host/relax_relative.cpp over ochip_relaxg_* relation blocks)
bool relax_relative_orientation(ochip_ctx *ctx, const MeasurementGraph &graph, std::vector<NodePose> &nodes,
                                const std::vector<size_t> &edges_to_optimize, RelaxTimers *timers, std::string *error);

} // namespace opencalibration_amd
