// ORACLE — test infrastructure only (see oracle.hpp).
//
// The relax stage restated in full over a small in-memory MeasurementGraph:
//   src/relax/relax.cpp:14-134                  relax() and the four run* drivers
//   src/relax/relax_problem.cpp:21-1507         RelaxProblem (all four problem flavours, priors, tracks, solve,
//                                               getSurfaceModel)
//   src/relax/relax_group.cpp:14-182            RelaxGroup::init / run / finalize
//   src/surface/intersect.cpp:10-163            MeshIntersectionSearcher
//   src/surface/expand_mesh.cpp:17-380          rebuildMesh / buildMinimalMesh
//   src/geometry/intersection.cpp:116-143       rayIntersection
//   src/distort/invert_distortion.cpp:105-191   convertModel (forward <-> inverse lens model)
//   include/opencalibration/types/{graph,mesh_graph,feature_track,union_find,surface_model}.hpp,
//   include/opencalibration/relax/grid_filter.hpp
// Graph ids: the reference draws random size_t ids (graph.hpp:74-84); every container it iterates is an
// ankerl::unordered_dense map/set, which iterates in insertion order, so ids here are plain
// insertion indices and every loop below runs in insertion order.
#pragma once

#include "mini_ceres.hpp"
#include "oracle.hpp"

#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace oracle
{
namespace rx
{

constexpr size_t NONE = (size_t)-1;

struct CameraModel : camera_model // types/camera_model.hpp:68-83
{
    size_t id = 0;
};
inline bool same_model(const CameraModel &a, const CameraModel &b) // CameraModel::operator== (:78-81, :37-43)
{
    return a.id == b.id && a.pixels_rows == b.pixels_rows && a.pixels_cols == b.pixels_cols &&
           a.focal_length_pixels == b.focal_length_pixels && a.principle_point[0] == b.principle_point[0] &&
           a.principle_point[1] == b.principle_point[1] && a.radial_distortion[0] == b.radial_distortion[0] &&
           a.radial_distortion[1] == b.radial_distortion[1] && a.radial_distortion[2] == b.radial_distortion[2] &&
           a.tangential_distortion[0] == b.tangential_distortion[0] && a.tangential_distortion[1] == b.tangential_distortion[1];
}

struct relation // types/camera_relations.hpp:13-35
{
    std::vector<feature_match_denormalized> inlier_matches;
    std::vector<feature_match> matches;
    Mat3 ransac_relation;
    bool is_homography = false; // RelationType::HOMOGRAPHY, else UNKNOWN
    std::array<decomposed_pose, 4> relative_poses;
};
struct image_node // types/image.hpp (fields the relax stage reads)
{
    std::string path;
    Vec3 position{NAN, NAN, NAN};
    Quat orientation;
    std::shared_ptr<CameraModel> model;
    std::vector<Vec2> feature_location; // features[i].location
    std::vector<size_t> edges;          // Node::getEdges(), insertion order
};
struct graph_edge
{
    size_t source, dest;
    relation payload;
};
struct MeasurementGraph // types/graph.hpp DirectedGraph<image, camera_relations>
{
    std::vector<image_node> nodes;
    std::vector<graph_edge> edges;
    size_t addNode(image_node n)
    {
        nodes.push_back(std::move(n));
        return nodes.size() - 1;
    }
    size_t addEdge(relation r, size_t source, size_t dest)
    {
        edges.push_back(graph_edge{source, dest, std::move(r)});
        nodes[source].edges.push_back(edges.size() - 1);
        if (dest != source)
            nodes[dest].edges.push_back(edges.size() - 1);
        return edges.size() - 1;
    }
    const image_node *getNode(size_t id) const
    {
        return id < nodes.size() ? &nodes[id] : nullptr;
    }
    image_node *getNode(size_t id)
    {
        return id < nodes.size() ? &nodes[id] : nullptr;
    }
    const graph_edge *getEdge(size_t id) const
    {
        return id < edges.size() ? &edges[id] : nullptr;
    }
};

struct mesh_node // types/mesh_graph.hpp:12-20
{
    Vec3 location;
};
struct mesh_edge // :22-31 + Edge source/dest
{
    size_t source = NONE, dest = NONE;
    bool border = false;
    size_t opposite[2] = {NONE, NONE}; // triangleOppositeNodes (zero-initialised = "no node" in the reference)
};
struct MeshGraph
{
    std::vector<mesh_node> nodes;
    std::vector<mesh_edge> edges;
    std::map<std::pair<size_t, size_t>, size_t> lookup;
    size_t addNode(const Vec3 &p)
    {
        nodes.push_back(mesh_node{p});
        return nodes.size() - 1;
    }
    size_t addEdge(mesh_edge e, size_t source, size_t dest)
    {
        e.source = source;
        e.dest = dest;
        edges.push_back(e);
        lookup.emplace(std::make_pair(source, dest), edges.size() - 1);
        return edges.size() - 1;
    }
    const mesh_edge *getEdge(size_t s, size_t d) const
    {
        auto it = lookup.find(std::make_pair(s, d));
        return it == lookup.end() ? nullptr : &edges[it->second];
    }
    mesh_edge *getEdge(size_t s, size_t d)
    {
        auto it = lookup.find(std::make_pair(s, d));
        return it == lookup.end() ? nullptr : &edges[it->second];
    }
    size_t size_nodes() const
    {
        return nodes.size();
    }
    size_t size_edges() const
    {
        return edges.size();
    }
};
using point_cloud = std::vector<Vec3>;
struct surface_model // types/surface_model.hpp
{
    std::vector<point_cloud> cloud;
    MeshGraph mesh;
};

struct NodePose // types/node_pose.hpp
{
    size_t node_id;
    Quat orientation;
    Vec3 position;
};

enum Option : uint32_t // types/relax_options.hpp:9-33, as bits
{
    ORIENTATION = 1u << 0,
    POSITION = 1u << 1,
    GROUND_PLANE = 1u << 2,
    GROUND_MESH = 1u << 3,
    POINTS_3D = 1u << 4,
    FOCAL_LENGTH = 1u << 5,
    PRINCIPAL_POINT = 1u << 6,
    LENS_DISTORTIONS_RADIAL = 1u << 7,
    LENS_DISTORTIONS_RADIAL_BROWN2_PARAMETERIZATION = 1u << 8,
    LENS_DISTORTIONS_RADIAL_BROWN24_PARAMETERIZATION = 1u << 9,
    LENS_DISTORTIONS_RADIAL_BROWN246_PARAMETERIZATION = 1u << 10,
    LENS_DISTORTIONS_TANGENTIAL = 1u << 11,
    MINIMAL_MESH = 1u << 12,
};
struct RelaxOptionSet
{
    uint32_t bits = 0;
    bool get(uint32_t o) const
    {
        return (bits & o) != 0;
    }
    bool hasAll(uint32_t o) const
    {
        return (bits & o) == o;
    }
    bool hasAny(uint32_t o) const
    {
        return (bits & o) != 0;
    }
};
struct RelaxConfig
{
    RelaxOptionSet options;
    double ground_mesh_grid_fraction = 0.1;
};

struct relax_stats // not in the reference: what the solves did, for parity checks
{
    int solves = 0, iterations_total = 0, last_iterations = 0, last_residual_blocks = 0, last_parameter_blocks = 0;
    double last_initial_cost = 0, last_final_cost = 0;
    int track_blocks = 0, two_ray_blocks = 0; // of the last problem set up
    std::vector<int> iterations_per_solve;
};

using model_map = std::vector<std::pair<size_t, CameraModel>>; // ankerl map<size_t, CameraModel>: insertion order

surface_model relax(const MeasurementGraph &graph, std::vector<NodePose> &nodes, model_map &cam_models,
                    const std::vector<size_t> &edges_to_optimize, const RelaxConfig &config,
                    const std::vector<surface_model> &previousSurfaces, relax_stats *stats = nullptr);
// TestRelaxProblem of test/test_relax.cpp:470-483 (setup3dPointProblem, then solve / relaxObservedModelOnly, with the tracks'
// points before and after): mode 0 = set-up only, 1 = + solve, 2 = + relaxObservedModelOnly
void points_problem_steps(const MeasurementGraph &graph, std::vector<NodePose> &nodes, model_map &cam_models,
                          const std::vector<size_t> &edges_to_optimize, const RelaxOptionSet &options, int mode,
                          std::vector<Vec3> *points_before, std::vector<Vec3> *points_after, relax_stats *stats);

// RelaxStage::init (src/pipeline/relax_stage.cpp:28-112): the primary node ids of every group, largest group first
// (oracle/relax_cluster.cpp: k-means / spectral clustering restated)
std::vector<std::vector<size_t>> relax_stage_groups(const MeasurementGraph &graph, const std::vector<size_t> &node_ids,
                                                    bool relax_all, bool disable_parallelism, uint32_t options,
                                                    size_t *graph_connection_depth);

// src/surface/expand_mesh.cpp
MeshGraph rebuildMesh(const point_cloud &cameraLocations, const std::vector<surface_model> &previousSurfaces);
MeshGraph buildMinimalMesh(const point_cloud &cameraLocations, const std::vector<surface_model> &previousSurfaces);

// src/distort/invert_distortion.cpp:105-191 (models carry a tag in the reference; here the caller knows which is which)
camera_model convertModelToInverse(const camera_model &standardModel);
camera_model convertModelToForward(const camera_model &invertedModel);
Vec3 image_to_3d_inverse_model(const double keypoint[2], const camera_model &inverse_model);

// src/surface/intersect.cpp
class MeshIntersectionSearcher
{
  public:
    enum Type
    {
        UNINITIALIZED,
        PENDING,
        INTERSECTION,
        OUTSIDE_BORDER,
        RAY_PARALLEL_TO_PLANE,
        GRAPH_STRUCTURE_INCONSISTENT
    };
    struct IntersectionInfo
    {
        Type type = PENDING;
        size_t nodeIndexes[3] = {0, 0, 0};
        const Vec3 *nodeLocations[3] = {nullptr, nullptr, nullptr};
        Vec3 intersectionLocation{NAN, NAN, NAN};
        size_t steps = 0;
    };
    bool init(const MeshGraph &meshGraph);
    const IntersectionInfo &triangleIntersect(const Vec3 &dir, const Vec3 &offset);

  private:
    const MeshGraph *_meshGraph = nullptr;
    IntersectionInfo _info;
};

// src/relax/relax_group.cpp.  imageGPSLocations.searchKnn(position, 10) arrives as a table (n_nodes x 10 node ids,
// NONE padded): the reference's jk::KDTree is compiled only into oracle/_ref.
class RelaxGroup
{
  public:
    void init(const MeasurementGraph &graph, const std::vector<size_t> &node_ids, const std::vector<size_t> &knn10,
              size_t graph_connection_depth, const RelaxConfig &config);
    surface_model run(const MeasurementGraph &graph, const std::vector<surface_model> &previousSurfaces,
                      relax_stats *stats = nullptr);
    std::vector<size_t> finalize(MeasurementGraph &graph);

    std::vector<NodePose> _local_poses;
    model_map _camera_models;
    std::vector<size_t> _edges_to_optimize; // insertion-ordered set
    RelaxConfig _config;

  private:
    std::unordered_set<size_t> _edges_set, _nodes_to_optimize;
    std::vector<size_t> _directly_connected;
    std::unordered_set<size_t> _directly_set;
    void build_optimization_edges(const MeasurementGraph &graph, const std::vector<size_t> &knn10, size_t node_id);
};

} // namespace rx
} // namespace oracle
