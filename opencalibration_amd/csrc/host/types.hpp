// Host-side types of the MI355X hot path.  Same names, fields and meaning as the reference's
// include/opencalibration/types/*.hpp so that its stage code reads the same; Eigen members become
// plain arrays (Eigen is not available in this image) with identical memory layout.
#pragma once

#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

namespace opencalibration_amd
{

struct feature_2d // types/feature_2d.hpp:9-21 — 88 bytes, descriptor = std::bitset<486> words
{
    static constexpr int DESCRIPTOR_BITS = 486;
    double location[2] = {NAN, NAN};
    float strength = 0;
    uint64_t descriptor[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
static_assert(sizeof(feature_2d) == 88, "feature_2d must keep the reference's 88-byte layout");

struct feature_match // types/feature_match.hpp:11-23
{
    size_t feature_index_1, feature_index_2;
    double distance;
};

struct feature_match_denormalized // types/feature_match.hpp:26-39
{
    double pixel_1[2] = {NAN, NAN}, pixel_2[2] = {NAN, NAN};
    size_t feature_index_1 = 0, feature_index_2 = 0, match_index = 0;
};

struct correspondence // types/correspondence.hpp:8-13
{
    double measurement1[3], measurement2[3];
    double quality = 0;
};
static_assert(sizeof(correspondence) == 56, "correspondence layout");

struct decomposed_pose // types/decomposed_pose.hpp:7-21
{
    double orientation[4] = {NAN, NAN, NAN, NAN}; // x y z w (Eigen coeffs order)
    double position[3] = {NAN, NAN, NAN};
    int score = 0;
};

struct camera_relations // types/camera_relations.hpp:13-35
{
    std::vector<feature_match_denormalized> inlier_matches;
    std::vector<feature_match> matches;
    double ransac_relation[9] = {NAN, NAN, NAN, NAN, NAN, NAN, NAN, NAN, NAN}; // row-major 3x3
    enum class RelationType
    {
        HOMOGRAPHY,
        FUNDAMENTAL_MATRIX,
        UNKNOWN
    } relationType = RelationType::UNKNOWN;
    std::array<decomposed_pose, 4> relative_poses;
};

struct CameraModel // types/camera_model.hpp:22-81 (PLANAR projection)
{
    size_t pixels_rows = 0, pixels_cols = 0;
    double focal_length_pixels = 0;
    double principle_point[2] = {0, 0};
    double radial_distortion[3] = {0, 0, 0};
    double tangential_distortion[2] = {0, 0};
    size_t id = 0;
    bool has_distortion() const
    {
        return radial_distortion[0] != 0 || radial_distortion[1] != 0 || radial_distortion[2] != 0 ||
               tangential_distortion[0] != 0 || tangential_distortion[1] != 0;
    }
};

struct image // types/image.hpp:17-33 (fields the hot path touches)
{
    std::string path;
    std::vector<feature_2d> features;
    size_t num_sparse_features = 0;
    // cache (not a field of the reference's type): spatially_subsample_feature_indices(features, coarse_spacing,
    // num_sparse_features) as the device computed it with the feature list; coarse_spacing == 0: none.  Anything that
    // changes `features` must reset it.
    std::vector<uint32_t> coarse_subset;
    double coarse_spacing = 0;
    std::shared_ptr<CameraModel> model;
    double position[3] = {NAN, NAN, NAN};
    double orientation[4] = {NAN, NAN, NAN, NAN};
    // carried through graph.json untouched (graph_io.cpp): the base64 PNG thumbnail and the "metadata" object's text
    std::string thumbnail_b64, metadata_json;
};

struct NodeLinks // types/node_links.hpp
{
    size_t node_id;
    std::vector<size_t> link_ids;
};

// types/graph.hpp DirectedGraph<image, camera_relations>: random 64-bit ids from a default-seeded
// std::default_random_engine (graph.hpp:74-84,287-288), iteration in insertion order (the
// ankerl::unordered_dense behaviour the reference relies on).
class MeasurementGraph
{
  public:
    struct Node
    {
        size_t id;
        image payload;
        std::vector<size_t> edges;
    };
    struct Edge
    {
        size_t id, source, dest;
        camera_relations payload;
    };

    size_t addNode(image &&payload)
    {
        size_t identifier = _distribution(_generator);
        while (_node_index.count(identifier) > 0)
            identifier = _distribution(_generator);
        _node_index.emplace(identifier, _nodes.size());
        _nodes.push_back(Node{identifier, std::move(payload), {}});
        return identifier;
    }
    size_t addEdge(camera_relations &&payload, size_t source, size_t dest)
    {
        size_t identifier = _distribution(_generator);
        while (_edge_index.count(identifier) > 0)
            identifier = _distribution(_generator);
        _edge_index.emplace(identifier, _edges.size());
        _edges.push_back(Edge{identifier, source, dest, std::move(payload)});
        _nodes[_node_index.at(source)].edges.push_back(identifier);
        _nodes[_node_index.at(dest)].edges.push_back(identifier);
        return identifier;
    }
    // deserialisation (io/deserialize_MeasurementGraph.cpp:210,268-269): ids and the nodes' edge lists come from the
    // file; iteration order becomes the file's order
    bool insertNode(size_t identifier, image &&payload, std::vector<size_t> &&edge_ids)
    {
        if (!_node_index.emplace(identifier, _nodes.size()).second)
            return false;
        _nodes.push_back(Node{identifier, std::move(payload), std::move(edge_ids)});
        return true;
    }
    bool insertEdge(size_t identifier, camera_relations &&payload, size_t source, size_t dest)
    {
        if (!_edge_index.emplace(identifier, _edges.size()).second)
            return false;
        _edges.push_back(Edge{identifier, source, dest, std::move(payload)});
        return true;
    }
    void clear()
    {
        _nodes.clear();
        _edges.clear();
        _node_index.clear();
        _edge_index.clear();
    }
    const Node *getNode(size_t id) const
    {
        auto it = _node_index.find(id);
        return it == _node_index.end() ? nullptr : &_nodes[it->second];
    }
    Node *getNode(size_t id)
    {
        auto it = _node_index.find(id);
        return it == _node_index.end() ? nullptr : &_nodes[it->second];
    }
    const Edge *getEdge(size_t id) const
    {
        auto it = _edge_index.find(id);
        return it == _edge_index.end() ? nullptr : &_edges[it->second];
    }
    size_t size_nodes() const
    {
        return _nodes.size();
    }
    size_t size_edges() const
    {
        return _edges.size();
    }
    const std::vector<Node> &nodes() const
    {
        return _nodes;
    }
    std::vector<Node> &nodes()
    {
        return _nodes;
    }
    const std::vector<Edge> &edges() const
    {
        return _edges;
    }
    std::vector<Edge> &edges()
    {
        return _edges;
    }
    size_t nodeIndex(size_t id) const
    {
        return _node_index.at(id);
    }

  private:
    std::vector<Node> _nodes;
    std::vector<Edge> _edges;
    std::unordered_map<size_t, size_t> _node_index, _edge_index;
    std::default_random_engine _generator;
    std::uniform_int_distribution<size_t> _distribution;
};

} // namespace opencalibration_amd
