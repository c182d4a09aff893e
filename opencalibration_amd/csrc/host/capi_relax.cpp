// Flat C driver API for the relax step (declared in include/oc_host.h).
#include "../../../include/oc_host.h"

#include "capi_graph.hpp"
#include "relax.hpp"

#include <cstring>

using namespace opencalibration_amd;

static void fill_summary(const RelaxTimers &t, double *summary)
{
    if (!summary)
        return;
    summary[0] = t.solves;
    summary[1] = t.iterations_total;
    summary[2] = t.last_iterations;
    summary[3] = t.last_initial_cost;
    summary[4] = t.last_final_cost;
    summary[5] = t.last_residual_blocks;
    summary[6] = t.setup_host;
    summary[7] = t.device;
}

extern "C"
{

// Stand-alone problem from flat arrays (same argument meaning as the oracle's oc_relax_ground_plane so the
// parity tests feed both the same data).  Returns 0 or -1 (och_relax_last_error has the text).
static std::string g_relax_error;
const char *och_relax_last_error(void)
{
    return g_relax_error.c_str();
}

int och_relax_ground_plane(ochip_ctx *ctx, size_t n_nodes, const double *node_pos, const double *node_ori,
                           const double *model10, size_t n_poses, const uint64_t *pose_node, double *pose_ori,
                           size_t n_edges, const uint64_t *edge_src, const uint64_t *edge_dst, const double *edge_H,
                           const uint8_t *edge_is_homography, const uint64_t *inl_off, const double *inl_px,
                           const uint64_t *inl_match_index, const uint64_t *dist_off, const double *dist,
                           size_t n_opt_edges, const uint64_t *opt_edges, double *plane_out, double *summary_out)
{
    MeasurementGraph graph;
    auto model = std::make_shared<CameraModel>();
    model->focal_length_pixels = model10[0];
    model->principle_point[0] = model10[1];
    model->principle_point[1] = model10[2];
    for (int i = 0; i < 3; i++)
        model->radial_distortion[i] = model10[3 + i];
    model->tangential_distortion[0] = model10[6];
    model->tangential_distortion[1] = model10[7];
    model->pixels_cols = (size_t)model10[8];
    model->pixels_rows = (size_t)model10[9];
    model->id = 42;
    std::vector<size_t> node_ids(n_nodes), edge_ids(n_edges);
    for (size_t i = 0; i < n_nodes; i++)
    {
        image img;
        img.model = model;
        std::memcpy(img.position, node_pos + 3 * i, 24);
        std::memcpy(img.orientation, node_ori + 4 * i, 32);
        node_ids[i] = graph.addNode(std::move(img));
    }
    for (size_t e = 0; e < n_edges; e++)
    {
        camera_relations rel;
        std::memcpy(rel.ransac_relation, edge_H + 9 * e, 72);
        rel.relationType = (edge_is_homography && edge_is_homography[e]) ? camera_relations::RelationType::HOMOGRAPHY
                                                                         : camera_relations::RelationType::UNKNOWN;
        for (uint64_t k = inl_off[e]; k < inl_off[e + 1]; k++)
        {
            feature_match_denormalized f;
            f.pixel_1[0] = inl_px[4 * k], f.pixel_1[1] = inl_px[4 * k + 1];
            f.pixel_2[0] = inl_px[4 * k + 2], f.pixel_2[1] = inl_px[4 * k + 3];
            f.match_index = inl_match_index[k];
            rel.inlier_matches.push_back(f);
        }
        if (dist_off)
            for (uint64_t k = dist_off[e]; k < dist_off[e + 1]; k++)
                rel.matches.push_back(feature_match{0, 0, dist[k]});
        edge_ids[e] = graph.addEdge(std::move(rel), node_ids[edge_src[e]], node_ids[edge_dst[e]]);
    }
    std::vector<NodePose> poses(n_poses);
    for (size_t i = 0; i < n_poses; i++)
    {
        poses[i].node_id = node_ids[pose_node[i]];
        std::memcpy(poses[i].orientation, pose_ori + 4 * i, 32);
        std::memcpy(poses[i].position, node_pos + 3 * pose_node[i], 24);
    }
    std::vector<size_t> opt(n_opt_edges);
    for (size_t i = 0; i < n_opt_edges; i++)
        opt[i] = edge_ids[opt_edges[i]];
    surface_model_plane surf;
    RelaxTimers t;
    if (!relax_ground_plane(ctx, graph, poses, opt, &surf, &t, &g_relax_error))
        return -1;
    for (size_t i = 0; i < n_poses; i++)
        std::memcpy(pose_ori + 4 * i, poses[i].orientation, 32);
    if (plane_out)
        std::memcpy(plane_out, surf.corner, 72);
    fill_summary(t, summary_out);
    return 0;
}

// Relax every node of a linked graph as ONE group (the single-group global solve of
// pipeline.cpp:653-655): poses = all nodes with the given initial orientations, whitelist = all edges.
// ori_inout: n_nodes x 4 in node order.  The graph's node orientations are updated on success
// (RelaxGroup::finalize, relax_group.cpp:125-135).
static int graph_relax_ground_plane(och_graph *g, ochip_ctx *ctx, double *ori_inout, double *plane_out, double *summary_out,
                                    const RelaxShard *shard)
{
    auto &nodes = g->graph.nodes();
    std::vector<NodePose> poses(nodes.size());
    for (size_t i = 0; i < nodes.size(); i++)
    {
        poses[i].node_id = nodes[i].id;
        std::memcpy(poses[i].orientation, ori_inout + 4 * i, 32);
        std::memcpy(poses[i].position, nodes[i].payload.position, 24);
    }
    std::vector<size_t> opt;
    opt.reserve(g->graph.size_edges());
    for (const auto &e : g->graph.edges())
        opt.push_back(e.id);
    surface_model_plane surf;
    RelaxTimers t;
    if (!relax_ground_plane(ctx, g->graph, poses, opt, &surf, &t, &g->error, shard))
        return -1;
    for (size_t i = 0; i < nodes.size(); i++)
    {
        std::memcpy(ori_inout + 4 * i, poses[i].orientation, 32);
        std::memcpy(nodes[i].payload.orientation, poses[i].orientation, 32);
    }
    if (plane_out)
        std::memcpy(plane_out, surf.corner, 72);
    fill_summary(t, summary_out);
    return 0;
}

int och_graph_relax_ground_plane(och_graph *g, ochip_ctx *ctx, double *ori_inout, double *plane_out,
                                 double *summary_out)
{
    return graph_relax_ground_plane(g, ctx, ori_inout, plane_out, summary_out, nullptr);
}

int och_graph_relax_ground_plane_sharded(och_graph *g, ochip_ctx *ctx, double *ori_inout, double *plane_out,
                                         double *summary_out, uint32_t rank, uint32_t world,
                                         ochip_relax_exchange_fn exchange, void *user)
{
    RelaxShard shard;
    shard.rank = rank;
    shard.world = world;
    shard.exchange = exchange;
    shard.user = user;
    return graph_relax_ground_plane(g, ctx, ori_inout, plane_out, summary_out, &shard);
}

} // extern "C"
