// Flat C driver API for the relax step (declared in include/oc_host.h).
#include "../../../include/oc_host.h"

#include "capi_graph.hpp"
#include "relax.hpp"
#include "relax_mesh.hpp"
#include "invert_distortion.hpp"
#include "relax_stage.hpp"

#include <thread>

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

int och_debug_relax_setup_check(int on)
{
    return opencalibration_amd::relax_setup_check(on);
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


// ---- surfaces and the general relax entry points ------------------------------------------------------------------
} // extern "C"

static void fill_summary12(const RelaxTimers &t, const RelaxMeshStats &st, double *summary)
{
    if (!summary)
        return;
    fill_summary(t, summary);
    summary[8] = st.track_blocks;
    summary[9] = st.two_ray_blocks;
    summary[10] = st.mesh_vertices;
    summary[11] = st.unknowns;
}

extern "C"
{

och_surface *och_surface_create(void)
{
    return new och_surface();
}
void och_surface_destroy(och_surface *s)
{
    delete s;
}
void och_surface_counts(const och_surface *s, size_t *n_vertices, size_t *n_edges, size_t *n_cloud)
{
    if (n_vertices)
        *n_vertices = s->s.mesh.size_nodes();
    if (n_edges)
        *n_edges = s->s.mesh.size_edges();
    if (n_cloud)
    {
        *n_cloud = 0;
        for (const auto &c : s->s.cloud)
            *n_cloud += c.size();
    }
}
void och_surface_get(const och_surface *s, double *vertices, uint64_t *edges5, double *cloud)
{
    const MeshGraph &m = s->s.mesh;
    if (vertices)
        for (size_t i = 0; i < m.nodes.size(); i++)
            std::memcpy(vertices + 3 * i, m.nodes[i].location, 24);
    if (edges5)
        for (size_t i = 0; i < m.edges.size(); i++)
        {
            const MeshEdge &e = m.edges[i];
            edges5[5 * i] = e.source, edges5[5 * i + 1] = e.dest, edges5[5 * i + 2] = e.border;
            edges5[5 * i + 3] = e.triangleOppositeNodes[0], edges5[5 * i + 4] = e.triangleOppositeNodes[1];
        }
    if (cloud)
    {
        size_t k = 0;
        for (const auto &c : s->s.cloud)
            for (const auto &p : c)
            {
                std::memcpy(cloud + 3 * k, p.data(), 24);
                k++;
            }
    }
}
void och_surface_set(och_surface *s, size_t n_vertices, const double *vertices, size_t n_edges, const uint64_t *edges5,
                     size_t n_cloud, const double *cloud)
{
    s->s = surface_model();
    for (size_t i = 0; i < n_vertices; i++)
        s->s.mesh.addNode(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
    for (size_t i = 0; i < n_edges; i++)
    {
        MeshEdge e;
        e.border = edges5[5 * i + 2] != 0;
        e.triangleOppositeNodes[0] = (size_t)edges5[5 * i + 3];
        e.triangleOppositeNodes[1] = (size_t)edges5[5 * i + 4];
        s->s.mesh.addEdge(e, (size_t)edges5[5 * i], (size_t)edges5[5 * i + 1]);
    }
    if (n_cloud)
    {
        point_cloud c(n_cloud);
        for (size_t i = 0; i < n_cloud; i++)
            c[i] = {cloud[3 * i], cloud[3 * i + 1], cloud[3 * i + 2]};
        s->s.cloud.push_back(std::move(c));
    }
}
void och_surface_set_heights(och_surface *s, const double *z)
{
    for (size_t i = 0; i < s->s.mesh.nodes.size(); i++)
        s->s.mesh.nodes[i].location[2] = z[i];
}
void och_rebuild_mesh(const double *cam_xyz, size_t n, const och_surface *previous, int minimal, och_surface *out)
{
    point_cloud cams(n);
    for (size_t i = 0; i < n; i++)
        cams[i] = {cam_xyz[3 * i], cam_xyz[3 * i + 1], cam_xyz[3 * i + 2]};
    std::vector<surface_model> prev;
    if (previous)
        prev.push_back(previous->s);
    out->s = surface_model();
    out->s.mesh = minimal ? buildMinimalMesh(cams, prev) : rebuildMesh(cams, prev);
}

int och_relax(ochip_ctx *ctx, size_t n_nodes, const double *node_pos, const double *node_ori, const double *model10,
              const uint64_t *feat_off, const double *feat_xy, size_t n_poses, const uint64_t *pose_node, double *pose_ori,
              size_t n_edges, const uint64_t *edge_src, const uint64_t *edge_dst, const double *edge_H,
              const uint8_t *edge_is_homography, const uint64_t *inl_off, const double *inl_px, const uint64_t *inl_feat,
              const uint64_t *inl_match_index, const uint64_t *dist_off, const double *dist, size_t n_opt_edges,
              const uint64_t *opt_edges, uint32_t options, double grid_fraction, const och_surface *previous,
              och_surface *surface_out, double *summary_out, double *model10_inout)
{
    return och_relax_ex(ctx, n_nodes, node_pos, node_ori, model10, feat_off, feat_xy, n_poses, pose_node, pose_ori, n_edges, edge_src,
                        edge_dst, edge_H, edge_is_homography, inl_off, inl_px, inl_feat, inl_match_index, dist_off, dist, n_opt_edges,
                        opt_edges, options, grid_fraction, previous, surface_out, summary_out, model10_inout, nullptr, -1, nullptr,
                        nullptr, 0, nullptr);
}

int och_relax_ex(ochip_ctx *ctx, size_t n_nodes, const double *node_pos, const double *node_ori, const double *model10,
                 const uint64_t *feat_off, const double *feat_xy, size_t n_poses, const uint64_t *pose_node, double *pose_ori,
                 size_t n_edges, const uint64_t *edge_src, const uint64_t *edge_dst, const double *edge_H,
                 const uint8_t *edge_is_homography, const uint64_t *inl_off, const double *inl_px, const uint64_t *inl_feat,
                 const uint64_t *inl_match_index, const uint64_t *dist_off, const double *dist, size_t n_opt_edges,
                 const uint64_t *opt_edges, uint32_t options, double grid_fraction, const och_surface *previous,
                 och_surface *surface_out, double *summary_out, double *model10_inout, const double *edge_poses32, int points_mode,
                 double *points_before, double *points_after, size_t points_cap, size_t *n_points_out)
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
        if (feat_off)
            for (uint64_t k = feat_off[i]; k < feat_off[i + 1]; k++)
            {
                feature_2d f;
                f.location[0] = feat_xy[2 * k], f.location[1] = feat_xy[2 * k + 1];
                img.features.push_back(f);
            }
        node_ids[i] = graph.addNode(std::move(img));
    }
    for (size_t e = 0; e < n_edges; e++)
    {
        camera_relations rel;
        if (edge_H)
            std::memcpy(rel.ransac_relation, edge_H + 9 * e, 72);
        rel.relationType = (edge_is_homography && edge_is_homography[e]) ? camera_relations::RelationType::HOMOGRAPHY
                                                                         : camera_relations::RelationType::UNKNOWN;
        for (uint64_t k = inl_off[e]; k < inl_off[e + 1]; k++)
        {
            feature_match_denormalized f;
            f.pixel_1[0] = inl_px[4 * k], f.pixel_1[1] = inl_px[4 * k + 1];
            f.pixel_2[0] = inl_px[4 * k + 2], f.pixel_2[1] = inl_px[4 * k + 3];
            if (inl_feat)
                f.feature_index_1 = inl_feat[2 * k], f.feature_index_2 = inl_feat[2 * k + 1];
            f.match_index = inl_match_index[k];
            rel.inlier_matches.push_back(f);
        }
        if (dist_off)
            for (uint64_t k = dist_off[e]; k < dist_off[e + 1]; k++)
                rel.matches.push_back(feature_match{0, 0, dist[k]});
        if (edge_poses32)
            for (int i = 0; i < 4; i++)
            {
                std::memcpy(rel.relative_poses[i].orientation, edge_poses32 + 32 * e + 8 * i, 32);
                std::memcpy(rel.relative_poses[i].position, edge_poses32 + 32 * e + 8 * i + 4, 24);
                rel.relative_poses[i].score = (int)edge_poses32[32 * e + 8 * i + 7];
            }
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
    std::vector<std::pair<size_t, CameraModel>> cam_models{{model->id, *model}};
    if (model10_inout) // the caller's cam_models entry (it may differ from the graph's model after an earlier relax)
    {
        CameraModel &cm = cam_models[0].second;
        cm.focal_length_pixels = model10_inout[0];
        cm.principle_point[0] = model10_inout[1], cm.principle_point[1] = model10_inout[2];
        for (int i = 0; i < 3; i++)
            cm.radial_distortion[i] = model10_inout[3 + i];
        cm.tangential_distortion[0] = model10_inout[6], cm.tangential_distortion[1] = model10_inout[7];
    }
    RelaxConfig cfg;
    cfg.options = options;
    cfg.ground_mesh_grid_fraction = grid_fraction;
    std::vector<surface_model> prev;
    if (previous)
        prev.push_back(previous->s);
    RelaxTimers t;
    RelaxMeshStats st;
    surface_model out;
    if (points_mode >= 0)
    {
        // TestRelaxProblem of test/test_relax.cpp:470-483: the 3-D point problem set up, then solved / structure-only / left
        std::vector<double> before, after;
        if (!relax_points(ctx, graph, poses, cam_models, opt, options, &out, &t, &g_relax_error, points_mode, &before, &after))
            return -1;
        if (n_points_out)
            *n_points_out = before.size() / 3;
        for (size_t i = 0; i < before.size() && i < 3 * points_cap; i++)
        {
            if (points_before)
                points_before[i] = before[i];
            if (points_after)
                points_after[i] = i < after.size() ? after[i] : NAN;
        }
    }
    else if (!relax(ctx, graph, poses, cam_models, opt, cfg, prev, &out, &t, &st, &g_relax_error))
        return -1;
    for (size_t i = 0; i < n_poses; i++)
        std::memcpy(pose_ori + 4 * i, poses[i].orientation, 32);
    if (surface_out)
        surface_out->s = std::move(out);
    fill_summary12(t, st, summary_out);
    if (model10_inout)
    {
        const CameraModel &cm = cam_models[0].second;
        model10_inout[0] = cm.focal_length_pixels;
        model10_inout[1] = cm.principle_point[0], model10_inout[2] = cm.principle_point[1];
        for (int i = 0; i < 3; i++)
            model10_inout[3 + i] = cm.radial_distortion[i];
        model10_inout[6] = cm.tangential_distortion[0], model10_inout[7] = cm.tangential_distortion[1];
    }
    return 0;
}

static int graph_relax(och_graph *g, ochip_ctx *ctx, double *ori_inout, uint32_t options, double grid_fraction,
                       const och_surface *previous, och_surface *surface_out, double *summary_out, const RelaxShard *shard)
{
    auto &nodes = g->graph.nodes();
    std::vector<NodePose> poses(nodes.size());
    std::vector<std::pair<size_t, CameraModel>> cam_models;
    for (size_t i = 0; i < nodes.size(); i++)
    {
        poses[i].node_id = nodes[i].id;
        std::memcpy(poses[i].orientation, ori_inout + 4 * i, 32);
        std::memcpy(poses[i].position, nodes[i].payload.position, 24);
        bool have = false;
        for (const auto &m : cam_models)
            have |= m.first == nodes[i].payload.model->id;
        if (!have)
            cam_models.emplace_back(nodes[i].payload.model->id, *nodes[i].payload.model);
    }
    std::vector<size_t> opt;
    opt.reserve(g->graph.size_edges());
    for (const auto &e : g->graph.edges())
        opt.push_back(e.id);
    RelaxConfig cfg;
    cfg.options = options;
    cfg.ground_mesh_grid_fraction = grid_fraction;
    std::vector<surface_model> prev;
    if (previous)
        prev.push_back(previous->s);
    RelaxTimers t;
    RelaxMeshStats st;
    surface_model out;
    if (!relax(ctx, g->graph, poses, cam_models, opt, cfg, prev, &out, &t, &st, &g->error, shard))
        return -1;
    for (size_t i = 0; i < nodes.size(); i++)
    {
        std::memcpy(ori_inout + 4 * i, poses[i].orientation, 32);
        std::memcpy(nodes[i].payload.orientation, poses[i].orientation, 32);
    }
    if (surface_out)
        surface_out->s = std::move(out);
    fill_summary12(t, st, summary_out);
    return 0;
}

int och_graph_relax(och_graph *g, ochip_ctx *ctx, double *ori_inout, uint32_t options, double grid_fraction,
                    const och_surface *previous, och_surface *surface_out, double *summary_out)
{
    return graph_relax(g, ctx, ori_inout, options, grid_fraction, previous, surface_out, summary_out, nullptr);
}

int och_graph_relax_sharded(och_graph *g, ochip_ctx *ctx, double *ori_inout, uint32_t options, double grid_fraction,
                            const och_surface *previous, och_surface *surface_out, double *summary_out, uint32_t rank,
                            uint32_t world, ochip_relax_exchange_fn exchange, void *user)
{
    RelaxShard shard;
    shard.rank = rank;
    shard.world = world;
    shard.exchange = exchange;
    shard.user = user;
    return graph_relax(g, ctx, ori_inout, options, grid_fraction, previous, surface_out, summary_out, &shard);
}


struct och_relax_stage
{
    och_graph *g = nullptr;
    RelaxStage stage;
    size_t n_groups = 0;
    std::vector<uint8_t> buf;
};

och_relax_stage *och_relax_stage_begin(och_graph *g, const uint64_t *node_ids, size_t n_ids, int relax_all, int disable_parallelism,
                                       uint32_t options, double grid_fraction, size_t max_groups, const och_surface *previous,
                                       int64_t *group_of_node)
{
    auto *st = new (std::nothrow) och_relax_stage();
    if (!st)
        return nullptr;
    st->g = g;
    RelaxStage &stage = st->stage;
    RelaxConfig cfg;
    cfg.options = options;
    cfg.ground_mesh_grid_fraction = grid_fraction;
    std::vector<size_t> ids(node_ids ? node_ids : nullptr, node_ids ? node_ids + n_ids : nullptr);
    stage.init(g->graph, ids, relax_all != 0, disable_parallelism != 0, cfg);
    if (max_groups > 0)
        stage.trim_groups(max_groups);
    if (group_of_node)
    {
        for (size_t i = 0; i < g->graph.size_nodes(); i++)
            group_of_node[i] = -1;
        const auto &part = stage.partition();
        for (size_t k = 0; k < part.size() && k < stage.num_groups(); k++)
            for (size_t id : part[k])
                group_of_node[g->graph.nodeIndex(id)] = (int64_t)k;
    }
    if (previous)
        stage.setSurfaceModels({previous->s});
    st->n_groups = stage.num_groups();
    return st;
}

size_t och_relax_stage_num_groups(const och_relax_stage *st)
{
    return st->n_groups;
}

int och_relax_stage_run_groups(och_relax_stage *st, ochip_ctx *ctx, uint32_t rank, uint32_t world)
{
    auto runners = st->stage.get_runners(ctx, st->g->graph, rank, std::max<uint32_t>(world, 1));
    run_parallel(runners, st->stage.runner_contexts()); // (a thread per device context: OCHIP_RELAX_RUNNERS)
    return 0;
}

int och_relax_stage_export(och_relax_stage *st, uint32_t rank, uint32_t world, const void **buf, uint64_t *bytes)
{
    st->buf.clear();
    st->stage.export_results(rank, std::max<uint32_t>(world, 1), st->buf);
    *buf = st->buf.data();
    *bytes = st->buf.size();
    return 0;
}

int och_relax_stage_import(och_relax_stage *st, const void *buf, uint64_t bytes)
{
    if (!st->stage.import_results((const uint8_t *)buf, bytes))
    {
        st->g->error = st->stage.error();
        return -1;
    }
    return 0;
}

int och_relax_stage_end(och_relax_stage *st, och_surface *surface_out, double *summary_out)
{
    och_graph *g = st->g;
    RelaxStage &stage = st->stage;
    stage.finalize(g->graph);
    int rc = 0;
    if (!stage.error().empty())
    {
        g->error = stage.error();
        rc = -1;
    }
    else
    {
        if (surface_out)
        {
            const auto &s = stage.getSurfaceModels();
            surface_out->s = s.empty() ? surface_model() : s[0];
        }
        if (summary_out)
        {
            fill_summary12(stage.timers, stage.stats, summary_out);
            summary_out[12] = (double)st->n_groups;
        }
    }
    delete st;
    return rc;
}

int och_relax_stage_run(och_graph *g, ochip_ctx *ctx, const uint64_t *node_ids, size_t n_ids, int relax_all,
                        int disable_parallelism, uint32_t options, double grid_fraction, size_t max_groups,
                        const och_surface *previous, och_surface *surface_out, int64_t *group_of_node, double *summary_out)
{
    och_relax_stage *st = och_relax_stage_begin(g, node_ids, n_ids, relax_all, disable_parallelism, options, grid_fraction,
                                                max_groups, previous, group_of_node);
    if (!st)
        return -1;
    och_relax_stage_run_groups(st, ctx, 0, 1);
    return och_relax_stage_end(st, surface_out, summary_out);
}

size_t och_relax_partition(const och_graph *g, size_t num_groups, int64_t *group_of_node, int64_t *position_in_group)
{
    std::vector<size_t> ids;
    for (const auto &n : g->graph.nodes())
        ids.push_back(n.id);
    const auto groups = relax_partition(g->graph, ids, num_groups);
    for (size_t i = 0; i < g->graph.size_nodes(); i++)
        group_of_node[i] = -1;
    for (size_t k = 0; k < groups.size(); k++)
        for (size_t j = 0; j < groups[k].size(); j++)
        {
            group_of_node[g->graph.nodeIndex(groups[k][j])] = (int64_t)k;
            if (position_in_group)
                position_in_group[g->graph.nodeIndex(groups[k][j])] = (int64_t)j;
        }
    return groups.size();
}

void och_merge_surfaces(const och_surface *const *surfaces, size_t n, och_surface *out)
{
    std::vector<surface_model> v;
    for (size_t i = 0; i < n; i++)
        v.push_back(surfaces[i]->s);
    out->s = mergeSurfaceModels(v);
}


// convertModel (src/distort/invert_distortion.cpp:105-191): forward -> inverse lens model fit, or back
void och_convert_model(const double *m10, int to_inverse, double *out10)
{
    CameraModel m;
    m.focal_length_pixels = m10[0];
    m.principle_point[0] = m10[1], m.principle_point[1] = m10[2];
    for (int i = 0; i < 3; i++)
        m.radial_distortion[i] = m10[3 + i];
    m.tangential_distortion[0] = m10[6], m.tangential_distortion[1] = m10[7];
    m.pixels_cols = (size_t)m10[8], m.pixels_rows = (size_t)m10[9];
    CameraModel r;
    if (to_inverse)
        r = convertModel(m);
    else
    {
        InverseCameraModel inv;
        static_cast<CameraModel &>(inv) = m;
        r = convertModel(inv, 0);
    }
    out10[0] = r.focal_length_pixels;
    out10[1] = r.principle_point[0], out10[2] = r.principle_point[1];
    for (int i = 0; i < 3; i++)
        out10[3 + i] = r.radial_distortion[i];
    out10[6] = r.tangential_distortion[0], out10[7] = r.tangential_distortion[1];
    out10[8] = (double)r.pixels_cols, out10[9] = (double)r.pixels_rows;
}

} // extern "C"
