// ORACLE — test infrastructure only.  Flat C driver over oracle/relax_full.{hpp,cpp} (handle based, for ctypes).
#include "relax_full.hpp"

#include <cstring>

using namespace oracle;
using namespace oracle::rx;

namespace
{
struct graph_handle
{
    MeasurementGraph graph;
    std::vector<std::shared_ptr<CameraModel>> models;
    // the caller's cam_models map kept across relax() calls (test/test_relax.cpp:436-463 relaxes ten times with the same
    // map, which every call updates while the graph's own models stay as they are)
    bool persist_cam_models = false;
    model_map cam_models;
};
CameraModel model_from10(const double *m, size_t id)
{
    CameraModel cm;
    cm.focal_length_pixels = m[0];
    cm.principle_point[0] = m[1];
    cm.principle_point[1] = m[2];
    for (int i = 0; i < 3; i++)
        cm.radial_distortion[i] = m[3 + i];
    cm.tangential_distortion[0] = m[6];
    cm.tangential_distortion[1] = m[7];
    cm.pixels_cols = (size_t)m[8];
    cm.pixels_rows = (size_t)m[9];
    cm.id = id;
    return cm;
}
void model_to10(const camera_model &cm, double *m)
{
    m[0] = cm.focal_length_pixels;
    m[1] = cm.principle_point[0];
    m[2] = cm.principle_point[1];
    for (int i = 0; i < 3; i++)
        m[3 + i] = cm.radial_distortion[i];
    m[6] = cm.tangential_distortion[0];
    m[7] = cm.tangential_distortion[1];
    m[8] = (double)cm.pixels_cols;
    m[9] = (double)cm.pixels_rows;
}
} // namespace

extern "C"
{

void *ocx_graph_create()
{
    return new graph_handle();
}
void ocx_graph_destroy(void *h)
{
    delete (graph_handle *)h;
}
// model10: focal, pp x y, k1 k2 k3, p1 p2, cols, rows.  Returns the model index; its CameraModel::id is `id`.
size_t ocx_graph_add_model(void *h, const double *model10, size_t id)
{
    auto *g = (graph_handle *)h;
    g->models.push_back(std::make_shared<CameraModel>(model_from10(model10, id)));
    return g->models.size() - 1;
}
void ocx_graph_get_model(void *h, size_t model_index, double *model10)
{
    model_to10(*((graph_handle *)h)->models[model_index], model10);
}
void ocx_graph_set_model(void *h, size_t model_index, const double *model10)
{
    auto *g = (graph_handle *)h;
    const size_t id = g->models[model_index]->id;
    *g->models[model_index] = model_from10(model10, id);
}
size_t ocx_graph_add_node(void *h, const char *path, const double *pos3, const double *ori4, size_t model_index,
                          size_t n_features, const double *feature_xy)
{
    auto *g = (graph_handle *)h;
    image_node n;
    n.path = path ? path : "";
    n.position = Vec3{pos3[0], pos3[1], pos3[2]};
    n.orientation = Quat{ori4[0], ori4[1], ori4[2], ori4[3]};
    n.model = g->models[model_index];
    n.feature_location.resize(n_features);
    for (size_t i = 0; i < n_features; i++)
        n.feature_location[i] = Vec2{feature_xy[2 * i], feature_xy[2 * i + 1]};
    return g->graph.addNode(std::move(n));
}
// inl_px: n_inl x 4 (pixel_1 xy, pixel_2 xy); inl_idx: n_inl x 3 (feature_index_1, feature_index_2, match_index);
// match_dist: relations.matches[i].distance (n_matches, may be 0); poses: 4 x {q xyzw, t xyz, score} or null
size_t ocx_graph_add_edge(void *h, size_t source, size_t dest, const double *H9, int is_homography, size_t n_inl,
                          const double *inl_px, const uint64_t *inl_idx, size_t n_matches, const double *match_dist,
                          const double *poses)
{
    auto *g = (graph_handle *)h;
    relation r;
    if (H9)
        std::memcpy(r.ransac_relation.m, H9, 72);
    else
        r.ransac_relation = identity3();
    r.is_homography = is_homography != 0;
    r.inlier_matches.resize(n_inl);
    for (size_t k = 0; k < n_inl; k++)
    {
        auto &f = r.inlier_matches[k];
        f.pixel_1[0] = inl_px[4 * k], f.pixel_1[1] = inl_px[4 * k + 1];
        f.pixel_2[0] = inl_px[4 * k + 2], f.pixel_2[1] = inl_px[4 * k + 3];
        f.feature_index_1 = inl_idx[3 * k], f.feature_index_2 = inl_idx[3 * k + 1], f.match_index = inl_idx[3 * k + 2];
    }
    r.matches.resize(n_matches);
    for (size_t k = 0; k < n_matches; k++)
    {
        r.matches[k].feature_index_1 = r.matches[k].feature_index_2 = 0;
        r.matches[k].distance = match_dist[k];
    }
    if (poses)
        for (int i = 0; i < 4; i++)
        {
            const double *p = poses + 8 * i;
            r.relative_poses[i].orientation = Quat{p[0], p[1], p[2], p[3]};
            r.relative_poses[i].position = Vec3{p[4], p[5], p[6]};
            r.relative_poses[i].score = (int)p[7];
        }
    return g->graph.addEdge(std::move(r), source, dest);
}
void ocx_graph_set_orientation(void *h, size_t node, const double *ori4)
{
    ((graph_handle *)h)->graph.nodes[node].orientation = Quat{ori4[0], ori4[1], ori4[2], ori4[3]};
}
void ocx_graph_get_orientations(void *h, double *ori4)
{
    auto &nodes = ((graph_handle *)h)->graph.nodes;
    for (size_t i = 0; i < nodes.size(); i++)
    {
        ori4[4 * i] = nodes[i].orientation.x, ori4[4 * i + 1] = nodes[i].orientation.y;
        ori4[4 * i + 2] = nodes[i].orientation.z, ori4[4 * i + 3] = nodes[i].orientation.w;
    }
}

// ---- surfaces
void *ocx_surface_create()
{
    return new surface_model();
}
void ocx_surface_destroy(void *s)
{
    delete (surface_model *)s;
}
size_t ocx_surface_num_vertices(void *s)
{
    return ((surface_model *)s)->mesh.size_nodes();
}
size_t ocx_surface_num_edges(void *s)
{
    return ((surface_model *)s)->mesh.size_edges();
}
size_t ocx_surface_num_cloud_points(void *s)
{
    size_t n = 0;
    for (auto &c : ((surface_model *)s)->cloud)
        n += c.size();
    return n;
}
void ocx_surface_get(void *s, double *vertices_xyz, uint64_t *edges5 /* source dest border opp0 opp1 */, double *cloud_xyz)
{
    auto *sm = (surface_model *)s;
    if (vertices_xyz)
        for (size_t i = 0; i < sm->mesh.nodes.size(); i++)
        {
            vertices_xyz[3 * i] = sm->mesh.nodes[i].location.x;
            vertices_xyz[3 * i + 1] = sm->mesh.nodes[i].location.y;
            vertices_xyz[3 * i + 2] = sm->mesh.nodes[i].location.z;
        }
    if (edges5)
        for (size_t i = 0; i < sm->mesh.edges.size(); i++)
        {
            const auto &e = sm->mesh.edges[i];
            edges5[5 * i] = e.source, edges5[5 * i + 1] = e.dest, edges5[5 * i + 2] = e.border;
            edges5[5 * i + 3] = e.opposite[0], edges5[5 * i + 4] = e.opposite[1];
        }
    if (cloud_xyz)
    {
        size_t k = 0;
        for (auto &c : sm->cloud)
            for (auto &p : c)
            {
                cloud_xyz[3 * k] = p.x, cloud_xyz[3 * k + 1] = p.y, cloud_xyz[3 * k + 2] = p.z;
                k++;
            }
    }
}
// build a surface from arrays (a refined mesh handed back as the previous surface)
void ocx_surface_set(void *s, size_t n_vertices, const double *vertices_xyz, size_t n_edges, const uint64_t *edges5,
                     size_t n_cloud, const double *cloud_xyz)
{
    auto *sm = (surface_model *)s;
    *sm = surface_model();
    for (size_t i = 0; i < n_vertices; i++)
        sm->mesh.addNode(Vec3{vertices_xyz[3 * i], vertices_xyz[3 * i + 1], vertices_xyz[3 * i + 2]});
    for (size_t i = 0; i < n_edges; i++)
    {
        mesh_edge e;
        e.border = edges5[5 * i + 2] != 0;
        e.opposite[0] = edges5[5 * i + 3], e.opposite[1] = edges5[5 * i + 4];
        sm->mesh.addEdge(e, edges5[5 * i], edges5[5 * i + 1]);
    }
    if (n_cloud)
    {
        point_cloud c(n_cloud);
        for (size_t i = 0; i < n_cloud; i++)
            c[i] = Vec3{cloud_xyz[3 * i], cloud_xyz[3 * i + 1], cloud_xyz[3 * i + 2]};
        sm->cloud.push_back(std::move(c));
    }
}
void ocx_rebuild_mesh(const double *cam_xyz, size_t n, void *prev_surface /* may be null */, int minimal, void *out_surface)
{
    point_cloud cams(n);
    for (size_t i = 0; i < n; i++)
        cams[i] = Vec3{cam_xyz[3 * i], cam_xyz[3 * i + 1], cam_xyz[3 * i + 2]};
    std::vector<surface_model> prev;
    if (prev_surface)
        prev.push_back(*(surface_model *)prev_surface);
    auto *o = (surface_model *)out_surface;
    *o = surface_model();
    o->mesh = minimal ? buildMinimalMesh(cams, prev) : rebuildMesh(cams, prev);
}
// vertical-ray triangle lookup on a surface's mesh; returns the IntersectionInfo type, tri3 = node indexes
int ocx_surface_triangle_at(void *s, double x, double y, double z_from, uint64_t *tri3, uint64_t *steps)
{
    MeshIntersectionSearcher searcher;
    if (!searcher.init(((surface_model *)s)->mesh))
        return -1;
    const auto &info = searcher.triangleIntersect(Vec3{0, 0, -1}, Vec3{x, y, z_from});
    for (int i = 0; i < 3; i++)
        tri3[i] = info.nodeIndexes[i];
    if (steps)
        *steps = info.steps;
    return (int)info.type;
}

static void stats_out(const relax_stats &st, double *summary_out, int32_t *iters_out, size_t iters_cap)
{
    if (summary_out)
    {
        summary_out[0] = st.solves;
        summary_out[1] = st.iterations_total;
        summary_out[2] = st.last_iterations;
        summary_out[3] = st.last_initial_cost;
        summary_out[4] = st.last_final_cost;
        summary_out[5] = st.last_residual_blocks;
        summary_out[6] = st.last_parameter_blocks;
        summary_out[7] = st.track_blocks;
        summary_out[8] = st.two_ray_blocks;
    }
    if (iters_out)
        for (size_t i = 0; i < iters_cap; i++)
            iters_out[i] = i < st.iterations_per_solve.size() ? st.iterations_per_solve[i] : -1;
}

// relax(graph, nodes, cam_models, edges_to_optimize, config, previousSurfaces) (relax.hpp:12-15).
//  poses: node ids + orientations (in/out, NaN = uninitialised); positions come from the graph
//  cam_models: the models of the pose nodes, in first-seen order (as RelaxGroup::init collects them); written back to
//  models_out (n_models_out x (id, model10)) when given
int ocx_relax(void *h, size_t n_poses, const uint64_t *pose_node, double *pose_ori, size_t n_opt_edges,
              const uint64_t *opt_edges, uint32_t options, double grid_fraction, void *prev_surface, void *out_surface,
              double *summary_out, int32_t *iters_out, size_t iters_cap, double *models_out, size_t models_cap)
{
    auto *g = (graph_handle *)h;
    std::vector<NodePose> poses(n_poses);
    model_map cam_models;
    for (size_t i = 0; i < n_poses; i++)
    {
        poses[i].node_id = pose_node[i];
        poses[i].orientation = Quat{pose_ori[4 * i], pose_ori[4 * i + 1], pose_ori[4 * i + 2], pose_ori[4 * i + 3]};
        const image_node &n = g->graph.nodes[pose_node[i]];
        poses[i].position = n.position;
        bool found = false;
        for (auto &m : cam_models)
            found |= m.first == n.model->id;
        if (!found)
        {
            const CameraModel *kept = nullptr;
            if (g->persist_cam_models)
                for (auto &m : g->cam_models)
                    if (m.first == n.model->id)
                        kept = &m.second;
            cam_models.emplace_back(n.model->id, kept ? *kept : *n.model);
        }
    }
    std::vector<size_t> opt(opt_edges, opt_edges + n_opt_edges);
    RelaxConfig cfg;
    cfg.options.bits = options;
    cfg.ground_mesh_grid_fraction = grid_fraction;
    std::vector<surface_model> prev;
    if (prev_surface)
        prev.push_back(*(surface_model *)prev_surface);
    relax_stats st;
    surface_model s = relax(g->graph, poses, cam_models, opt, cfg, prev, &st);
    for (size_t i = 0; i < n_poses; i++)
    {
        pose_ori[4 * i] = poses[i].orientation.x, pose_ori[4 * i + 1] = poses[i].orientation.y;
        pose_ori[4 * i + 2] = poses[i].orientation.z, pose_ori[4 * i + 3] = poses[i].orientation.w;
    }
    if (out_surface)
        *(surface_model *)out_surface = std::move(s);
    if (g->persist_cam_models)
        g->cam_models = cam_models;
    stats_out(st, summary_out, iters_out, iters_cap);
    if (models_out)
        for (size_t i = 0; i < cam_models.size() && i < models_cap; i++)
        {
            models_out[11 * i] = (double)cam_models[i].first;
            model_to10(cam_models[i].second, models_out + 11 * i + 1);
        }
    return (int)cam_models.size();
}

// TestRelaxProblem (test/test_relax.cpp:470-483): setup3dPointProblem, then nothing / solve / relaxObservedModelOnly
// (mode 0 / 1 / 2).  before_xyz / after_xyz (cap points each): the tracks' 3-D points after the set-up and at the end.
// model10_inout: the cam_models entry of the (single) camera model, in/out.  Returns the number of track points.
size_t ocx_points_problem(void *h, size_t n_poses, const uint64_t *pose_node, double *pose_ori, size_t n_opt_edges,
                          const uint64_t *opt_edges, uint32_t options, int mode, double *before_xyz, double *after_xyz, size_t cap,
                          double *summary_out, double *model10_inout)
{
    auto *g = (graph_handle *)h;
    std::vector<NodePose> poses(n_poses);
    model_map cam_models;
    for (size_t i = 0; i < n_poses; i++)
    {
        poses[i].node_id = pose_node[i];
        poses[i].orientation = Quat{pose_ori[4 * i], pose_ori[4 * i + 1], pose_ori[4 * i + 2], pose_ori[4 * i + 3]};
        const image_node &n = g->graph.nodes[pose_node[i]];
        poses[i].position = n.position;
        bool found = false;
        for (auto &m : cam_models)
            found |= m.first == n.model->id;
        if (!found)
        {
            cam_models.emplace_back(n.model->id, *n.model);
            if (model10_inout)
            {
                const size_t id = n.model->id;
                cam_models.back().second = model_from10(model10_inout, id);
            }
        }
    }
    std::vector<size_t> opt(opt_edges, opt_edges + n_opt_edges);
    RelaxOptionSet o;
    o.bits = options;
    relax_stats st;
    std::vector<Vec3> before, after;
    points_problem_steps(g->graph, poses, cam_models, opt, o, mode, &before, &after, &st);
    for (size_t i = 0; i < n_poses; i++)
    {
        pose_ori[4 * i] = poses[i].orientation.x, pose_ori[4 * i + 1] = poses[i].orientation.y;
        pose_ori[4 * i + 2] = poses[i].orientation.z, pose_ori[4 * i + 3] = poses[i].orientation.w;
    }
    for (size_t i = 0; i < before.size() && i < cap; i++)
    {
        if (before_xyz)
            before_xyz[3 * i] = before[i].x, before_xyz[3 * i + 1] = before[i].y, before_xyz[3 * i + 2] = before[i].z;
        if (after_xyz)
            after_xyz[3 * i] = after[i].x, after_xyz[3 * i + 1] = after[i].y, after_xyz[3 * i + 2] = after[i].z;
    }
    stats_out(st, summary_out, nullptr, 0);
    if (model10_inout && !cam_models.empty())
        model_to10(cam_models[0].second, model10_inout);
    return before.size();
}

// RelaxGroup::init + run + finalize (relax_group.cpp) on the graph.  knn10: n_nodes x 10 node ids (UINT64_MAX padded) =
// imageGPSLocations.searchKnn(position, 10) per node.  Returns the number of local poses; local_nodes_out (cap entries)
// receives their node ids in the sorted order, opt_edges_out the chosen edges.
size_t ocx_relax_group(void *h, size_t n_ids, const uint64_t *node_ids, const uint64_t *knn10, size_t depth, uint32_t options,
                       double grid_fraction, void *prev_surface, void *out_surface, int run, double *summary_out,
                       uint64_t *local_nodes_out, size_t local_cap, uint64_t *opt_edges_out, size_t edges_cap,
                       size_t *n_edges_out)
{
    auto *g = (graph_handle *)h;
    std::vector<size_t> ids(node_ids, node_ids + n_ids), knn(g->graph.nodes.size() * 10);
    for (size_t i = 0; i < knn.size(); i++)
        knn[i] = knn10[i] == UINT64_MAX ? NONE : (size_t)knn10[i];
    RelaxConfig cfg;
    cfg.options.bits = options;
    cfg.ground_mesh_grid_fraction = grid_fraction;
    RelaxGroup group;
    group.init(g->graph, ids, knn, depth, cfg);
    const size_t n_local = group._local_poses.size();
    for (size_t i = 0; i < n_local && i < local_cap; i++)
        local_nodes_out[i] = group._local_poses[i].node_id;
    for (size_t i = 0; i < group._edges_to_optimize.size() && i < edges_cap; i++)
        opt_edges_out[i] = group._edges_to_optimize[i];
    if (n_edges_out)
        *n_edges_out = group._edges_to_optimize.size();
    if (run)
    {
        std::vector<surface_model> prev;
        if (prev_surface)
            prev.push_back(*(surface_model *)prev_surface);
        relax_stats st;
        surface_model s = group.run(g->graph, prev, &st);
        if (out_surface)
            *(surface_model *)out_surface = std::move(s);
        group.finalize(g->graph);
        stats_out(st, summary_out, nullptr, 0);
    }
    return n_local;
}

// RelaxStage::init's partition: group_of_node[i] = index of the group (largest first) node i is a primary node of, or -1
size_t ocx_relax_stage_groups(void *h, size_t n_ids, const uint64_t *node_ids, int relax_all, int disable_parallelism,
                              uint32_t options, int64_t *group_of_node, size_t *depth_out, int64_t *position_in_group)
{
    auto *g = (graph_handle *)h;
    std::vector<size_t> ids(node_ids, node_ids + n_ids);
    size_t depth = 0;
    const auto groups = relax_stage_groups(g->graph, ids, relax_all != 0, disable_parallelism != 0, options, &depth);
    for (size_t i = 0; i < g->graph.nodes.size(); i++)
        group_of_node[i] = -1;
    for (size_t k = 0; k < groups.size(); k++)
        for (size_t j = 0; j < groups[k].size(); j++)
        {
            group_of_node[groups[k][j]] = (int64_t)k;
            if (position_in_group)
                position_in_group[groups[k][j]] = (int64_t)j;
        }
    if (depth_out)
        *depth_out = depth;
    return groups.size();
}

void ocx_graph_persist_cam_models(void *h, int on)
{
    auto *g = (graph_handle *)h;
    g->persist_cam_models = on != 0;
    if (!on)
        g->cam_models.clear();
}

// forward <-> inverse lens model fits (invert_distortion.cpp:105-191)
void ocx_convert_model(const double *model10, int to_inverse, double *out10)
{
    const CameraModel m = model_from10(model10, 0);
    model_to10(to_inverse ? convertModelToInverse(m) : convertModelToForward(m), out10);
}
void ocx_image_to_3d_inverse(const double *px, size_t n, const double *inverse_model10, double *rays)
{
    const CameraModel m = model_from10(inverse_model10, 0);
    for (size_t i = 0; i < n; i++)
    {
        const Vec3 r = image_to_3d_inverse_model(px + 2 * i, m);
        rays[3 * i] = r.x, rays[3 * i + 1] = r.y, rays[3 * i + 2] = r.z;
    }
}

} // extern "C"

// a ray against a surface's mesh (MeshIntersectionSearcher::triangleIntersect, intersect.cpp:39-163): the IntersectionInfo type,
// the intersection point and the triangle's node indexes - for test/test_meshgraph.cpp restated
extern "C" int ocx_surface_intersect(void *s, const double *dir3, const double *off3, double *loc3, uint64_t *tri3)
{
    MeshIntersectionSearcher searcher;
    if (!searcher.init(((surface_model *)s)->mesh))
        return -1;
    const auto &info = searcher.triangleIntersect(Vec3{dir3[0], dir3[1], dir3[2]}, Vec3{off3[0], off3[1], off3[2]});
    loc3[0] = info.intersectionLocation.x, loc3[1] = info.intersectionLocation.y, loc3[2] = info.intersectionLocation.z;
    for (int i = 0; i < 3; i++)
        tri3[i] = info.nodeIndexes[i];
    return (int)info.type;
}
