// Flat C driver API over the C++ host side (liboc_host.so) so that tests/ and bench.py (Python,
// ctypes) can drive the same code a C++ application would link.  Declared in include/oc_host.h.
#include "../../../include/oc_host.h"

#include "match_features.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>

using namespace opencalibration_amd;

static std::vector<feature_2d> features_from(const double *loc, const float *strength, const uint64_t *desc, size_t n)
{
    std::vector<feature_2d> f(n);
    for (size_t i = 0; i < n; i++)
    {
        if (loc)
        {
            f[i].location[0] = loc[2 * i];
            f[i].location[1] = loc[2 * i + 1];
        }
        if (strength)
            f[i].strength = strength[i];
        if (desc)
            std::memcpy(f[i].descriptor, desc + 8 * i, 64);
    }
    return f;
}

extern "C"
{

size_t och_subsample(const double *loc, const float *strength, size_t n, double spacing, size_t count, uint64_t *out)
{
    auto f = features_from(loc, strength, nullptr, n);
    auto idx = spatially_subsample_feature_indices(f, spacing, count);
    for (size_t i = 0; i < idx.size(); i++)
        out[i] = idx[i];
    return idx.size();
}

size_t och_matches_from_device(const ochip_match *raw, const uint64_t *idx1, size_t n1, const uint64_t *idx2,
                               size_t n2, uint64_t *out_i1, uint64_t *out_i2, double *out_dist)
{
    std::vector<size_t> i1(idx1, idx1 + n1), i2(idx2, idx2 + n2);
    auto m = matches_from_device(raw, i1, i2);
    for (size_t i = 0; i < m.size(); i++)
    {
        out_i1[i] = m[i].feature_index_1;
        out_i2[i] = m[i].feature_index_2;
        out_dist[i] = m[i].distance;
    }
    return m.size();
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// graph + link stage driver
#include "capi_graph.hpp"

extern "C"
{

och_graph *och_graph_create(void)
{
    return new (std::nothrow) och_graph();
}

// homography_model::decompose (homography_model.cpp:138-185) on the inlier rays m1m2 = n x {measurement1, measurement2};
// poses: 4 x {orientation xyzw, position xyz, score}.  Returns can_decompose.  No device involved.
int och_homography_decompose(const double *H9, const double *m1m2, size_t n, double *poses)
{
    homography_model h;
    std::memcpy(h.homography, H9, sizeof h.homography);
    std::array<decomposed_pose, 4> rel;
    const bool ok = h.decompose_inlier_rays(m1m2, n, rel);
    for (int i = 0; i < 4; i++)
    {
        double *o = poses + 8 * i;
        std::memcpy(o, rel[i].orientation, 32);
        std::memcpy(o + 4, rel[i].position, 24);
        o[7] = rel[i].score;
    }
    return ok ? 1 : 0;
}

// image_to_3d (distort_keypoints.cpp:68-103) of n pixels with model10 = {f, ppx, ppy, k1, k2, k3, p1, p2, cols, rows}
void och_image_to_3d(const double *px, size_t n, const double *m10, double *rays)
{
    CameraModel m;
    m.focal_length_pixels = m10[0];
    m.principle_point[0] = m10[1];
    m.principle_point[1] = m10[2];
    for (int i = 0; i < 3; i++)
        m.radial_distortion[i] = m10[3 + i];
    m.tangential_distortion[0] = m10[6];
    m.tangential_distortion[1] = m10[7];
    for (size_t i = 0; i < n; i++)
        image_to_3d(px + 2 * i, m, rays + 3 * i);
}

// ransac<fundamental_matrix_model> (model 0) / ransac<essential_matrix_model> (model 1) on the device for one set of
// correspondences (rays6: n x {measurement1, measurement2}; quality: n or NULL).  The PROSAC order (std::sort of the
// indices by quality, ransac.cpp:83-90) and the shuffled evaluation order (:158) come from libstdc++ here, as for the
// homography model.  M9: the model, inliers: n flags, counts3: {iterations, improvements, inliers}.  Returns the score
// (ransac()'s return value), NAN on a device error.
double och_ransac_epipolar(ochip_ctx *ctx, int model, const double *rays6, const double *quality, size_t n, double threshold,
                           double *M9, uint8_t *inliers, uint32_t *counts3)
{
    static EvalOrderCache cache;
    bool has_quality = false;
    for (size_t i = 0; quality && i < n; i++)
        has_quality = has_quality || quality[i] != 0;
    std::vector<uint32_t> sorted_idx(std::max<size_t>(n, 1), 0);
    if (has_quality)
    {
        std::vector<size_t> idx(n);
        for (size_t i = 0; i < n; i++)
            idx[i] = i;
        std::sort(idx.begin(), idx.end(), [quality](size_t a, size_t b) { return quality[a] < quality[b]; });
        for (size_t i = 0; i < n; i++)
            sorted_idx[i] = (uint32_t)idx[i];
    }
    const eval_order_entry &eo = cache.get(n);
    ochip_epipolar_job job{};
    job.n = (uint32_t)n;
    job.rng_state = eo.rng_state;
    job.has_quality = has_quality ? 1u : 0u;
    ochip_ransac_result res{};
    std::vector<uint8_t> inl(std::max<size_t>(n, 1), 0);
    std::vector<uint32_t> order(eo.order.begin(), eo.order.end());
    if (order.empty())
        order.push_back(0);
    const double none[6] = {0, 0, 1, 0, 0, 1};
    if (ochip_ransac_epipolar_batch(ctx, model, &job, 1, n ? rays6 : none, sorted_idx.data(), n, order.data(), n, threshold, &res,
                                    inl.data()) != OCHIP_OK)
        return NAN;
    if (M9)
        std::memcpy(M9, res.H, 72);
    if (inliers)
        std::memcpy(inliers, inl.data(), n);
    if (counts3)
    {
        counts3[0] = res.iterations;
        counts3[1] = res.improvements;
        counts3[2] = res.n_inliers;
    }
    return res.score;
}

void och_graph_destroy(och_graph *g)
{
    delete g;
}

const char *och_last_error(const och_graph *g)
{
    return g ? g->error.c_str() : "null graph";
}

uint32_t och_graph_add_model(och_graph *g, const double *m10)
{
    auto m = std::make_shared<CameraModel>();
    m->focal_length_pixels = m10[0];
    m->principle_point[0] = m10[1];
    m->principle_point[1] = m10[2];
    for (int i = 0; i < 3; i++)
        m->radial_distortion[i] = m10[3 + i];
    m->tangential_distortion[0] = m10[6];
    m->tangential_distortion[1] = m10[7];
    m->pixels_cols = (size_t)m10[8];
    m->pixels_rows = (size_t)m10[9];
    m->id = g->models.size() + 1;
    g->models.push_back(m);
    return (uint32_t)(g->models.size() - 1);
}

uint64_t och_graph_add_image(och_graph *g, const double *loc, const float *strength, const uint64_t *desc, size_t n,
                             size_t num_sparse, uint32_t model, const double *position3)
{
    image img;
    img.features = features_from(loc, strength, desc, n);
    img.num_sparse_features = num_sparse;
    img.model = g->models.at(model);
    for (int i = 0; i < 3; i++)
        img.position[i] = position3[i];
    img.path = "synthetic_" + std::to_string(g->graph.size_nodes());
    return g->graph.addNode(std::move(img));
}

// graph.addEdge(relations, source, dest) from flat arrays (what a deserialised graph or a test hands over)
uint64_t och_graph_add_edge(och_graph *g, uint64_t source_id, uint64_t dest_id, const double *H9, int is_homography, size_t n_inliers,
                            const double *inl_px4, const uint64_t *inl_idx3, size_t n_matches, const uint64_t *match_idx2,
                            const double *match_dist, const double *poses32)
{
    camera_relations rel;
    if (H9)
        std::memcpy(rel.ransac_relation, H9, 72);
    rel.relationType = is_homography ? camera_relations::RelationType::HOMOGRAPHY : camera_relations::RelationType::UNKNOWN;
    rel.inlier_matches.resize(n_inliers);
    for (size_t k = 0; k < n_inliers; k++)
    {
        feature_match_denormalized &f = rel.inlier_matches[k];
        f.pixel_1[0] = inl_px4[4 * k], f.pixel_1[1] = inl_px4[4 * k + 1];
        f.pixel_2[0] = inl_px4[4 * k + 2], f.pixel_2[1] = inl_px4[4 * k + 3];
        f.feature_index_1 = inl_idx3[3 * k], f.feature_index_2 = inl_idx3[3 * k + 1], f.match_index = inl_idx3[3 * k + 2];
    }
    rel.matches.resize(n_matches);
    for (size_t k = 0; k < n_matches; k++)
        rel.matches[k] = feature_match{match_idx2 ? (size_t)match_idx2[2 * k] : 0, match_idx2 ? (size_t)match_idx2[2 * k + 1] : 0,
                                       match_dist ? match_dist[k] : 0.0};
    if (poses32)
        for (int i = 0; i < 4; i++)
        {
            std::memcpy(rel.relative_poses[i].orientation, poses32 + 8 * i, 32);
            std::memcpy(rel.relative_poses[i].position, poses32 + 8 * i + 4, 24);
            rel.relative_poses[i].score = (int)poses32[8 * i + 7];
        }
    if (!g->graph.getNode(source_id) || !g->graph.getNode(dest_id))
    {
        g->error = "och_graph_add_edge: unknown node id";
        return 0;
    }
    return g->graph.addEdge(std::move(rel), source_id, dest_id);
}

void och_link_match_work(const och_graph *g, double *out3)
{
    out3[0] = out3[1] = out3[2] = 0;
    if (g->link)
        g->link->match_work(out3);
}

void och_graph_get_orientations(const och_graph *g, double *ori)
{
    size_t i = 0;
    for (const auto &n : g->graph.nodes())
        std::memcpy(ori + 4 * (i++), n.payload.orientation, 32);
}

size_t och_graph_num_nodes(const och_graph *g)
{
    return g->graph.size_nodes();
}
size_t och_graph_num_edges(const och_graph *g)
{
    return g->graph.size_edges();
}
void och_graph_node_ids(const och_graph *g, uint64_t *out)
{
    size_t i = 0;
    for (const auto &n : g->graph.nodes())
        out[i++] = n.id;
}

// init + the single runner + finalize over the given nodes.  timers: 8 doubles (LinkTimers order).
int och_link_stage_run(och_graph *g, ochip_ctx *ctx, const uint64_t *node_ids, size_t n, int keep_debug, double *timers)
{
    g->link = std::make_unique<LinkStage>(ctx);
    g->link->keep_debug = keep_debug != 0;
    std::vector<size_t> ids(node_ids, node_ids + n);
    g->link->init(g->graph, ids);
    {
        // the runners are independent batches: run them concurrently, as the reference's pipeline runs its closures
        auto runners = g->link->get_runners(g->graph);
        std::vector<std::thread> threads;
        for (size_t r = 1; r < runners.size(); r++)
            threads.emplace_back(runners[r]);
        if (!runners.empty())
            runners[0]();
        for (auto &t : threads)
            t.join();
    }
    if (!g->link->error.empty())
    {
        g->error = g->link->error;
        return -1;
    }
    g->link->finalize(g->graph);
    if (timers)
    {
        const LinkTimers &t = g->link->timers;
        const double v[8] = {t.link_init,  t.subsample,     t.upload,         t.match_device,
                             t.match_host, t.ransac_device, t.decompose_host, t.link_finalize};
        std::memcpy(timers, v, sizeof v);
    }
    return 0;
}

size_t och_link_debug_count(const och_graph *g)
{
    return g->link ? g->link->debug.size() : 0;
}

void och_link_debug_pair(const och_graph *g, size_t p, uint64_t *ids2, uint64_t *n_matches, double *score,
                         uint32_t *iters3 /*iterations, improvements, can_decompose*/)
{
    const auto &d = g->link->debug.at(p);
    ids2[0] = d.node_id;
    ids2[1] = d.match_node_id;
    *n_matches = d.matches.size();
    *score = d.score;
    iters3[0] = d.iterations;
    iters3[1] = d.improvements;
    iters3[2] = d.can_decompose ? 1 : 0;
}

void och_link_debug_matches(const och_graph *g, size_t p, uint64_t *i1, uint64_t *i2, double *dist, uint8_t *inl)
{
    const auto &d = g->link->debug.at(p);
    for (size_t i = 0; i < d.matches.size(); i++)
    {
        i1[i] = d.matches[i].feature_index_1;
        i2[i] = d.matches[i].feature_index_2;
        dist[i] = d.matches[i].distance;
        inl[i] = d.inliers[i];
    }
}

// edge e in insertion order: ids2 = {source, dest}; counts2 = {matches, inlier_matches}; H 9; poses 4x8
void och_graph_edge_info(const och_graph *g, size_t e, uint64_t *ids2, uint64_t *counts2, double *H, double *poses)
{
    const auto &ed = g->graph.edges().at(e);
    ids2[0] = ed.source;
    ids2[1] = ed.dest;
    counts2[0] = ed.payload.matches.size();
    counts2[1] = ed.payload.inlier_matches.size();
    std::memcpy(H, ed.payload.ransac_relation, 72);
    for (int i = 0; i < 4; i++)
    {
        const auto &p = ed.payload.relative_poses[i];
        double *o = poses + 8 * i;
        for (int k = 0; k < 4; k++)
            o[k] = p.orientation[k];
        for (int k = 0; k < 3; k++)
            o[4 + k] = p.position[k];
        o[7] = p.score;
    }
}

void och_graph_edge_match_distances(const och_graph *g, size_t e, double *out)
{
    const auto &ed = g->graph.edges().at(e);
    for (size_t i = 0; i < ed.payload.matches.size(); i++)
        out[i] = ed.payload.matches[i].distance;
}

void och_graph_edge_matches(const och_graph *g, size_t e, uint64_t *idx2, double *dist, int *is_homography)
{
    const auto &ed = g->graph.edges().at(e);
    for (size_t i = 0; i < ed.payload.matches.size(); i++)
    {
        if (idx2)
        {
            idx2[2 * i] = ed.payload.matches[i].feature_index_1;
            idx2[2 * i + 1] = ed.payload.matches[i].feature_index_2;
        }
        if (dist)
            dist[i] = ed.payload.matches[i].distance;
    }
    if (is_homography)
        *is_homography = ed.payload.relationType == camera_relations::RelationType::HOMOGRAPHY;
}

void och_graph_set_orientations(och_graph *g, const double *ori)
{
    auto &nodes = g->graph.nodes();
    for (size_t i = 0; i < nodes.size(); i++)
        std::memcpy(nodes[i].payload.orientation, ori + 4 * i, 32);
}

void och_graph_edge_inliers(const och_graph *g, size_t e, uint64_t *f1, uint64_t *f2, uint64_t *match_index, double *px4)
{
    const auto &ed = g->graph.edges().at(e);
    for (size_t i = 0; i < ed.payload.inlier_matches.size(); i++)
    {
        const auto &m = ed.payload.inlier_matches[i];
        f1[i] = m.feature_index_1;
        f2[i] = m.feature_index_2;
        match_index[i] = m.match_index;
        px4[4 * i] = m.pixel_1[0];
        px4[4 * i + 1] = m.pixel_1[1];
        px4[4 * i + 2] = m.pixel_2[0];
        px4[4 * i + 3] = m.pixel_2[1];
    }
}

} // extern "C"
