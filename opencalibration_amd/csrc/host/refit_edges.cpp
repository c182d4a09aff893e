// The per-edge loop RelaxGroup::finalize runs after a relax changed a camera model
// (src/relax/relax_group.cpp:137-177): every edge's correspondences are recomputed with the current models
// (distort_keypoints), its homography is re-fitted on its previous inliers three times (fitInliers + evaluate, "a sort
// of maximum likelihood based on the previous inliers"), decomposed again, and the inlier list is reassembled.  The fits
// and evaluations of all edges are one device launch (ochip_refit_homography_batch, one wavefront per edge); the
// decomposition and the assembly stay on the host like in the link stage.
#include "../../../include/oc_host.h"

#include "capi_graph.hpp"
#include "ransac.hpp"

#include <algorithm>
#include <cstring>

using namespace opencalibration_amd;

extern "C" int och_graph_set_model(och_graph *g, uint32_t model, const double *m10)
{
    if (!g || !m10 || model >= g->models.size())
        return -1;
    CameraModel &m = *g->models[model]; // shared by the images that use it, like the reference's shared_ptr<CameraModel>
    m.focal_length_pixels = m10[0];
    m.principle_point[0] = m10[1];
    m.principle_point[1] = m10[2];
    for (int i = 0; i < 3; i++)
        m.radial_distortion[i] = m10[3 + i];
    m.tangential_distortion[0] = m10[6];
    m.tangential_distortion[1] = m10[7];
    return 0;
}

extern "C" int och_graph_refit_edges(och_graph *g, ochip_ctx *ctx)
{
    if (!g || !ctx)
        return -1;
    MeasurementGraph &graph = g->graph;
    auto &edges = graph.edges();
    const auto &nodes = graph.nodes();
    if (edges.empty())
        return 0;
    // ---- the features the edges' matches use, per image, in ascending feature order -> device slots
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    std::vector<std::vector<uint32_t>> pos_of(nodes.size()); // node -> feature index -> position in the slot
    std::vector<std::vector<size_t>> used(nodes.size());
    for (const auto &e : edges)
    {
        if (e.payload.matches.empty())
            continue;
        const size_t s = graph.nodeIndex(e.source), d = graph.nodeIndex(e.dest);
        if (pos_of[s].empty())
            pos_of[s].assign(nodes[s].payload.features.size(), NONE);
        if (pos_of[d].empty())
            pos_of[d].assign(nodes[d].payload.features.size(), NONE);
        for (const feature_match &m : e.payload.matches)
        {
            pos_of[s][m.feature_index_1] = 0;
            pos_of[d][m.feature_index_2] = 0;
        }
    }
    std::vector<uint32_t> slot_of_node(nodes.size(), NONE), counts;
    std::vector<size_t> slot_node;
    for (size_t n = 0; n < nodes.size(); n++)
    {
        if (pos_of[n].empty())
            continue;
        for (size_t f = 0; f < pos_of[n].size(); f++)
            if (pos_of[n][f] != NONE)
            {
                pos_of[n][f] = (uint32_t)used[n].size();
                used[n].push_back(f);
            }
        slot_of_node[n] = (uint32_t)slot_node.size();
        slot_node.push_back(n);
        counts.push_back((uint32_t)used[n].size());
    }
    if (slot_node.empty())
    {
        // no edge kept its matches: one empty slot so that the jobs below have an image to name
        slot_node.push_back(0);
        counts.push_back(0);
    }
    const size_t n_slots = slot_node.size();
    std::vector<uint64_t> slot_off(n_slots + 1, 0);
    for (size_t s = 0; s < n_slots; s++)
        slot_off[s + 1] = slot_off[s] + counts[s];
    std::vector<uint64_t> dbuf(std::max<uint64_t>(slot_off[n_slots], 1) * 8);
    std::vector<double> xybuf(std::max<uint64_t>(slot_off[n_slots], 1) * 2), models(n_slots * 8);
    for (size_t s = 0; s < n_slots; s++)
    {
        const image &img = nodes[slot_node[s]].payload;
        for (size_t k = 0; k < counts[s]; k++)
        {
            const feature_2d &f = img.features[used[slot_node[s]][k]];
            std::memcpy(&dbuf[(slot_off[s] + k) * 8], f.descriptor, 64);
            xybuf[(slot_off[s] + k) * 2] = f.location[0];
            xybuf[(slot_off[s] + k) * 2 + 1] = f.location[1];
        }
        const CameraModel &m = *img.model;
        const double model8[8] = {m.focal_length_pixels,      m.principle_point[0],      m.principle_point[1],
                                  m.radial_distortion[0],     m.radial_distortion[1],    m.radial_distortion[2],
                                  m.tangential_distortion[0], m.tangential_distortion[1]};
        std::memcpy(&models[s * 8], model8, sizeof model8);
    }
    if (ochip_upload_batch(ctx, (uint32_t)n_slots, counts.data(), dbuf.data(), xybuf.data(), models.data()) != OCHIP_OK)
    {
        g->error = std::string("ochip_upload_batch: ") + ochip_last_error(ctx);
        return -1;
    }
    // ---- one job per edge (an edge without matches is an empty job: the reference re-fits those too)
    std::vector<ochip_ransac_job> jobs(edges.size());
    uint64_t total = 0;
    for (size_t i = 0; i < edges.size(); i++)
    {
        const auto &e = edges[i];
        const bool has = !e.payload.matches.empty();
        jobs[i] = ochip_ransac_job{has ? slot_of_node[graph.nodeIndex(e.source)] : 0u, has ? slot_of_node[graph.nodeIndex(e.dest)] : 0u,
                                   (uint32_t)e.payload.matches.size(), 0u, total, 0};
        total += e.payload.matches.size();
    }
    std::vector<ochip_ransac_match> rm(std::max<uint64_t>(total, 1));
    std::vector<uint8_t> inl(std::max<uint64_t>(total, 1), 0);
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < edges.size(); i++)
    {
        const auto &e = edges[i];
        const size_t s = graph.nodeIndex(e.source), d = graph.nodeIndex(e.dest);
        for (size_t k = 0; k < e.payload.matches.size(); k++)
        {
            const feature_match &m = e.payload.matches[k];
            rm[jobs[i].match_offset + k] = ochip_ransac_match{pos_of[s][m.feature_index_1], pos_of[d][m.feature_index_2], 0, 0};
        }
        for (const auto &old_inlier : e.payload.inlier_matches) // relax_group.cpp:152-155
            inl[jobs[i].match_offset + old_inlier.match_index] = 1;
    }
    std::vector<ochip_ransac_result> results(edges.size());
    const homography_model defaults;
    if (ochip_refit_homography_batch(ctx, jobs.data(), (uint32_t)jobs.size(), rm.data(), total, 3, defaults.inlier_threshold,
                                     results.data(), inl.data()) != OCHIP_OK)
    {
        g->error = std::string("ochip_refit_homography_batch: ") + ochip_last_error(ctx);
        return -1;
    }
    // ---- decompose, accept, assemble (relax_group.cpp:163-174)
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t i = 0; i < edges.size(); i++)
    {
        auto &e = edges[i];
        const image &src = nodes[graph.nodeIndex(e.source)].payload, &dst = nodes[graph.nodeIndex(e.dest)].payload;
        const size_t M = e.payload.matches.size();
        homography_model h;
        std::memcpy(h.homography, results[i].H, sizeof h.homography);
        std::memcpy(e.payload.ransac_relation, results[i].H, sizeof e.payload.ransac_relation);
        e.payload.relationType = camera_relations::RelationType::HOMOGRAPHY;
        std::vector<bool> inliers(M);
        std::vector<double> inlier_rays;
        size_t num_inliers = 0;
        for (size_t k = 0; k < M; k++)
        {
            inliers[k] = inl[jobs[i].match_offset + k] != 0;
            if (inliers[k])
            {
                double r1[3], r2[3];
                image_to_3d(src.features[e.payload.matches[k].feature_index_1].location, *src.model, r1);
                image_to_3d(dst.features[e.payload.matches[k].feature_index_2].location, *dst.model, r2);
                inlier_rays.insert(inlier_rays.end(), r1, r1 + 3);
                inlier_rays.insert(inlier_rays.end(), r2, r2 + 3);
                num_inliers++;
            }
        }
        const bool can_decompose = h.decompose_inlier_rays(inlier_rays.data(), num_inliers, e.payload.relative_poses);
        e.payload.inlier_matches.clear();
        if (can_decompose && num_inliers > h.MINIMUM_POINTS * 1.5)
            assembleInliers(e.payload.matches, inliers, src.features, dst.features, e.payload.inlier_matches);
    }
    return 0;
}
