// Helpers shared by the host assembly of the relax flavours (relax.cpp: ground plane on the pair-record engine;
// relax_mesh.cpp: mesh / tracks / intrinsics on the general engine).
#pragma once

#include "ransac.hpp" // image_to_3d
#include "relax.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace opencalibration_amd
{
namespace relax_detail
{
using clk = std::chrono::steady_clock;
inline double since(clk::time_point t0)
{
    return std::chrono::duration<double>(clk::now() - t0).count();
}

struct v3
{
    double x, y, z;
};
inline v3 sub(const v3 &a, const v3 &b)
{
    return {a.x - b.x, a.y - b.y, a.z - b.z};
}
inline v3 add(const v3 &a, const v3 &b)
{
    return {a.x + b.x, a.y + b.y, a.z + b.z};
}
inline v3 mul(const v3 &a, double s)
{
    return {a.x * s, a.y * s, a.z * s};
}
inline double dot(const v3 &a, const v3 &b)
{
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
inline v3 cross(const v3 &a, const v3 &b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline bool finite4(const double *q)
{
    return std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]) && std::isfinite(q[3]);
}
inline bool finite3(const double *p)
{
    return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]);
}
inline bool hasnan4(const double *q)
{
    return std::isnan(q[0]) || std::isnan(q[1]) || std::isnan(q[2]) || std::isnan(q[3]);
}

// Eigen::Quaternion::toRotationMatrix
inline void to_matrix(const double *q, double R[3][3])
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0][0] = 1 - (tyy + tzz), R[0][1] = txy - twz, R[0][2] = txz + twy;
    R[1][0] = txy + twz, R[1][1] = 1 - (txx + tzz), R[1][2] = tyz - twx;
    R[2][0] = txz - twy, R[2][1] = tyz + twx, R[2][2] = 1 - (txx + tyy);
}
inline v3 apply(const double R[3][3], const v3 &v)
{
    return {R[0][0] * v.x + R[0][1] * v.y + R[0][2] * v.z, R[1][0] * v.x + R[1][1] * v.y + R[1][2] * v.z,
            R[2][0] * v.x + R[2][1] * v.y + R[2][2] * v.z};
}
// Eigen QuaternionBase::_transformVector
inline v3 rotate(const double *q, const v3 &v)
{
    const v3 qv{q[0], q[1], q[2]};
    v3 uv = cross(qv, v);
    uv = add(uv, uv);
    return add(add(v, mul(uv, q[3])), cross(qv, uv));
}

// src/geometry/intersection.cpp:116-143: midpoint of closest approach, signed squared gap
inline void ray_intersection(const v3 &d1, const v3 &o1, const v3 &d2, const v3 &o2, v3 *mid, double *err)
{
    *mid = {NAN, NAN, NAN};
    *err = NAN;
    const double n11 = dot(d1, d1), n12 = dot(d1, d2), n22 = dot(d2, d2);
    const double denom = n11 * n22 - n12 * n12;
    if (std::abs(denom) > 1e-9)
    {
        const v3 off = sub(o1, o2);
        const double od1 = dot(off, d1), od2 = dot(off, d2);
        const double t = (n12 * od2 - n22 * od1) / denom;
        const double s = (n11 * od2 - n12 * od1) / denom;
        const v3 p1 = add(o1, mul(d1, t)), p2 = add(o2, mul(d2, s));
        *mid = mul(add(p1, p2), 0.5);
        const v3 g = sub(p1, p2);
        *err = dot(g, g) * (t >= 0 && s >= 0 ? 1 : -1);
    }
}

struct pose_ref // OptimizationPackage::PoseOpt (relax_problem.cpp:182-232)
{
    bool optimize = false;
    const double *loc = nullptr;
    const double *rot = nullptr;
    uint32_t cam = 0; // index into the device camera table
};


// Scores of gridFilterMatchesPerImage + GridFilter::addMeasurement (grid_filter.hpp:33-51): the
// measurements arrive best-first, so the first one in a cell stays.  Returns per inlier: bit0 = on the
// source whitelist, bit1 = on the dest whitelist.
inline std::vector<uint8_t> grid_filter(const MeasurementGraph &_graph, const MeasurementGraph::Edge &edge, const pose_ref &s,
                                    const pose_ref &d, double res)
{
    const camera_relations &rel = edge.payload;
    const CameraModel &sm = *_graph.getNode(edge.source)->payload.model, &dm = *_graph.getNode(edge.dest)->payload.model;
    double Rs[3][3], Rd[3][3];
    to_matrix(s.rot, Rs);
    to_matrix(d.rot, Rd);
    const v3 so{s.loc[0], s.loc[1], s.loc[2]}, d_o{d.loc[0], d.loc[1], d.loc[2]};
    std::vector<std::pair<double, size_t>> scored;
    scored.reserve(rel.inlier_matches.size());
    for (size_t idx = 0; idx < rel.inlier_matches.size(); idx++)
    {
        const feature_match_denormalized &m = rel.inlier_matches[idx];
        double r1[3], r2[3];
        image_to_3d(m.pixel_1, sm, r1);
        image_to_3d(m.pixel_2, dm, r2);
        const v3 sd = apply(Rs, v3{r1[0], r1[1], r1[2]}), dd = apply(Rd, v3{r2[0], r2[1], r2[2]});
        v3 mid;
        double gap;
        ray_intersection(sd, so, dd, d_o, &mid, &gap);
        const double intersection_score = gap < 0 ? 0. : 1. / (1. + gap);
        const double cos_angle = dot(sd, dd);
        const double angle_score = 1.0 - cos_angle * cos_angle;
        const double descriptor_score =
            m.match_index < rel.matches.size() ? 1.0 - rel.matches[m.match_index].distance : 1.0;
        double ransac_score = 1.0;
        if (rel.relationType == camera_relations::RelationType::HOMOGRAPHY)
        {
            const double sx = (m.pixel_1[0] - sm.principle_point[0]) / sm.focal_length_pixels;
            const double sy = (m.pixel_1[1] - sm.principle_point[1]) / sm.focal_length_pixels;
            const double dx = (m.pixel_2[0] - dm.principle_point[0]) / dm.focal_length_pixels;
            const double dy = (m.pixel_2[1] - dm.principle_point[1]) / dm.focal_length_pixels;
            const double *H = rel.ransac_relation;
            const double hx = H[0] * sx + H[1] * sy + H[2] * 1.0, hy = H[3] * sx + H[4] * sy + H[5] * 1.0,
                         hz = H[6] * sx + H[7] * sy + H[8] * 1.0;
            const double ex = dx - hx / hz, ey = dy - hy / hz;
            ransac_score = 1.0 / (1.0 + std::sqrt(ex * ex + ey * ey));
        }
        scored.emplace_back(intersection_score * angle_score * descriptor_score * ransac_score, idx);
    }
    std::vector<uint8_t> keep(rel.inlier_matches.size(), 0);
    auto cell = [res](double x, double y, int *cx, int *cy) {
        *cx = (int)std::floor(x / res);
        *cy = (int)std::floor(y / res);
    };
    // GridFilter::addMeasurement keeps, per cell of each image, the first measurement in descending score order,
    // i.e. the best-scoring one.  That needs no sort unless two candidates for a cell's best tie exactly (the
    // unstable std::sort then decides): one pass over a small dense cell table, and the sorted walk only as the
    // fall-back for ties or cells outside the table.
    constexpr int G = 48; // cells per axis the table covers (pixels / image size lies in [0, 1): 1 / 0.15 < 7; a mesh run at grid fraction 0.025 has 40)
    int best_s[G * G], best_d[G * G];
    std::fill(best_s, best_s + G * G, -1);
    std::fill(best_d, best_d + G * G, -1);
    bool exact = true;
    for (size_t k = 0; k < scored.size() && exact; k++)
    {
        const double score = scored[k].first;
        if (!(score > 0))
            continue;
        const feature_match_denormalized &m = rel.inlier_matches[scored[k].second];
        int cx, cy, dx, dy;
        cell(m.pixel_1[0] / sm.pixels_cols, m.pixel_1[1] / sm.pixels_rows, &cx, &cy);
        cell(m.pixel_2[0] / dm.pixels_cols, m.pixel_2[1] / dm.pixels_rows, &dx, &dy);
        if (cx < 0 || cy < 0 || cx >= G || cy >= G || dx < 0 || dy < 0 || dx >= G || dy >= G)
        {
            exact = false;
            break;
        }
        int &bs = best_s[cx * G + cy], &bd = best_d[dx * G + dy];
        if (bs < 0 || score > scored[bs].first)
            bs = (int)k;
        else if (score == scored[bs].first)
            exact = false;
        if (bd < 0 || score > scored[bd].first)
            bd = (int)k;
        else if (score == scored[bd].first)
            exact = false;
    }
    if (exact)
    {
        for (int c = 0; c < G * G; c++)
        {
            if (best_s[c] >= 0)
                keep[scored[best_s[c]].second] |= 1;
            if (best_d[c] >= 0)
                keep[scored[best_d[c]].second] |= 2;
        }
        return keep;
    }
    std::fill(keep.begin(), keep.end(), 0);
    std::sort(scored.begin(), scored.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
    std::unordered_map<uint64_t, char> scell, dcell;
    auto key = [res](double x, double y) {
        return (static_cast<uint64_t>((int)std::floor(x / res)) << 32) | static_cast<uint32_t>((int)std::floor(y / res));
    };
    for (const auto &[score, idx] : scored)
    {
        if (!(score > 0))
            continue;
        const feature_match_denormalized &m = rel.inlier_matches[idx];
        if (scell.emplace(key(m.pixel_1[0] / sm.pixels_cols, m.pixel_1[1] / sm.pixels_rows), 1).second)
            keep[idx] |= 1;
        if (dcell.emplace(key(m.pixel_2[0] / dm.pixels_cols, m.pixel_2[1] / dm.pixels_rows), 1).second)
            keep[idx] |= 2;
    }
    return keep;
}


// gridFilterMatchesPerImage (relax_problem.cpp:234-309) of a list of edges on the device (ochip_plane_setup_create,
// csrc/relax_setup.hip): the edges' inlier matches are packed into page-locked staging (40 bytes each), one wavefront per
// edge scores and filters them, and the edges the device only flags - two matches sharing a cell's best score, which the
// reference's unstable std::sort decides, or a match outside the cell table - are decided by grid_filter above.
// keep: the edges' flags back to back (edge j's start at pe[j].inlier_offset).  The handle stays open for the caller
// (the ground-plane set-up goes on to ochip_plane_setup_blocks); destroy it with ochip_plane_setup_destroy.
struct device_filter
{
    std::vector<ochip_plane_edge> pe;
    std::vector<const MeasurementGraph::Edge *> edge;
    std::vector<pose_ref> src, dst;
    std::vector<uint8_t> keep;
    ochip_plane_setup *handle = nullptr;
    double seconds_pack = 0;
    ~device_filter()
    {
        if (handle)
            ochip_plane_setup_destroy(handle);
    }
    // edges[k] with poses src[k] / dst[k]; nullptr entries are skipped.  triangle: the plane's border triangle (only the
    // block list uses it).  Returns false with *error set when a device call fails.
    bool run(ochip_ctx *ctx, const MeasurementGraph &graph, const std::vector<const MeasurementGraph::Edge *> &edges,
             const std::vector<pose_ref> &srcs, const std::vector<pose_ref> &dsts, size_t n_edges, const std::vector<double> &cam_pos,
             const std::vector<double> &cam_q, const double triangle_xy6[6], double fraction, std::string *error)
    {
        const auto t0 = clk::now();
        std::unordered_map<const CameraModel *, uint32_t> model_index;
        std::vector<double> models10;
        auto model_of = [&](const CameraModel *m) {
            auto it = model_index.find(m);
            if (it != model_index.end())
                return it->second;
            const double row[10] = {m->focal_length_pixels,   m->principle_point[0],   m->principle_point[1],      m->radial_distortion[0],
                                    m->radial_distortion[1],  m->radial_distortion[2], m->tangential_distortion[0], m->tangential_distortion[1],
                                    (double)m->pixels_cols,   (double)m->pixels_rows};
            models10.insert(models10.end(), row, row + 10);
            return model_index.emplace(m, (uint32_t)model_index.size()).first->second;
        };
        uint64_t n_inliers = 0;
        for (size_t k = 0; k < n_edges; k++)
        {
            const MeasurementGraph::Edge *e = edges[k];
            if (e == nullptr)
                continue;
            const camera_relations &rel = e->payload;
            ochip_plane_edge r{};
            r.cam_a = srcs[k].cam;
            r.cam_b = dsts[k].cam;
            r.model_a = model_of(graph.getNode(e->source)->payload.model.get());
            r.model_b = model_of(graph.getNode(e->dest)->payload.model.get());
            r.n_inliers = (uint32_t)rel.inlier_matches.size();
            r.flags = rel.relationType == camera_relations::RelationType::HOMOGRAPHY ? 1u : 0u;
            r.inlier_offset = n_inliers;
            std::memcpy(r.H, rel.ransac_relation, sizeof r.H);
            n_inliers += r.n_inliers;
            pe.push_back(r);
            edge.push_back(e);
            src.push_back(srcs[k]);
            dst.push_back(dsts[k]);
        }
        // page-locked staging from the context's pool: no first-touch faults on ~40 bytes x every inlier of the survey,
        // and the upload runs at the link rate
        struct staging
        {
            ochip_ctx *ctx;
            void *p = nullptr;
            ~staging()
            {
                if (p)
                    ochip_host_free(ctx, p);
            }
        } inl{ctx};
        if (ochip_host_alloc(ctx, (n_inliers ? n_inliers : 1) * sizeof(ochip_plane_inlier), &inl.p) != OCHIP_OK)
        {
            if (error)
                *error = std::string("ochip_host_alloc: ") + ochip_last_error(ctx);
            return false;
        }
        ochip_plane_inlier *const rec = static_cast<ochip_plane_inlier *>(inl.p);
#pragma omp parallel for schedule(dynamic, 16)
        for (size_t j = 0; j < pe.size(); j++)
        {
            const camera_relations &rel = edge[j]->payload;
            ochip_plane_inlier *o = rec + pe[j].inlier_offset;
            for (size_t idx = 0; idx < rel.inlier_matches.size(); idx++)
            {
                const feature_match_denormalized &m = rel.inlier_matches[idx];
                o[idx].px1[0] = m.pixel_1[0], o[idx].px1[1] = m.pixel_1[1];
                o[idx].px2[0] = m.pixel_2[0], o[idx].px2[1] = m.pixel_2[1];
                o[idx].descriptor_score = m.match_index < rel.matches.size() ? 1.0 - rel.matches[m.match_index].distance : 1.0;
            }
        }
        seconds_pack = since(t0);
        keep.assign(n_inliers, 0);
        std::vector<uint8_t> inexact(pe.size());
        if (ochip_plane_setup_create(ctx, pe.data(), (uint32_t)pe.size(), rec, n_inliers, cam_pos.data(), cam_q.data(),
                                     (uint32_t)(cam_q.size() / 4), models10.data(), (uint32_t)model_index.size(), triangle_xy6, fraction,
                                     keep.data(), inexact.data(), &handle) != OCHIP_OK)
        {
            if (error)
                *error = std::string("ochip_plane_setup_create: ") + ochip_last_error(ctx);
            return false;
        }
        for (size_t j = 0; j < pe.size(); j++)
            if (inexact[j])
            {
                const std::vector<uint8_t> k8 = grid_filter(graph, *edge[j], src[j], dst[j], fraction);
                std::copy(k8.begin(), k8.end(), keep.begin() + pe[j].inlier_offset);
                if (ochip_plane_setup_override(handle, pe[j].inlier_offset, k8.size(), k8.data()) != OCHIP_OK)
                {
                    if (error)
                        *error = std::string("ochip_plane_setup_override: ") + ochip_last_error(ctx);
                    return false;
                }
            }
        return true;
    }
};


} // namespace relax_detail
} // namespace opencalibration_amd
