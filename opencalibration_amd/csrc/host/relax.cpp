#include "relax.hpp"

#include "ransac.hpp" // image_to_3d

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

namespace opencalibration_amd
{

namespace
{
using clk = std::chrono::steady_clock;
double since(clk::time_point t0)
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
void to_matrix(const double *q, double R[3][3])
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
void ray_intersection(const v3 &d1, const v3 &o1, const v3 &d2, const v3 &o2, v3 *mid, double *err)
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

// One ground-plane problem: host assembly + device solve.
class GroundPlaneProblem
{
  public:
    GroundPlaneProblem(ochip_ctx *ctx, const MeasurementGraph &graph) : _ctx(ctx), _graph(graph)
    {
    }
    ~GroundPlaneProblem()
    {
        if (_dev)
            ochip_relax_problem_destroy(_dev);
    }

    // setupGroundPlaneProblem (relax_problem.cpp:61-81)
    bool setup(std::vector<NodePose> &poses, const std::vector<size_t> &edges_to_optimize, std::string *error,
               const RelaxShard *shard = nullptr)
    {
        const bool verbose = getenv("OCHIP_RELAX_VERBOSE") != nullptr;
        auto tmark = clk::now();
        auto lap = [&](const char *what) {
            if (verbose)
                fprintf(stderr, "[relax setup] %-28s %.3f ms\n", what, since(tmark) * 1e3);
            tmark = clk::now();
        };
        _poses = &poses;
        for (size_t i = 0; i < poses.size(); i++)
            _opt_index.emplace(poses[i].node_id, i);
        initialize_plane();

        // camera table: optimised poses first (their order), context cameras appended on first use
        for (size_t i = 0; i < poses.size(); i++)
        {
            _cam_of_node.emplace(poses[i].node_id, (uint32_t)i);
            push_camera(poses[i].position, poses[i].orientation, true);
        }

        // gridFilterMatchesPerImage (:234-309): an edge without usable poses stops the whole pass
        size_t n_filter = edges_to_optimize.size();
        std::vector<pose_ref> src(edges_to_optimize.size()), dst(edges_to_optimize.size());
        for (size_t k = 0; k < edges_to_optimize.size(); k++)
        {
            const MeasurementGraph::Edge *e = _graph.getEdge(edges_to_optimize[k]);
            if (e == nullptr)
                continue;
            src[k] = lookup(e->source);
            dst[k] = lookup(e->dest);
            if ((src[k].loc == nullptr || dst[k].loc == nullptr) && n_filter == edges_to_optimize.size())
                n_filter = k;
        }
        lap("pose lookup");
        std::vector<std::vector<uint8_t>> keep(edges_to_optimize.size());
#pragma omp parallel for schedule(dynamic, 1)
        for (size_t k = 0; k < n_filter; k++)
        {
            const MeasurementGraph::Edge *e = _graph.getEdge(edges_to_optimize[k]);
            if (e != nullptr)
                keep[k] = grid_filter(*e, src[k], dst[k], 0.15);
        }

        // addRayTriangleMeasurementCost (:388-560), fixed intrinsics.  The searcher's orientation fix-up of
        // the single triangle happens on its first use and is the same for every edge, so do it once and
        // build the per-edge block lists in parallel; they are concatenated in edge order.
        lap("grid filter");
        fix_triangle_orientation();
        std::vector<edge_blocks> per_edge(edges_to_optimize.size());
#pragma omp parallel for schedule(dynamic, 8)
        for (size_t k = 0; k < edges_to_optimize.size(); k++)
        {
            const MeasurementGraph::Edge *e = _graph.getEdge(edges_to_optimize[k]);
            if (e == nullptr || src[k].loc == nullptr || dst[k].loc == nullptr)
                continue;
            add_edge_blocks(*e, src[k], dst[k], keep[k], per_edge[k]);
        }
        lap("edge blocks");
        size_t total_blocks = 0;
        for (const auto &pe : per_edge)
            total_blocks += pe.a.size();
        // concatenated in edge order (offsets first, then a parallel copy)
        std::vector<size_t> first_block(per_edge.size() + 1, 0);
        for (size_t k = 0; k < per_edge.size(); k++)
            first_block[k + 1] = first_block[k] + per_edge[k].a.size();
        _blk_a.resize(total_blocks);
        _blk_b.resize(total_blocks);
        _blk_rays.resize(total_blocks * 6);
#pragma omp parallel for schedule(static)
        for (size_t k = 0; k < per_edge.size(); k++)
        {
            const auto &pe = per_edge[k];
            std::copy(pe.a.begin(), pe.a.end(), _blk_a.begin() + first_block[k]);
            std::copy(pe.b.begin(), pe.b.end(), _blk_b.begin() + first_block[k]);
            std::copy(pe.rays.begin(), pe.rays.end(), _blk_rays.begin() + 6 * first_block[k]);
        }
        // addDownwardsPrior (:1290-1301)
        for (size_t i = 0; i < poses.size(); i++)
            if (!hasnan4(poses[i].orientation))
                _prior_cam.push_back((uint32_t)i);

        lap("concatenate");
        ochip_relax_desc d{};
        d.n_cams = (uint32_t)_cam_opt.size();
        d.cam_pos = _cam_pos.data();
        d.cam_q = _cam_q.data();
        d.cam_optimize = _cam_opt.data();
        for (int i = 0; i < 3; i++)
        {
            d.plane_xy[2 * i] = _xy[_tri[i]][0];
            d.plane_xy[2 * i + 1] = _xy[_tri[i]][1];
            d.plane_z[i] = _z[_tri[i]];
            d.z_optimize[i] = 1;
        }
        d.n_blocks = (uint32_t)_blk_a.size();
        d.blk_cam_a = _blk_a.data();
        d.blk_cam_b = _blk_b.data();
        d.blk_rays = _blk_rays.data();
        d.n_prior = (uint32_t)_prior_cam.size();
        d.prior_cam = _prior_cam.data();
        d.huber_a = 1 * M_PI / 180; // HuberLoss(1 degree), :68
        d.prior_weight = 1e-3;
        if (ochip_relax_problem_create(_ctx, &d, &_dev) != OCHIP_OK)
        {
            *error = std::string("ochip_relax_problem_create: ") + ochip_last_error(_ctx);
            return false;
        }
        if (shard && shard->world > 1 &&
            ochip_relax_set_shard(_dev, shard->rank, shard->world, shard->exchange, shard->user) != OCHIP_OK)
        {
            *error = std::string("ochip_relax_set_shard: ") + ochip_last_error(_ctx);
            return false;
        }
        lap("ochip_relax_problem_create");
        return true;
    }

    // relaxObservedModelOnly (:931-984) then solve (:1390-1420)
    bool relax_observed_model_only(RelaxTimers *t, std::string *error)
    {
        if (ochip_relax_set_cameras_constant(_dev, 1) != OCHIP_OK)
            return fail(error, "ochip_relax_set_cameras_constant");
        const bool ok = solve(t, error);
        if (ochip_relax_set_cameras_constant(_dev, 0) != OCHIP_OK)
            return fail(error, "ochip_relax_set_cameras_constant");
        return ok;
    }

    bool solve(RelaxTimers *t, std::string *error)
    {
        if (_blk_a.empty() && _prior_cam.empty())
            return true; // NumResidualBlocks() == 0: early exit of :1398-1402
        ochip_relax_options o{100, 1.0, 1e-6, 1e-10, 1e-8}; // relax_problem.cpp:30-37 + Ceres defaults
        ochip_relax_summary s{};
        if (ochip_relax_solve(_dev, &o, &s) != OCHIP_OK)
            return fail(error, "ochip_relax_solve");
        if (t)
        {
            t->solves++;
            t->iterations_total += s.iterations;
            t->last_iterations = s.iterations;
            t->last_initial_cost = s.initial_cost;
            t->last_final_cost = s.final_cost;
            t->last_residual_blocks = s.num_residual_blocks;
        }
        std::vector<double> q(_cam_opt.size() * 4);
        double z[3];
        if (ochip_relax_get_state(_dev, q.data(), z) != OCHIP_OK)
            return fail(error, "ochip_relax_get_state");
        for (size_t i = 0; i < _poses->size(); i++) // p.second->orientation.normalize(), :1410-1413
        {
            double *o4 = (*_poses)[i].orientation;
            const double *s4 = &q[4 * i];
            const double n = std::sqrt(s4[0] * s4[0] + s4[1] * s4[1] + s4[2] * s4[2] + s4[3] * s4[3]);
            for (int k = 0; k < 4; k++)
                o4[k] = s4[k] / n;
        }
        for (int i = 0; i < 3; i++)
            _z[_tri[i]] = z[i];
        return true;
    }

    void surface(surface_model_plane *out) const
    {
        for (int i = 0; i < 3; i++)
        {
            out->corner[i][0] = _xy[i][0];
            out->corner[i][1] = _xy[i][1];
            out->corner[i][2] = _z[i];
        }
    }

  private:
    bool fail(std::string *error, const char *what)
    {
        *error = std::string(what) + ": " + ochip_last_error(_ctx);
        return false;
    }
    void push_camera(const double *pos, const double *q, bool optimize)
    {
        _cam_pos.insert(_cam_pos.end(), pos, pos + 3);
        _cam_q.insert(_cam_q.end(), q, q + 4);
        _cam_opt.push_back(optimize ? 1 : 0);
    }
    pose_ref lookup(size_t node_id) // nodeid2poseopt
    {
        pose_ref po;
        auto it = _opt_index.find(node_id);
        if (it != _opt_index.end())
        {
            NodePose &np = (*_poses)[it->second];
            po.optimize = true;
            po.loc = np.position;
            po.rot = np.orientation;
            po.cam = (uint32_t)it->second;
            return po;
        }
        const MeasurementGraph::Node *node = _graph.getNode(node_id);
        if (node != nullptr && finite4(node->payload.orientation) && finite3(node->payload.position))
        {
            po.loc = node->payload.position;
            po.rot = node->payload.orientation;
            auto c = _cam_of_node.find(node_id);
            if (c == _cam_of_node.end())
            {
                c = _cam_of_node.emplace(node_id, (uint32_t)_cam_opt.size()).first;
                push_camera(po.loc, po.rot, false);
            }
            po.cam = c->second;
        }
        return po;
    }

    void initialize_plane() // initializeGroundPlane (:1189-1242)
    {
        double lo[2] = {1e12, 1e12}, hi[2] = {-1e12, -1e12}, height = 0;
        for (const NodePose &p : *_poses)
        {
            for (int a = 0; a < 2; a++)
            {
                lo[a] = std::min(lo[a], p.position[a]);
                hi[a] = std::max(hi[a], p.position[a]);
            }
            height += p.position[2];
        }
        height /= (double)_poses->size();
        const double margin = 50;
        height -= margin;
        const double cx = (lo[0] + hi[0]) / 2, cy = (lo[1] + hi[1]) / 2;
        const double spacing = std::max(hi[0] - lo[0], hi[1] - lo[1]) + margin;
        const double c[3][2] = {{-spacing + cx, -spacing + cy}, {spacing + cx, -spacing + cy}, {0 + cx, spacing + cy}};
        for (int i = 0; i < 3; i++)
        {
            _xy[i][0] = c[i][0];
            _xy[i][1] = c[i][1];
            _z[i] = height;
        }
    }

    // Scores of gridFilterMatchesPerImage + GridFilter::addMeasurement (grid_filter.hpp:33-51): the
    // measurements arrive best-first, so the first one in a cell stays.  Returns per inlier: bit0 = on the
    // source whitelist, bit1 = on the dest whitelist.
    std::vector<uint8_t> grid_filter(const MeasurementGraph::Edge &edge, const pose_ref &s, const pose_ref &d,
                                     double res) const
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
        constexpr int G = 16; // cells per axis the table covers (pixels / image size lies in [0, 1): 1 / 0.15 < 7)
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

    // MeshIntersectionSearcher::triangleIntersect (src/surface/intersect.cpp:56-163) on the single
    // border triangle: orientation fix-up (persistent), then the three edge tests.
    static bool anticlockwise(const double *a, const double *b, const double *c)
    {
        return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]) < 0;
    }
    void fix_triangle_orientation()
    {
        if (anticlockwise(_xy[_tri[0]], _xy[_tri[1]], _xy[_tri[2]]))
            std::swap(_tri[0], _tri[1]);
    }
    bool vertical_ray_hits_plane_triangle(double px, double py) const
    {
        const double P[2] = {px, py};
        for (int i = 0; i < 3; i++)
            if (anticlockwise(P, _xy[_tri[i]], _xy[_tri[(i + 1) % 3]]))
                return false;
        return true;
    }

    struct edge_blocks
    {
        std::vector<uint32_t> a, b;
        std::vector<double> rays;
    };

    void add_edge_blocks(const MeasurementGraph::Edge &edge, const pose_ref &s, const pose_ref &d,
                         const std::vector<uint8_t> &keep, edge_blocks &out) const
    {
        const camera_relations &rel = edge.payload;
        const CameraModel &sm = *_graph.getNode(edge.source)->payload.model, &dm = *_graph.getNode(edge.dest)->payload.model;
        const v3 so{s.loc[0], s.loc[1], s.loc[2]}, d_o{d.loc[0], d.loc[1], d.loc[2]};
        for (size_t idx = 0; idx < rel.inlier_matches.size(); idx++)
        {
            if (idx >= keep.size() || keep[idx] == 0)
                continue;
            const feature_match_denormalized &m = rel.inlier_matches[idx];
            double r1[3], r2[3];
            image_to_3d(m.pixel_1, sm, r1);
            image_to_3d(m.pixel_2, dm, r2);
            v3 mid;
            double gap;
            ray_intersection(rotate(s.rot, v3{r1[0], r1[1], r1[2]}), so, rotate(d.rot, v3{r2[0], r2[1], r2[2]}), d_o, &mid,
                             &gap);
            if (std::isnan(mid.x) || std::isnan(mid.y))
                continue;
            if (!vertical_ray_hits_plane_triangle(mid.x, mid.y))
                continue;
            out.a.push_back(s.cam);
            out.b.push_back(d.cam);
            out.rays.insert(out.rays.end(), r1, r1 + 3);
            out.rays.insert(out.rays.end(), r2, r2 + 3);
        }
    }

    ochip_ctx *_ctx;
    const MeasurementGraph &_graph;
    std::vector<NodePose> *_poses = nullptr;
    std::unordered_map<size_t, size_t> _opt_index;
    std::unordered_map<size_t, uint32_t> _cam_of_node;
    std::vector<double> _cam_pos, _cam_q, _blk_rays;
    std::vector<uint8_t> _cam_opt;
    std::vector<uint32_t> _blk_a, _blk_b, _prior_cam;
    double _xy[3][2], _z[3];
    int _tri[3] = {0, 1, 2};
    ochip_relax_problem *_dev = nullptr;
};

} // namespace

bool relax_ground_plane(ochip_ctx *ctx, const MeasurementGraph &graph, std::vector<NodePose> &nodes,
                        const std::vector<size_t> &edges_to_optimize, surface_model_plane *surface,
                        RelaxTimers *timers, std::string *error, const RelaxShard *shard)
{
    // runGroundPlane (src/relax/relax.cpp:44-87)
    auto run = [&](std::vector<NodePose> &poses, surface_model_plane *out) -> bool {
        auto t0 = clk::now();
        GroundPlaneProblem rp(ctx, graph);
        if (!rp.setup(poses, edges_to_optimize, error, shard))
            return false;
        if (timers)
            timers->setup_host += since(t0);
        t0 = clk::now();
        const bool ok = rp.relax_observed_model_only(timers, error) && rp.solve(timers, error);
        if (timers)
            timers->device += since(t0);
        if (ok && out)
            rp.surface(out);
        return ok;
    };
    // DOWN_ORIENTED_NORTH = Quaterniond(AngleAxisd(M_PI, UnitX)), relax.cpp:12
    double previous[4] = {std::sin(M_PI / 2), 0.0, 0.0, std::cos(M_PI / 2)};
    for (auto &node : nodes)
    {
        if (hasnan4(node.orientation))
        {
            std::memcpy(node.orientation, previous, sizeof previous);
            if (graph.size_nodes() > 2 * nodes.size())
            {
                std::vector<NodePose> justThis{node};
                if (!run(justThis, nullptr))
                    return false;
                node = justThis[0];
            }
            else if (!run(nodes, nullptr))
                return false;
        }
        std::memcpy(previous, node.orientation, sizeof previous);
    }
    return run(nodes, surface);
}

} // namespace opencalibration_amd
