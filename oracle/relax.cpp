// ORACLE — test infrastructure only (see oracle.hpp).
// Restates the ground-plane flavour of the relax stage:
//   src/relax/relax.cpp:44-87 (runGroundPlane), src/relax/relax_problem.cpp:21-38 (options),
//   :61-81 (setupGroundPlaneProblem), :146-232 (initialize / nodeid2poseopt), :234-309
//   (gridFilterMatchesPerImage), :388-560 (addRayTriangleMeasurementCost, fixed-intrinsics branch),
//   :931-984 (relaxObservedModelOnly), :1189-1242 (initializeGroundPlane), :1290-1301 (addDownwardsPrior),
//   :1390-1420 (solve); include/opencalibration/relax/grid_filter.hpp; src/geometry/intersection.cpp:116-143;
//   src/surface/intersect.cpp:10-163 specialised to the single border triangle of the ground plane.
#include "mini_ceres.hpp"
#include "oracle.hpp"
#include "relax_functors.hpp"

#include <algorithm>
#include <cstring>
#include <map>
#include <unordered_map>
#include <unordered_set>

namespace oracle
{

struct relax_edge // MeasurementGraph::Edge with camera_relations payload (fields the relax stage reads)
{
    size_t source, dest; // node indices
    Mat3 ransac_relation;
    bool is_homography = true;
    std::vector<feature_match_denormalized> inlier_matches;
    std::vector<double> match_distance; // relations.matches[i].distance, may be empty
};
struct relax_node // image fields the relax stage reads
{
    Vec3 position;
    Quat orientation;
    camera_model model;
};
struct NodePose // types/node_pose.hpp
{
    size_t node_id;
    Quat orientation;
    Vec3 position;
};

static inline bool finite3(const Vec3 &v)
{
    return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z);
}
static inline bool finiteq(const Quat &q)
{
    return std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z) && std::isfinite(q.w);
}
static inline bool hasnanq(const Quat &q)
{
    return std::isnan(q.x) || std::isnan(q.y) || std::isnan(q.z) || std::isnan(q.w);
}

// Eigen Quaternion::toRotationMatrix()
static Mat3 quat_to_matrix(const Quat &q)
{
    Mat3 R;
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R.m[0][0] = 1 - (tyy + tzz);
    R.m[0][1] = txy - twz;
    R.m[0][2] = txz + twy;
    R.m[1][0] = txy + twz;
    R.m[1][1] = 1 - (txx + tzz);
    R.m[1][2] = tyz - twx;
    R.m[2][0] = txz - twy;
    R.m[2][1] = tyz + twx;
    R.m[2][2] = 1 - (txx + tyy);
    return R;
}
static Vec3 quat_rotate_d(const Quat &q, const Vec3 &v)
{
    const double qq[4] = {q.x, q.y, q.z, q.w};
    const V3<double> r = quat_rotate<double>(qq, V3<double>{v.x, v.y, v.z});
    return Vec3{r.x, r.y, r.z};
}

// src/geometry/intersection.cpp:116-143
static std::pair<Vec3, double> rayIntersection(const Vec3 &d1, const Vec3 &o1, const Vec3 &d2, const Vec3 &o2)
{
    Vec3 res{NAN, NAN, NAN};
    double error = NAN;
    const double n1dn1 = dot(d1, d1), n1dn2 = dot(d1, d2), n2dn2 = dot(d2, d2);
    const double scale_denom = n1dn1 * n2dn2 - n1dn2 * n1dn2;
    if (std::abs(scale_denom) > 1e-9)
    {
        const Vec3 offset = o1 - o2;
        const double offsetdn1 = dot(offset, d1), offsetdn2 = dot(offset, d2);
        const double t = (n1dn2 * offsetdn2 - n2dn2 * offsetdn1) / scale_denom;
        const double s = (n1dn1 * offsetdn2 - n1dn2 * offsetdn1) / scale_denom;
        const Vec3 p1 = o1 + d1 * t, p2 = o2 + d2 * s;
        res = (p1 + p2) * 0.5;
        const Vec3 dd = p1 - p2;
        error = dot(dd, dd) * (t >= 0 && s >= 0 ? 1 : -1);
    }
    return {res, error};
}

// include/opencalibration/relax/grid_filter.hpp (T = const feature_match_denormalized*)
class GridFilter
{
  public:
    void setResolution(double r)
    {
        if (_map.empty())
            _res = r;
    }
    void addMeasurement(double x, double y, double score, const feature_match_denormalized *value)
    {
        const uint64_t index =
            (static_cast<uint64_t>((int)std::floor(x / _res)) << 32) | static_cast<uint32_t>((int)std::floor(y / _res));
        auto it = _map.find(index);
        if (it == _map.end())
        {
            _map.emplace(index, std::make_pair(score, value));
            _best.insert(value);
        }
        else if (it->second.first < score)
        {
            _best.erase(it->second.second);
            it->second = std::make_pair(score, value);
            _best.insert(value);
        }
    }
    const std::unordered_set<const feature_match_denormalized *> &best() const
    {
        return _best;
    }

  private:
    double _res = 0.075;
    std::unordered_map<uint64_t, std::pair<double, const feature_match_denormalized *>> _map;
    std::unordered_set<const feature_match_denormalized *> _best;
};

struct relax_summary
{
    int solves = 0;
    int iterations_total = 0; // sum over solves of summary.iterations.size()
    int last_iterations = 0;
    double last_initial_cost = 0, last_final_cost = 0;
    int last_residual_blocks = 0;
};

class RelaxProblem
{
  public:
    RelaxProblem(const std::vector<relax_node> &nodes, const std::vector<relax_edge> &edges) : _nodes(nodes), _edges(edges)
    {
        _opt.max_num_iterations = 100; // relax_problem.cpp:30-37
        _opt.initial_trust_region_radius = 1;
    }

    void setupGroundPlaneProblem(std::vector<NodePose> &poses, const std::vector<size_t> &edges_to_optimize)
    {
        // initialize (:146-161)
        for (NodePose &n : poses)
            _nodes_to_optimize.emplace_back(n.node_id, &n);
        initializeGroundPlane();
        _loss.reset(new mc::HuberLoss(1 * M_PI / 180));
        gridFilterMatchesPerImage(edges_to_optimize, 0.15);
        for (size_t e : edges_to_optimize)
            if (!_edges_used.count(e))
                addRayTriangleMeasurementCost(e);
        addDownwardsPrior();
    }

    void relaxObservedModelOnly(relax_summary *sum) // :931-984
    {
        std::vector<double *> params = _problem.GetParameterBlocks();
        std::vector<std::pair<double *, bool>> backup;
        for (double *p : params)
        {
            backup.emplace_back(p, _problem.IsParameterBlockConstant(p));
            _problem.SetParameterBlockConstant(p);
        }
        for (int i = 0; i < 3; i++)
            for (auto &b : backup)
                if (b.first == &_mesh_z[i] && !b.second)
                    _problem.SetParameterBlockVariable(&_mesh_z[i]);
        solve(sum);
        for (auto &b : backup)
        {
            if (b.second)
                _problem.SetParameterBlockConstant(b.first);
            else
                _problem.SetParameterBlockVariable(b.first);
        }
    }

    void solve(relax_summary *sum) // :1390-1420
    {
        if (_problem.NumParameterBlocks() == 0 || _problem.NumResidualBlocks() == 0)
            return;
        mc::SolverSummary s;
        mc::Solve(_opt, &_problem, &s);
        if (sum)
        {
            sum->solves++;
            sum->iterations_total += (int)s.iterations.size();
            sum->last_iterations = (int)s.iterations.size();
            sum->last_initial_cost = s.initial_cost;
            sum->last_final_cost = s.final_cost;
            sum->last_residual_blocks = _problem.NumResidualBlocks();
        }
        for (auto &p : _nodes_to_optimize)
        {
            Quat &q = p.second->orientation; // Eigen normalize(): coeffs /= norm()
            const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
            q.x /= n;
            q.y /= n;
            q.z /= n;
            q.w /= n;
        }
    }

    double _mesh_z[3] = {NAN, NAN, NAN};
    double _mesh_xy[3][2];

  private:
    struct PoseOpt
    {
        bool optimize = false;
        const Vec3 *loc_ptr = nullptr;
        Quat *rot_ptr = nullptr;
    };
    PoseOpt nodeid2poseopt(size_t node_id) // :182-232
    {
        PoseOpt po;
        for (auto &p : _nodes_to_optimize)
            if (p.first == node_id)
            {
                po.optimize = true;
                po.loc_ptr = &p.second->position;
                po.rot_ptr = &p.second->orientation;
                return po;
            }
        const relax_node &n = _nodes[node_id];
        if (finiteq(n.orientation) && finite3(n.position))
        {
            po.loc_ptr = &n.position;
            po.rot_ptr = const_cast<Quat *>(&n.orientation);
        }
        return po;
    }

    void initializeGroundPlane() // :1189-1242
    {
        double xmin = 1e12, ymin = 1e12, xmax = -1e12, ymax = -1e12, height = 0;
        for (auto &p : _nodes_to_optimize)
        {
            const Vec3 &loc = p.second->position;
            xmin = std::min(xmin, loc.x);
            ymin = std::min(ymin, loc.y);
            xmax = std::max(xmax, loc.x);
            ymax = std::max(ymax, loc.y);
            height += loc.z;
        }
        height /= (double)_nodes_to_optimize.size();
        const double margin = 50;
        height -= margin;
        const double cx = (xmin + xmax) / 2, cy = (ymin + ymax) / 2;
        const double spacing = std::max(xmax - xmin, ymax - ymin) + margin;
        const double c[3][2] = {{-spacing + cx, -spacing + cy}, {spacing + cx, -spacing + cy}, {0 + cx, spacing + cy}};
        for (int i = 0; i < 3; i++)
        {
            _mesh_xy[i][0] = c[i][0];
            _mesh_xy[i][1] = c[i][1];
            _mesh_z[i] = height;
        }
        // MeshIntersectionSearcher::init starts from the first edge (node0, node1, opposite node2)
        _tri[0] = 0;
        _tri[1] = 1;
        _tri[2] = 2;
    }

    void gridFilterMatchesPerImage(const std::vector<size_t> &edges_to_optimize, double frac) // :234-309
    {
        for (size_t edge_id : edges_to_optimize)
        {
            const relax_edge &edge = _edges[edge_id];
            PoseOpt src = nodeid2poseopt(edge.source), dst = nodeid2poseopt(edge.dest);
            if (src.loc_ptr == nullptr || dst.loc_ptr == nullptr)
                return; // sic: `return`, not `continue`
            const camera_model &sm = _nodes[edge.source].model, &dm = _nodes[edge.dest].model;
            const Mat3 srot = quat_to_matrix(*src.rot_ptr), drot = quat_to_matrix(*dst.rot_ptr);
            GridFilter &sf = _grid_filter[edge.source][edge_id], &df = _grid_filter[edge.dest][edge_id];
            sf.setResolution(frac);
            df.setResolution(frac);
            std::vector<std::pair<double, size_t>> scored;
            scored.reserve(edge.inlier_matches.size());
            for (size_t idx = 0; idx < edge.inlier_matches.size(); idx++)
            {
                const auto &inl = edge.inlier_matches[idx];
                const Vec3 sdir = mul(srot, image_to_3d(inl.pixel_1, sm)), ddir = mul(drot, image_to_3d(inl.pixel_2, dm));
                const auto isect = rayIntersection(sdir, *src.loc_ptr, ddir, *dst.loc_ptr);
                const double intersection_score = isect.second < 0 ? 0. : 1. / (1. + isect.second);
                const double cos_angle = dot(sdir, ddir);
                const double angle_score = 1.0 - cos_angle * cos_angle;
                const double descriptor_score =
                    inl.match_index < edge.match_distance.size() ? 1.0 - edge.match_distance[inl.match_index] : 1.0;
                const double snx = (inl.pixel_1[0] - sm.principle_point[0]) / sm.focal_length_pixels;
                const double sny = (inl.pixel_1[1] - sm.principle_point[1]) / sm.focal_length_pixels;
                const double dnx = (inl.pixel_2[0] - dm.principle_point[0]) / dm.focal_length_pixels;
                const double dny = (inl.pixel_2[1] - dm.principle_point[1]) / dm.focal_length_pixels;
                double ransac_score = 1.0;
                if (edge.is_homography)
                {
                    const Vec2 h = hnormalized(mul(edge.ransac_relation, Vec3{snx, sny, 1.0}));
                    const double ex = dnx - h.x, ey = dny - h.y;
                    ransac_score = 1.0 / (1.0 + std::sqrt(ex * ex + ey * ey));
                }
                scored.emplace_back(intersection_score * angle_score * descriptor_score * ransac_score, idx);
            }
            std::sort(scored.begin(), scored.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
            for (const auto &[score, idx] : scored)
                if (score > 0)
                {
                    const auto &inl = edge.inlier_matches[idx];
                    sf.addMeasurement(inl.pixel_1[0] / sm.pixels_cols, inl.pixel_1[1] / sm.pixels_rows, score, &inl);
                    df.addMeasurement(inl.pixel_2[0] / dm.pixels_cols, inl.pixel_2[1] / dm.pixels_rows, score, &inl);
                }
        }
    }

    // MeshIntersectionSearcher::triangleIntersect (intersect.cpp:56-163) on the one-triangle mesh:
    // every edge is a border, so the walk ends at its first step either way.
    bool triangleIntersectVertical(double px, double py)
    {
        auto anticlockwise = [](const double a[2], const double b[2], const double c[2]) {
            return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]) < 0;
        };
        if (anticlockwise(_mesh_xy[_tri[0]], _mesh_xy[_tri[1]], _mesh_xy[_tri[2]]))
            std::swap(_tri[0], _tri[1]);
        // ray (0,0,-1) through (px,py): the plane intersection keeps x,y
        const double P[2] = {px, py};
        for (int i = 0; i < 3; i++)
            if (anticlockwise(P, _mesh_xy[_tri[i]], _mesh_xy[_tri[(i + 1) % 3]]))
                return false; // OUTSIDE_BORDER
        return true;
    }

    void addRayTriangleMeasurementCost(size_t edge_id) // :388-560, fixed intrinsics
    {
        const relax_edge &edge = _edges[edge_id];
        PoseOpt src = nodeid2poseopt(edge.source), dst = nodeid2poseopt(edge.dest);
        if (src.loc_ptr == nullptr || dst.loc_ptr == nullptr)
            return;
        const camera_model &sm = _nodes[edge.source].model, &dm = _nodes[edge.dest].model;
        const auto &swl = _grid_filter[edge.source][edge_id].best(), &dwl = _grid_filter[edge.dest][edge_id].best();
        double *datas[2] = {&src.rot_ptr->x, &dst.rot_ptr->x};
        bool points_added = false;
        for (const auto &inl : edge.inlier_matches)
        {
            if (swl.find(&inl) == swl.end() && dwl.find(&inl) == dwl.end())
                continue;
            const Vec3 sray = image_to_3d(inl.pixel_1, sm), dray = image_to_3d(inl.pixel_2, dm);
            const auto isect = rayIntersection(quat_rotate_d(*src.rot_ptr, sray), *src.loc_ptr,
                                               quat_rotate_d(*dst.rot_ptr, dray), *dst.loc_ptr);
            if (std::isnan(isect.first.x) || std::isnan(isect.first.y))
                continue; // RAY_PARALLEL_TO_PLANE via the NaN check of intersect.cpp:85-90
            if (!triangleIntersectVertical(isect.first.x, isect.first.y))
                continue;
            auto *f = new PlaneIntersectionAngleCost();
            const Vec3 locs[2] = {*src.loc_ptr, *dst.loc_ptr}, rays[2] = {sray, dray};
            for (int i = 0; i < 2; i++)
            {
                f->camera_loc[i][0] = locs[i].x, f->camera_loc[i][1] = locs[i].y, f->camera_loc[i][2] = locs[i].z;
                f->camera_ray[i][0] = rays[i].x, f->camera_ray[i][1] = rays[i].y, f->camera_ray[i][2] = rays[i].z;
            }
            for (int i = 0; i < 3; i++)
            {
                f->plane_point[i][0] = _mesh_xy[_tri[i]][0];
                f->plane_point[i][1] = _mesh_xy[_tri[i]][1];
            }
            _problem.AddResidualBlock(new mc::AutoDiffCostFunction<PlaneIntersectionAngleCost, 6, 4, 4, 1, 1, 1>(f),
                                      _loss.get(),
                                      {datas[0], datas[1], &_mesh_z[_tri[0]], &_mesh_z[_tri[1]], &_mesh_z[_tri[2]]});
            points_added = true;
        }
        if (points_added)
        {
            _problem.SetManifold(datas[0], mc::Manifold::EIGEN_QUATERNION);
            _problem.SetManifold(datas[1], mc::Manifold::EIGEN_QUATERNION);
            if (!src.optimize)
                _problem.SetParameterBlockConstant(datas[0]);
            if (!dst.optimize)
                _problem.SetParameterBlockConstant(datas[1]);
        }
        _edges_used.insert(edge_id);
    }

    void addDownwardsPrior() // :1290-1301
    {
        for (auto &p : _nodes_to_optimize)
            if (!hasnanq(p.second->orientation))
            {
                double *d = &p.second->orientation.x;
                _problem.AddResidualBlock(new mc::AutoDiffCostFunction<PointsDownwardsPrior, 1, 4>(new PointsDownwardsPrior(1e-3)),
                                          nullptr, {d});
                _problem.SetManifold(d, mc::Manifold::EIGEN_QUATERNION);
            }
    }

    const std::vector<relax_node> &_nodes;
    const std::vector<relax_edge> &_edges;
    std::vector<std::pair<size_t, NodePose *>> _nodes_to_optimize; // insertion order, like unordered_dense
    std::map<size_t, std::map<size_t, GridFilter>> _grid_filter;
    std::unordered_set<size_t> _edges_used;
    std::unique_ptr<mc::LossFunction> _loss;
    mc::Problem _problem;
    mc::SolverOptions _opt;
    int _tri[3] = {0, 1, 2};
};

// src/relax/relax.cpp:44-87
static void runGroundPlane(const std::vector<relax_node> &graph_nodes, const std::vector<relax_edge> &edges,
                           std::vector<NodePose> &nodes, const std::vector<size_t> &edges_to_optimize,
                           relax_summary *sum, double plane_out[9])
{
    // DOWN_ORIENTED_NORTH = Quaterniond(AngleAxisd(M_PI, UnitX)): w = cos(pi/2), xyz = sin(pi/2) * axis
    Quat previous{std::sin(M_PI / 2), 0.0, 0.0, std::cos(M_PI / 2)};
    for (auto &node : nodes)
    {
        if (hasnanq(node.orientation))
        {
            node.orientation = previous;
            if (graph_nodes.size() > 2 * nodes.size())
            {
                std::vector<NodePose> justThis{node};
                RelaxProblem rp(graph_nodes, edges);
                rp.setupGroundPlaneProblem(justThis, edges_to_optimize);
                rp.relaxObservedModelOnly(sum);
                rp.solve(sum);
                node = justThis[0];
            }
            else
            {
                RelaxProblem rp(graph_nodes, edges);
                rp.setupGroundPlaneProblem(nodes, edges_to_optimize);
                rp.relaxObservedModelOnly(sum);
                rp.solve(sum);
            }
        }
        previous = node.orientation;
    }
    RelaxProblem rp(graph_nodes, edges);
    rp.setupGroundPlaneProblem(nodes, edges_to_optimize);
    rp.relaxObservedModelOnly(sum);
    rp.solve(sum);
    for (int i = 0; i < 3; i++)
    {
        plane_out[3 * i] = rp._mesh_xy[i][0];
        plane_out[3 * i + 1] = rp._mesh_xy[i][1];
        plane_out[3 * i + 2] = rp._mesh_z[i];
    }
}

} // namespace oracle

using namespace oracle;

extern "C"
{

// Flat driver for relax(graph, nodes, cam_models, edges, {ORIENTATION, GROUND_PLANE}).
//  graph: n_nodes x {pos3, ori4 (xyzw, may be NaN)}, one shared camera model (model10)
//  poses: n_poses node indices + orientations (in/out; NaN = uninitialised), positions from the graph
//  edges: src/dst node index, H (9), inlier offsets, per inlier {px1 xy, px2 xy}, match_index, and the
//         per-edge match distance list (offsets + values; may be empty)
//  edges_to_optimize: edge indices in whitelist order
//  summary_out: {solves, iterations_total, last_iterations, last_initial_cost, last_final_cost, last_residual_blocks}
void oc_relax_ground_plane(size_t n_nodes, const double *node_pos, const double *node_ori, const double *model10,
                           size_t n_poses, const uint64_t *pose_node, double *pose_ori, size_t n_edges,
                           const uint64_t *edge_src, const uint64_t *edge_dst, const double *edge_H,
                           const uint8_t *edge_is_homography, const uint64_t *inl_off, const double *inl_px, const uint64_t *inl_match_index,
                           const uint64_t *dist_off, const double *dist, size_t n_opt_edges,
                           const uint64_t *opt_edges, double *plane_out, double *summary_out)
{
    camera_model cm;
    cm.focal_length_pixels = model10[0];
    cm.principle_point[0] = model10[1];
    cm.principle_point[1] = model10[2];
    for (int i = 0; i < 3; i++)
        cm.radial_distortion[i] = model10[3 + i];
    cm.tangential_distortion[0] = model10[6];
    cm.tangential_distortion[1] = model10[7];
    cm.pixels_cols = (size_t)model10[8];
    cm.pixels_rows = (size_t)model10[9];
    std::vector<relax_node> nodes(n_nodes);
    for (size_t i = 0; i < n_nodes; i++)
    {
        nodes[i].position = Vec3{node_pos[3 * i], node_pos[3 * i + 1], node_pos[3 * i + 2]};
        nodes[i].orientation = Quat{node_ori[4 * i], node_ori[4 * i + 1], node_ori[4 * i + 2], node_ori[4 * i + 3]};
        nodes[i].model = cm;
    }
    std::vector<relax_edge> edges(n_edges);
    for (size_t e = 0; e < n_edges; e++)
    {
        edges[e].source = edge_src[e];
        edges[e].dest = edge_dst[e];
        std::memcpy(edges[e].ransac_relation.m, edge_H + 9 * e, 72);
        edges[e].is_homography = edge_is_homography ? edge_is_homography[e] != 0 : true;
        for (uint64_t k = inl_off[e]; k < inl_off[e + 1]; k++)
        {
            feature_match_denormalized f;
            f.pixel_1[0] = inl_px[4 * k], f.pixel_1[1] = inl_px[4 * k + 1];
            f.pixel_2[0] = inl_px[4 * k + 2], f.pixel_2[1] = inl_px[4 * k + 3];
            f.feature_index_1 = f.feature_index_2 = 0;
            f.match_index = inl_match_index[k];
            edges[e].inlier_matches.push_back(f);
        }
        if (dist_off)
            edges[e].match_distance.assign(dist + dist_off[e], dist + dist_off[e + 1]);
    }
    std::vector<NodePose> poses(n_poses);
    for (size_t i = 0; i < n_poses; i++)
    {
        poses[i].node_id = pose_node[i];
        poses[i].orientation = Quat{pose_ori[4 * i], pose_ori[4 * i + 1], pose_ori[4 * i + 2], pose_ori[4 * i + 3]};
        poses[i].position = nodes[pose_node[i]].position;
    }
    std::vector<size_t> opt(opt_edges, opt_edges + n_opt_edges);
    relax_summary sum;
    double plane[9];
    runGroundPlane(nodes, edges, poses, opt, &sum, plane);
    for (size_t i = 0; i < n_poses; i++)
    {
        pose_ori[4 * i] = poses[i].orientation.x;
        pose_ori[4 * i + 1] = poses[i].orientation.y;
        pose_ori[4 * i + 2] = poses[i].orientation.z;
        pose_ori[4 * i + 3] = poses[i].orientation.w;
    }
    if (plane_out)
        std::memcpy(plane_out, plane, sizeof plane);
    if (summary_out)
    {
        summary_out[0] = sum.solves;
        summary_out[1] = sum.iterations_total;
        summary_out[2] = sum.last_iterations;
        summary_out[3] = sum.last_initial_cost;
        summary_out[4] = sum.last_final_cost;
        summary_out[5] = sum.last_residual_blocks;
    }
}

// cost functor known answers (test/test_relax.cpp:169-188, :1052-1096)
double oc_points_downwards_prior(const double *q, double weight)
{
    PointsDownwardsPrior p(weight);
    double r = NAN;
    p(q, &r);
    return r;
}
void oc_robust_centroid(const double *pts, int n, double thr, double *out)
{
    V3<double> p[ROBUST_CENTROID_MAX_POINTS];
    for (int i = 0; i < n; i++)
        p[i] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    const V3<double> c = robustCentroid<double>(p, n, thr);
    out[0] = c.x, out[1] = c.y, out[2] = c.z;
}
// residuals (6) and ambient Jacobian (6 x 11, row-major: q0 4 | q1 4 | z 3) of one 2-ray block
int oc_plane_intersection_cost(const double *locs6, const double *rays6, const double *plane_xy6, const double *q0,
                               const double *q1, const double *z3, double *residuals, double *jac)
{
    auto *f = new PlaneIntersectionAngleCost();
    std::memcpy(f->camera_loc, locs6, 48);
    std::memcpy(f->camera_ray, rays6, 48);
    std::memcpy(f->plane_point, plane_xy6, 48);
    mc::AutoDiffCostFunction<PlaneIntersectionAngleCost, 6, 4, 4, 1, 1, 1> cf(f);
    const double *params[5] = {q0, q1, z3, z3 + 1, z3 + 2};
    double j0[24], j1[24], j2[6], j3[6], j4[6];
    double *jacs[5] = {j0, j1, j2, j3, j4};
    const bool ok = cf.Evaluate(params, residuals, jac ? jacs : nullptr);
    if (jac)
        for (int r = 0; r < 6; r++)
        {
            for (int c = 0; c < 4; c++)
            {
                jac[r * 11 + c] = j0[r * 4 + c];
                jac[r * 11 + 4 + c] = j1[r * 4 + c];
            }
            jac[r * 11 + 8] = j2[r];
            jac[r * 11 + 9] = j3[r];
            jac[r * 11 + 10] = j4[r];
        }
    return ok ? 1 : 0;
}

} // extern "C"
