// Host side of the two relax flavours only the reference's tests reach (test/test_relax.cpp:298-682), on the device:
//   runPoints (src/relax/relax.cpp:103-115): setup3dPointProblem (src/relax/relax_problem.cpp:122-145) - grid filter at 0.05,
//     one 3-D point per whitelisted inlier triangulated by rayIntersection + TinySolver (src/geometry/intersection.cpp:163-186),
//     two reprojection blocks per point (addPointMeasurementsCost, :986-1187) - then relaxObservedModelOnly and the joint
//     solve, both on ochip_relaxp_* (csrc/relax_points.hip: points eliminated by their 3 x 3 Schur complements);
//   runRelativeOrientation (relax.cpp:14-42): setupDecompositionProblem (:40-59) - one MultiDecomposedRotationCost block per
//     edge from the homography decompositions, downward prior - with the NaN-orientation bootstrap, on the general engine
//     (ochip_relaxg_*: relation blocks).
#include "relax_mesh.hpp"

#include "invert_distortion.hpp"
#include "relax_util.hpp"
#include "tiny_solver.hpp"

namespace opencalibration_amd
{

namespace
{
using namespace relax_detail;

// value + 3 partials (the arithmetic of ceres::Jet<double, 3>), for the triangulation's Jacobian
struct J3
{
    double a, v[3];
    J3(double s = 0) : a(s), v{0, 0, 0}
    {
    }
};
inline J3 operator+(const J3 &f, const J3 &g)
{
    J3 h(f.a + g.a);
    for (int i = 0; i < 3; i++)
        h.v[i] = f.v[i] + g.v[i];
    return h;
}
inline J3 operator-(const J3 &f, const J3 &g)
{
    J3 h(f.a - g.a);
    for (int i = 0; i < 3; i++)
        h.v[i] = f.v[i] - g.v[i];
    return h;
}
inline J3 operator*(const J3 &f, const J3 &g)
{
    J3 h(f.a * g.a);
    for (int i = 0; i < 3; i++)
        h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
    return h;
}
inline J3 operator/(const J3 &f, const J3 &g)
{
    const double ginv = 1.0 / g.a, fg = f.a * ginv;
    J3 h(fg);
    for (int i = 0; i < 3; i++)
        h.v[i] = (f.v[i] - fg * g.v[i]) * ginv;
    return h;
}
inline double val(double x)
{
    return x;
}
inline double val(const J3 &x)
{
    return x.a;
}

// image_from_3d(point, model, camera_location, camera_orientation) (distort_keypoints.hpp:26-88) on T = double or J3
template <typename T> void project_point(const T X[3], const CameraModel &m, const double *loc, const double *q, T px[2])
{
    // camera_orientation.inverse() * (point - location): Eigen's inverse = conjugate / squared norm
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    double qi[4] = {0, 0, 0, 0};
    if (n2 > 0)
        qi[0] = -q[0] / n2, qi[1] = -q[1] / n2, qi[2] = -q[2] / n2, qi[3] = q[3] / n2;
    const T d[3] = {X[0] - T(loc[0]), X[1] - T(loc[1]), X[2] - T(loc[2])};
    // QuaternionBase::_transformVector: v + w * uv + qv x uv, uv = 2 (qv x v)
    T uv[3] = {T(qi[1]) * d[2] - T(qi[2]) * d[1], T(qi[2]) * d[0] - T(qi[0]) * d[2], T(qi[0]) * d[1] - T(qi[1]) * d[0]};
    for (int i = 0; i < 3; i++)
        uv[i] = uv[i] + uv[i];
    const T cr[3] = {T(qi[1]) * uv[2] - T(qi[2]) * uv[1], T(qi[2]) * uv[0] - T(qi[0]) * uv[2], T(qi[0]) * uv[1] - T(qi[1]) * uv[0]};
    T ray[3];
    for (int i = 0; i < 3; i++)
        ray[i] = d[i] + uv[i] * T(qi[3]) + cr[i];
    const T cz = val(ray[2]) < 1e-3 ? T(1e-3) : ray[2];
    const T u[2] = {ray[0] / cz, ray[1] / cz};
    T r2[3];
    r2[0] = u[0] * u[0] + u[1] * u[1];
    r2[1] = r2[0] * r2[0];
    r2[2] = r2[1] * r2[0];
    const T radial_dot = T(m.radial_distortion[0]) * r2[0] + T(m.radial_distortion[1]) * r2[1] + T(m.radial_distortion[2]) * r2[2];
    const T prod = u[0] * u[1];
    for (int i = 0; i < 2; i++)
    {
        const T dd = (T(1.0) + radial_dot) * u[i] + T(2.0) * prod * T(m.tangential_distortion[i]) +
                     T(m.tangential_distortion[1 - i]) * (r2[0] + T(2.0) * u[i] * u[i]);
        px[i] = dd * T(m.focal_length_pixels) + T(m.principle_point[i]);
    }
}

// rayIntersection(model1, model2, pos1, pos2, rot1, rot2, px1, px2) (intersection.cpp:163-186)
void ray_intersection_refined(const CameraModel &m1, const CameraModel &m2, const double *pos1, const double *pos2, const double *rot1,
                              const double *rot2, const double *px1, const double *px2, double point[3], double *error)
{
    double r1[3], r2[3];
    image_to_3d(px1, m1, r1);
    image_to_3d(px2, m2, r2);
    v3 mid;
    double gap;
    ray_intersection(rotate(rot1, v3{r1[0], r1[1], r1[2]}), v3{pos1[0], pos1[1], pos1[2]}, rotate(rot2, v3{r2[0], r2[1], r2[2]}),
                     v3{pos2[0], pos2[1], pos2[2]}, &mid, &gap);
    point[0] = mid.x, point[1] = mid.y, point[2] = mid.z;
    const CameraModel *models[2] = {&m1, &m2};
    const double *pos[2] = {pos1, pos2}, *rot[2] = {rot1, rot2}, *px[2] = {px1, px2};
    auto eval = [&](const double *x, double *res, double *jac) {
        for (int c = 0; c < 2; c++)
        {
            if (jac)
            {
                J3 X[3], p[2];
                for (int i = 0; i < 3; i++)
                {
                    X[i] = J3(x[i]);
                    X[i].v[i] = 1;
                }
                project_point<J3>(X, *models[c], pos[c], rot[c], p);
                for (int i = 0; i < 2; i++)
                {
                    res[2 * c + i] = p[i].a - px[c][i];
                    for (int k = 0; k < 3; k++)
                        jac[(size_t)(2 * c + i) * 3 + k] = p[i].v[k];
                }
            }
            else
            {
                double p[2];
                project_point<double>(x, *models[c], pos[c], rot[c], p);
                res[2 * c] = p[0] - px[c][0];
                res[2 * c + 1] = p[1] - px[c][1];
            }
        }
    };
    tiny_solver_options o;
    o.max_num_iterations = 50;
    o.cost_threshold = 1e-7;
    o.parameter_tolerance = 1e-14;
    o.gradient_tolerance = 1e-12;
    o.initial_trust_region_radius = 1e6;
    *error = tiny_solver_n<3>(eval, 4, point, o);
}

class PointsProblem
{
  public:
    PointsProblem(ochip_ctx *ctx, const MeasurementGraph &graph) : _ctx(ctx), _graph(graph)
    {
    }
    ~PointsProblem()
    {
        if (_dev)
            ochip_relaxp_problem_destroy(_dev);
    }

    // setup3dPointProblem (relax_problem.cpp:122-145)
    bool setup(std::vector<NodePose> &poses, std::vector<std::pair<size_t, CameraModel>> &cam_models,
               const std::vector<size_t> &edges_to_optimize, uint32_t options, std::string *error)
    {
        _poses = &poses;
        _cam_models = &cam_models;
        _options = options;
        _pose_cam.assign(poses.size(), UINT32_MAX);
        for (size_t i = 0; i < poses.size(); i++)
            if (_opt_index.emplace(poses[i].node_id, i).second)
            {
                _pose_cam[i] = (uint32_t)_cam_opt.size();
                _cam_of_node.emplace(poses[i].node_id, _pose_cam[i]);
                push_camera(poses[i].position, poses[i].orientation, true);
            }
        // which functor (:1038-1086)
        auto has_any = [&](uint32_t o) { return (options & o) != 0; };
        auto has_all = [&](uint32_t o) { return (options & o) == o; };
        int functor = -1;
        if (has_any(OPT_LENS_DISTORTIONS_TANGENTIAL) && has_all(OPT_LENS_DISTORTIONS_RADIAL | OPT_FOCAL_LENGTH | OPT_ORIENTATION | OPT_POINTS_3D))
            functor = 3;
        else if (has_any(OPT_LENS_DISTORTIONS_RADIAL) && has_all(OPT_FOCAL_LENGTH | OPT_ORIENTATION | OPT_POINTS_3D))
            functor = 2;
        else if (has_any(OPT_FOCAL_LENGTH | OPT_PRINCIPAL_POINT) && has_all(OPT_ORIENTATION | OPT_POINTS_3D))
            functor = 1;
        else if (has_all(OPT_ORIENTATION | OPT_POINTS_3D))
            functor = 0;
        _functor = functor;
        // gridFilterMatchesPerImage (:234-309) at 0.05: an edge without usable poses stops the whole pass
        const size_t ne = edges_to_optimize.size();
        size_t n_filter = ne;
        std::vector<pose_ref> src(ne), dst(ne);
        for (size_t k = 0; k < ne; k++)
        {
            const MeasurementGraph::Edge *e = _graph.getEdge(edges_to_optimize[k]);
            if (e == nullptr)
                continue;
            src[k] = lookup(e->source);
            dst[k] = lookup(e->dest);
            if ((src[k].loc == nullptr || dst[k].loc == nullptr) && n_filter == ne)
                n_filter = k;
        }
        std::vector<std::vector<uint8_t>> keep(ne);
#pragma omp parallel for schedule(dynamic, 1)
        for (size_t k = 0; k < n_filter; k++)
        {
            const MeasurementGraph::Edge *e = _graph.getEdge(edges_to_optimize[k]);
            if (e != nullptr)
                keep[k] = grid_filter(_graph, *e, src[k], dst[k], 0.05);
        }
        // addPointMeasurementsCost per edge, in whitelist order (an edge listed twice is used once)
        std::unordered_map<size_t, char> used;
        _grp_first.push_back(0);
        for (size_t k = 0; k < ne; k++)
        {
            const MeasurementGraph::Edge *e = _graph.getEdge(edges_to_optimize[k]);
            if (e == nullptr || used.count(edges_to_optimize[k]))
                continue;
            if (!add_edge(*e, src[k], dst[k], keep[k], error))
                return false;
            if (src[k].loc != nullptr && dst[k].loc != nullptr && _functor >= 0)
                used.emplace(edges_to_optimize[k], 1);
        }
        ochip_relaxp_desc d{};
        d.n_cams = (uint32_t)_cam_opt.size();
        d.cam_pos = _cam_pos.data();
        d.cam_q = _cam_q.data();
        d.cam_optimize = _cam_opt.data();
        d.n_points = (uint32_t)(_points.size() / 3);
        d.point_xyz = _points.data();
        d.n_groups = (uint32_t)(_grp_first.size() - 1);
        d.grp_first = _grp_first.data();
        d.grp_cam = _grp_cam.data();
        d.obs_px = _obs_px.data();
        d.functor = std::max(_functor, 0);
        const CameraModel *m = _shared_model ? _shared_model : nullptr;
        if (m)
        {
            d.model[0] = m->focal_length_pixels;
            d.model[1] = m->principle_point[0], d.model[2] = m->principle_point[1];
            for (int i = 0; i < 3; i++)
                d.model[3 + i] = m->radial_distortion[i];
            d.model[6] = m->tangential_distortion[0], d.model[7] = m->tangential_distortion[1];
        }
        else
            d.model[0] = 1.0;
        d.opt_focal = has_any(OPT_FOCAL_LENGTH);
        d.opt_principal = has_any(OPT_PRINCIPAL_POINT);
        d.n_radial_free = 3; // BROWN246 or no parameterisation chosen: a free block (:1160-1180)
        if (has_all(OPT_LENS_DISTORTIONS_RADIAL) && !has_all(OPT_LENS_DISTORTIONS_RADIAL_BROWN246_PARAMETERIZATION))
        {
            if (has_all(OPT_LENS_DISTORTIONS_RADIAL_BROWN24_PARAMETERIZATION))
                d.n_radial_free = 2;
            else if (has_all(OPT_LENS_DISTORTIONS_RADIAL_BROWN2_PARAMETERIZATION))
                d.n_radial_free = 1;
        }
        d.focal_lo = 100.0;
        d.focal_hi = 20000.0;
        d.huber_a = 10.0; // HuberLoss(10), :128
        d.mono_observations = (uint32_t)_mono_count;
        d.mono_r_max = _mono_r_max;
        if (ochip_relaxp_problem_create(_ctx, &d, &_dev) != OCHIP_OK)
        {
            *error = std::string("ochip_relaxp_problem_create: ") + ochip_last_error(_ctx);
            return false;
        }
        return true;
    }

    bool relax_observed_model_only(RelaxTimers *t, std::string *error)
    {
        if (ochip_relaxp_set_structure_only(_dev, 1) != OCHIP_OK)
            return fail(error, "ochip_relaxp_set_structure_only");
        const bool ok = solve(t, error);
        if (ochip_relaxp_set_structure_only(_dev, 0) != OCHIP_OK)
            return fail(error, "ochip_relaxp_set_structure_only");
        return ok;
    }

    bool solve(RelaxTimers *t, std::string *error)
    {
        if (_points.empty())
            return true; // NumResidualBlocks() == 0 (:1398-1402)
        ochip_relax_options o{1000, 1.0, 1e-6, 1e-10, 1e-8}; // max_num_iterations = 1000 (:143)
        ochip_relax_summary s{};
        if (ochip_relaxp_solve(_dev, &o, &s) != OCHIP_OK)
            return fail(error, "ochip_relaxp_solve");
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
        double model[8];
        if (ochip_relaxp_get_state(_dev, q.data(), _points.data(), model) != OCHIP_OK)
            return fail(error, "ochip_relaxp_get_state");
        for (size_t i = 0; i < _poses->size(); i++) // orientation.normalize(), :1410-1413
        {
            if (_pose_cam[i] == UINT32_MAX)
                continue;
            double *o4 = (*_poses)[i].orientation;
            const double *s4 = &q[4 * (size_t)_pose_cam[i]];
            const double n = std::sqrt(s4[0] * s4[0] + s4[1] * s4[1] + s4[2] * s4[2] + s4[3] * s4[3]);
            for (int k = 0; k < 4; k++)
                o4[k] = s4[k] / n;
        }
        if (_shared_model && _functor >= 1)
        {
            _shared_model->focal_length_pixels = model[0];
            _shared_model->principle_point[0] = model[1], _shared_model->principle_point[1] = model[2];
            for (int i = 0; i < 3; i++)
                _shared_model->radial_distortion[i] = model[3 + i];
            _shared_model->tangential_distortion[0] = model[6], _shared_model->tangential_distortion[1] = model[7];
        }
        return true;
    }

    const std::vector<double> &points() const // the tracks' points, edge by edge (TestRelaxProblem::test_get_tracks)
    {
        return _points;
    }
    // getSurfaceModel: the points as the surface's cloud (relax_problem.cpp:1422-1437)
    void surface(surface_model *out) const
    {
        *out = surface_model();
        point_cloud c(_points.size() / 3);
        for (size_t i = 0; i < c.size(); i++)
            c[i] = {_points[3 * i], _points[3 * i + 1], _points[3 * i + 2]};
        out->cloud.push_back(std::move(c));
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
    pose_ref lookup(size_t node_id) // nodeid2poseopt (:182-232)
    {
        pose_ref po;
        auto it = _opt_index.find(node_id);
        if (it != _opt_index.end())
        {
            NodePose &np = (*_poses)[it->second];
            po.optimize = true;
            po.loc = np.position;
            po.rot = np.orientation;
            po.cam = _pose_cam[it->second];
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
    CameraModel *model_of(size_t node_id) // the group's copy of the node's model when it holds one, else the node's own
    {
        const MeasurementGraph::Node *node = _graph.getNode(node_id);
        if (node == nullptr || !node->payload.model)
            return nullptr;
        for (auto &m : *_cam_models)
            if (m.first == node->payload.model->id)
                return &m.second;
        return node->payload.model.get();
    }

    // addPointMeasurementsCost (:986-1187)
    bool add_edge(const MeasurementGraph::Edge &edge, const pose_ref &s, const pose_ref &d, const std::vector<uint8_t> &keep,
                  std::string *error)
    {
        if (s.loc == nullptr || d.loc == nullptr)
            return true;
        CameraModel *sm = model_of(edge.source), *dm = model_of(edge.dest);
        if (sm == nullptr || dm == nullptr)
            return true;
        if (_functor < 0)
            return true; // "No viable bundle options found"
        if (sm != dm || (_shared_model && _shared_model != sm))
        {
            *error = "relax (3-D points): the device path optimises one shared camera model; this problem holds several";
            return false;
        }
        _shared_model = sm;
        const camera_relations &rel = edge.payload;
        const size_t first = _points.size() / 3;
        for (size_t idx = 0; idx < rel.inlier_matches.size(); idx++)
        {
            if (idx >= keep.size() || keep[idx] == 0)
                continue;
            const feature_match_denormalized &m = rel.inlier_matches[idx];
            double X[3], err;
            ray_intersection_refined(*sm, *dm, s.loc, d.loc, s.rot, d.rot, m.pixel_1, m.pixel_2, X, &err);
            // both blocks must evaluate to finite residuals (:1088-1103)
            double p1[2], p2[2];
            project_point<double>(X, *sm, s.loc, s.rot, p1);
            project_point<double>(X, *dm, d.loc, d.rot, p2);
            const double r[4] = {p1[0] - m.pixel_1[0], p1[1] - m.pixel_1[1], p2[0] - m.pixel_2[0], p2[1] - m.pixel_2[1]};
            if (!std::isfinite(r[0]) || !std::isfinite(r[1]) || !std::isfinite(r[2]) || !std::isfinite(r[3]))
            {
                // (the reference keeps the track's point but adds no residual block for it; such a point takes no part in
                // the solve, so it is left out of the device problem)
                continue;
            }
            _points.insert(_points.end(), X, X + 3);
            _obs_px.insert(_obs_px.end(), m.pixel_1, m.pixel_1 + 2);
            _obs_px.insert(_obs_px.end(), m.pixel_2, m.pixel_2 + 2);
            if (_options & OPT_LENS_DISTORTIONS_RADIAL) // trackRadialObservation for both blocks (:1113-1120,1368-1379)
            {
                if (_mono_count == 0)
                {
                    const double hc = sm->pixels_cols / 2.0, hr = sm->pixels_rows / 2.0;
                    _mono_r_max = std::sqrt(hc * hc + hr * hr) / sm->focal_length_pixels;
                }
                _mono_count += 2;
            }
        }
        if (_points.size() / 3 > first)
        {
            _grp_cam.push_back(s.cam);
            _grp_cam.push_back(d.cam);
            _grp_first.push_back((uint32_t)(_points.size() / 3));
        }
        return true;
    }

    ochip_ctx *_ctx;
    const MeasurementGraph &_graph;
    std::vector<NodePose> *_poses = nullptr;
    std::vector<std::pair<size_t, CameraModel>> *_cam_models = nullptr;
    uint32_t _options = 0;
    int _functor = 0;
    CameraModel *_shared_model = nullptr;
    std::unordered_map<size_t, size_t> _opt_index;
    std::unordered_map<size_t, uint32_t> _cam_of_node;
    std::vector<double> _cam_pos, _cam_q, _points, _obs_px;
    std::vector<uint8_t> _cam_opt;
    std::vector<uint32_t> _pose_cam, _grp_first, _grp_cam;
    size_t _mono_count = 0;
    double _mono_r_max = 0;
    ochip_relaxp_problem *_dev = nullptr;
};

} // namespace

bool relax_points(ochip_ctx *ctx, const MeasurementGraph &graph, std::vector<NodePose> &nodes,
                  std::vector<std::pair<size_t, CameraModel>> &cam_models, const std::vector<size_t> &edges_to_optimize,
                  uint32_t options, surface_model *surface, RelaxTimers *timers, std::string *error, int mode,
                  std::vector<double> *points_before, std::vector<double> *points_after)
{
    auto t0 = clk::now();
    PointsProblem rp(ctx, graph);
    std::vector<NodePose> backup = nodes;
    std::vector<std::pair<size_t, CameraModel>> models_backup = cam_models;
    if (!rp.setup(nodes, cam_models, edges_to_optimize, options, error))
        return false;
    if (timers)
        timers->setup_host += since(t0);
    if (points_before)
        *points_before = rp.points();
    t0 = clk::now();
    bool ok = true;
    if (mode < 0) // runPoints: relaxObservedModelOnly, then solve
        ok = rp.relax_observed_model_only(timers, error) && rp.solve(timers, error);
    else if (mode == 1)
        ok = rp.solve(timers, error);
    else if (mode == 2)
        ok = rp.relax_observed_model_only(timers, error);
    if (timers)
        timers->device += since(t0);
    if (!ok)
    {
        nodes = backup;
        cam_models = models_backup;
        return false;
    }
    if (points_after)
        *points_after = rp.points();
    if (surface)
        rp.surface(surface);
    return true;
}

} // namespace opencalibration_amd
