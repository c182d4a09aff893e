#include "../env.hpp"
#include "relax.hpp"

#include "relax_util.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <memory>
#include <unordered_map>

namespace opencalibration_amd
{

namespace
{
using namespace relax_detail;

std::atomic<int> g_setup_check{0}, g_setup_checked{0};

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
        const bool verbose = ochip_verbose("relax");
        auto tmark = clk::now();
        auto lap = [&](const char *what) {
            if (verbose)
                fprintf(stderr, "[relax setup] %-28s %.3f ms\n", what, since(tmark) * 1e3);
            tmark = clk::now();
        };
        _poses = &poses;
        // camera table: optimised poses first (their order), context cameras appended on first use.  A node that appears
        // twice in the pose list (RelaxGroup::init appends the first ring of context cameras once per round,
        // relax_group.cpp:40-66) is optimised through its FIRST pose only: _nodes_to_optimize.emplace keeps the first
        // (relax_problem.cpp:150-154); the other pose is never touched.
        _pose_cam.assign(poses.size(), UINT32_MAX);
        for (size_t i = 0; i < poses.size(); i++)
            if (_opt_index.emplace(poses[i].node_id, i).second)
            {
                _pose_cam[i] = (uint32_t)_cam_opt.size();
                _cam_of_node.emplace(poses[i].node_id, _pose_cam[i]);
                push_camera(poses[i].position, poses[i].orientation, true);
            }
        initialize_plane();

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
        // gridFilterMatchesPerImage (:234-309) and addRayTriangleMeasurementCost (:388-560, fixed intrinsics) on the
        // device: ochip_plane_setup_* (csrc/relax_setup.hip).  The searcher's orientation fix-up of the single triangle
        // happens on its first use and is the same for every edge.  Edges past the first one without poses get no
        // filter pass, hence no whitelist and no blocks.
        fix_triangle_orientation();
        double tri[6];
        for (int i = 0; i < 3; i++)
        {
            tri[2 * i] = _xy[_tri[i]][0];
            tri[2 * i + 1] = _xy[_tri[i]][1];
        }
        std::vector<const MeasurementGraph::Edge *> edges(n_filter, nullptr);
        for (size_t k = 0; k < n_filter; k++)
            edges[k] = _graph.getEdge(edges_to_optimize[k]);
        device_filter df;
        if (!df.run(_ctx, _graph, edges, src, dst, n_filter, _cam_pos, _cam_q, tri, 0.15, error))
            return false;
        ochip_plane_setup *const ps = df.handle;
        const std::vector<ochip_plane_edge> &pe = df.pe;
        const std::vector<const MeasurementGraph::Edge *> &pe_edge = df.edge;
        const std::vector<pose_ref> &pe_src = df.src, &pe_dst = df.dst;
        lap("grid filter (device)");
        uint64_t total_blocks = 0;
        if (ochip_plane_setup_blocks(ps, nullptr, nullptr, nullptr, 0, &total_blocks) != OCHIP_OK)
            return fail(error, "ochip_plane_setup_blocks");
        _blk_a.resize(total_blocks);
        _blk_b.resize(total_blocks);
        _blk_rays.resize(total_blocks * 6);
        if (ochip_plane_setup_blocks(ps, _blk_a.data(), _blk_b.data(), _blk_rays.data(), total_blocks, &total_blocks) != OCHIP_OK)
            return fail(error, "ochip_plane_setup_blocks");
        lap("edge blocks (device)");
        if (g_setup_check.load())
        {
            // the same two passes with the host code (relax_util.hpp), compared bit for bit
            edge_blocks all;
            for (size_t j = 0; j < pe.size(); j++)
                add_edge_blocks(*pe_edge[j], pe_src[j], pe_dst[j], grid_filter(_graph, *pe_edge[j], pe_src[j], pe_dst[j], 0.15), all);
            if (all.a != _blk_a || all.b != _blk_b || all.rays.size() != _blk_rays.size() ||
                std::memcmp(all.rays.data(), _blk_rays.data(), all.rays.size() * sizeof(double)) != 0)
            {
                if (error)
                    *error = "relax set-up: the device's residual blocks differ from the host's (" + std::to_string(_blk_a.size()) + " vs " +
                         std::to_string(all.a.size()) + ")";
                return false;
            }
            g_setup_checked.fetch_add(1);
        }
        // addDownwardsPrior (:1290-1301)
        for (size_t i = 0; i < poses.size(); i++)
            if (_pose_cam[i] != UINT32_MAX && !hasnan4(poses[i].orientation))
                _prior_cam.push_back(_pose_cam[i]);

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
            if (error)
                *error = std::string("ochip_relax_problem_create: ") + ochip_last_error(_ctx);
            return false;
        }
        if (shard && (shard->world > 1 || shard->exchange) &&
            ochip_relax_set_shard(_dev, shard->rank, shard->world, shard->exchange, shard->user) != OCHIP_OK)
        {
            if (error)
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
            if (_pose_cam[i] == UINT32_MAX)
                continue;
            double *o4 = (*_poses)[i].orientation;
            const double *s4 = &q[4 * (size_t)_pose_cam[i]];
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
        if (error)
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

    void initialize_plane() // initializeGroundPlane (:1189-1242)
    {
        double lo[2] = {1e12, 1e12}, hi[2] = {-1e12, -1e12}, height = 0;
        size_t n_unique = 0;
        for (size_t i = 0; i < _poses->size(); i++)
        {
            if (_pose_cam[i] == UINT32_MAX)
                continue;
            const NodePose &p = (*_poses)[i];
            n_unique++;
            for (int a = 0; a < 2; a++)
            {
                lo[a] = std::min(lo[a], p.position[a]);
                hi[a] = std::max(hi[a], p.position[a]);
            }
            height += p.position[2];
        }
        height /= (double)n_unique;
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
    std::vector<uint32_t> _blk_a, _blk_b, _prior_cam, _pose_cam;
    double _xy[3][2], _z[3];
    int _tri[3] = {0, 1, 2};
    ochip_relax_problem *_dev = nullptr;
};

} // namespace

int relax_setup_check(int on)
{
    if (on >= 0)
        g_setup_check.store(on);
    return g_setup_checked.load();
}

bool relax_setup_check_on()
{
    return g_setup_check.load() != 0;
}
void relax_setup_check_passed()
{
    g_setup_checked.fetch_add(1);
}

namespace
{

// initializeGroundPlane (:1189-1242) + the searcher's orientation fix-up of the single triangle, for a list of camera positions:
// the corners in the searcher's order (corner_of[i]: which corner of the surface model stands at place i) and the start height
void bootstrap_plane(const std::vector<const double *> &positions, double tri_xy[6], double *z0, int corner_of[3] = nullptr)
{
    double lo[2] = {1e12, 1e12}, hi[2] = {-1e12, -1e12}, height = 0;
    for (const double *p : positions)
    {
        for (int a = 0; a < 2; a++)
        {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
        height += p[2];
    }
    height /= (double)positions.size();
    const double margin = 50;
    height -= margin;
    const double cx = (lo[0] + hi[0]) / 2, cy = (lo[1] + hi[1]) / 2;
    const double spacing = std::max(hi[0] - lo[0], hi[1] - lo[1]) + margin;
    const double c[3][2] = {{-spacing + cx, -spacing + cy}, {spacing + cx, -spacing + cy}, {0 + cx, spacing + cy}};
    int tri[3] = {0, 1, 2};
    if ((c[1][0] - c[0][0]) * (c[2][1] - c[0][1]) - (c[1][1] - c[0][1]) * (c[2][0] - c[0][0]) < 0) // anticlockwise(a, b, c)
        std::swap(tri[0], tri[1]);
    for (int i = 0; i < 3; i++)
    {
        tri_xy[2 * i] = c[tri[i]][0];
        tri_xy[2 * i + 1] = c[tri[i]][1];
        if (corner_of)
            corner_of[i] = tri[i];
    }
    *z0 = height;
}

// runGroundPlane (src/relax/relax.cpp:44-87) as one resident launch on the device (ochip_plane_chain_*, csrc/relax_chain.hip):
// the loop over the poses without an orientation - each takes the orientation of the pose in front of it and is relaxed,
// on its own against the oriented cameras of the graph or together with the group - and the group's own solve behind it.
// Returns 1: the poses up to *resume_pose are done (== nodes.size(): all of them; *all_done: the group's solve too, `surface`
// is written), 0: not taken (something the chain does not do: the caller's loop runs from the first pose), -1: error.
int bootstrap_on_device(ochip_ctx *ctx, const MeasurementGraph &graph, std::vector<NodePose> &nodes,
                        const std::vector<size_t> &edges_to_optimize, surface_model_plane *surface, RelaxTimers *timers,
                        std::string *error, size_t *resume_pose, bool *all_done)
{
    const auto t_begin = clk::now();
    *all_done = false;
    const bool just_this = graph.size_nodes() > 2 * nodes.size(); // relax.cpp:61
    // poses: the first pose of a node is the one that is optimised (relax_problem.cpp:150-154)
    std::unordered_map<size_t, size_t> first_pose;
    std::vector<size_t> stepped;
    for (size_t i = 0; i < nodes.size(); i++)
    {
        const bool first = first_pose.emplace(nodes[i].node_id, i).second;
        if (hasnan4(nodes[i].orientation))
        {
            if (!first)
                return 0; // a second pose of a node without an orientation: the host loop knows what that means
            stepped.push_back(i);
        }
    }
    if (stepped.empty())
        return 0;
    auto graph_finite = [&](size_t node_id) -> const MeasurementGraph::Node * {
        const MeasurementGraph::Node *n = graph.getNode(node_id);
        return n != nullptr && finite4(n->payload.orientation) && finite3(n->payload.position) ? n : nullptr;
    };
    // cameras: the first pose of every node (the group's solve optimises them all), then the graph's oriented cameras the
    // edges reach.  While one camera is relaxed on its own the others stand as the GRAPH has them (nodeid2poseopt, :182-232):
    // that is what the pose list holds too - RelaxGroup::init copies it - and where it is not, the host loop runs
    std::vector<double> cam_pos, cam_q;
    std::vector<uint8_t> cam_opt, cam_stepped;
    std::unordered_map<size_t, uint32_t> cam_of_node;
    std::vector<uint32_t> pose_cam(nodes.size(), UINT32_MAX);
    auto push_cam = [&](size_t node_id, const double *pos, const double *q, bool optimise, bool is_stepped) {
        const uint32_t c = (uint32_t)cam_opt.size();
        cam_pos.insert(cam_pos.end(), pos, pos + 3);
        cam_q.insert(cam_q.end(), q, q + 4);
        cam_opt.push_back(optimise ? 1 : 0);
        cam_stepped.push_back(is_stepped ? 1 : 0);
        cam_of_node.emplace(node_id, c);
        return c;
    };
    std::vector<char> is_stepped(nodes.size(), 0);
    for (size_t i : stepped)
        is_stepped[i] = 1;
    for (size_t i = 0; i < nodes.size(); i++)
    {
        if (first_pose.at(nodes[i].node_id) != i)
            continue;
        if (just_this)
        {
            const MeasurementGraph::Node *n = graph.getNode(nodes[i].node_id);
            if (n == nullptr)
                return 0;
            if (is_stepped[i] ? graph_finite(nodes[i].node_id) != nullptr
                              : (std::memcmp(n->payload.orientation, nodes[i].orientation, sizeof nodes[i].orientation) != 0 ||
                                 std::memcmp(n->payload.position, nodes[i].position, sizeof nodes[i].position) != 0))
                return 0;
        }
        pose_cam[i] = push_cam(nodes[i].node_id, nodes[i].position, nodes[i].orientation, true, is_stepped[i] != 0);
    }
    if (cam_opt.size() > 330)
        return 0; // (the chain's dense system ends at 1 023 unknowns)
    auto cam_of = [&](size_t node_id) -> uint32_t {
        auto it = cam_of_node.find(node_id);
        if (it != cam_of_node.end())
            return it->second;
        const MeasurementGraph::Node *n = graph_finite(node_id);
        return n ? push_cam(node_id, n->payload.position, n->payload.orientation, false, false) : UINT32_MAX;
    };
    // the edges in the whitelist's order; gridFilterMatchesPerImage stops at the first one without usable poses
    std::unordered_map<const CameraModel *, uint32_t> model_index;
    std::vector<double> models10;
    auto model_of = [&](const CameraModel *m) {
        auto it = model_index.find(m);
        if (it != model_index.end())
            return it->second;
        const double row[10] = {m->focal_length_pixels,  m->principle_point[0],   m->principle_point[1],       m->radial_distortion[0],
                                m->radial_distortion[1], m->radial_distortion[2], m->tangential_distortion[0], m->tangential_distortion[1],
                                (double)m->pixels_cols,  (double)m->pixels_rows};
        models10.insert(models10.end(), row, row + 10);
        return model_index.emplace(m, (uint32_t)model_index.size()).first->second;
    };
    std::vector<ochip_plane_edge> pe;
    std::vector<const MeasurementGraph::Edge *> pe_edge;
    uint64_t n_inliers = 0;
    // one camera at a time: a camera without an orientation in the graph is usable only as the step's own camera.  idx1: the
    // first edge with such a camera (u1), idx2: the first edge with another one - or with two of them
    size_t idx1 = SIZE_MAX, idx2 = SIZE_MAX;
    uint32_t u1 = UINT32_MAX;
    for (size_t k = 0; k < edges_to_optimize.size(); k++)
    {
        const MeasurementGraph::Edge *e = graph.getEdge(edges_to_optimize[k]);
        if (e == nullptr)
            continue;
        if (e->source == e->dest)
            return 0;
        const uint32_t ca = cam_of(e->source), cb = cam_of(e->dest);
        if (ca == UINT32_MAX || cb == UINT32_MAX)
            break; // never usable: nothing gets past it
        if (idx2 == SIZE_MAX)
        {
            const bool sa = cam_stepped[ca] != 0, sb = cam_stepped[cb] != 0;
            if (sa && sb)
                idx2 = pe.size(), idx1 = std::min(idx1, idx2);
            else if (sa || sb)
            {
                const uint32_t u = sa ? ca : cb;
                if (idx1 == SIZE_MAX)
                    idx1 = pe.size(), u1 = u;
                else if (u != u1)
                    idx2 = pe.size();
            }
        }
        const camera_relations &rel = e->payload;
        ochip_plane_edge r{};
        r.cam_a = ca;
        r.cam_b = cb;
        r.model_a = model_of(graph.getNode(e->source)->payload.model.get());
        r.model_b = model_of(graph.getNode(e->dest)->payload.model.get());
        r.n_inliers = (uint32_t)rel.inlier_matches.size();
        r.flags = rel.relationType == camera_relations::RelationType::HOMOGRAPHY ? 1u : 0u;
        r.inlier_offset = n_inliers;
        std::memcpy(r.H, rel.ransac_relation, sizeof r.H);
        n_inliers += r.n_inliers;
        pe.push_back(r);
        pe_edge.push_back(e);
    }
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
        return -1;
    }
    ochip_plane_inlier *const rec = static_cast<ochip_plane_inlier *>(inl.p);
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t j = 0; j < pe.size(); j++)
    {
        const camera_relations &rel = pe_edge[j]->payload;
        ochip_plane_inlier *o = rec + pe[j].inlier_offset;
        for (size_t idx = 0; idx < rel.inlier_matches.size(); idx++)
        {
            const feature_match_denormalized &m = rel.inlier_matches[idx];
            o[idx].px1[0] = m.pixel_1[0], o[idx].px1[1] = m.pixel_1[1];
            o[idx].px2[0] = m.pixel_2[0], o[idx].px2[1] = m.pixel_2[1];
            o[idx].descriptor_score = m.match_index < rel.matches.size() ? 1.0 - rel.matches[m.match_index].distance : 1.0;
        }
    }
    // the steps: the poses without an orientation in the list's order, then the group's own solve
    std::vector<ochip_plane_chain_step> steps;
    steps.reserve(stepped.size() + 1);
    steps.resize(stepped.size());
    double tri_all[6], z_all = 0;
    int corner_of[3] = {0, 1, 2};
    {
        std::vector<const double *> positions;
        for (size_t i = 0; i < nodes.size(); i++)
            if (pose_cam[i] != UINT32_MAX)
                positions.push_back(nodes[i].position);
        bootstrap_plane(positions, tri_all, &z_all, corner_of);
    }
    const double down[4] = {std::sin(M_PI / 2), 0.0, 0.0, std::cos(M_PI / 2)}; // DOWN_ORIENTED_NORTH, relax.cpp:12
    for (size_t k = 0; k < stepped.size(); k++)
    {
        const size_t i = stepped[k];
        ochip_plane_chain_step &st = steps[k];
        st.cam = pose_cam[i];
        st.mode = just_this ? 0u : 1u;
        st.prev_cam = -1;
        if (i == 0)
            std::memcpy(st.prev_q, down, sizeof down);
        else
        {
            // previous = the pose in front, as it is when this step starts: a camera of the chain's state where the chain moves
            // it (a stepped camera; with the group every first pose), else what the pose list holds
            const bool moving = pose_cam[i - 1] != UINT32_MAX && (just_this ? is_stepped[i - 1] != 0 : true);
            if (moving)
                st.prev_cam = (int32_t)pose_cam[i - 1];
            std::memcpy(st.prev_q, nodes[i - 1].orientation, sizeof st.prev_q);
        }
        if (just_this)
        {
            bootstrap_plane({nodes[i].position}, st.tri_xy, &st.z0);
            const size_t own = pose_cam[i] == u1 ? idx2 : idx1;
            st.n_filter = (uint32_t)std::min(pe.size(), own);
        }
        else
        {
            std::memcpy(st.tri_xy, tri_all, sizeof tri_all);
            st.z0 = z_all;
            st.n_filter = (uint32_t)pe.size();
        }
    }
    {
        ochip_plane_chain_step st{}; // runGroundPlane's last problem (relax.cpp:81-84)
        st.cam = 0;
        st.mode = 2u;
        st.prev_cam = -1;
        std::memcpy(st.prev_q, down, sizeof down);
        std::memcpy(st.tri_xy, tri_all, sizeof tri_all);
        st.z0 = z_all;
        st.n_filter = (uint32_t)pe.size();
        steps.push_back(st);
    }
    // OCHIP_TEST_HOOKS=chain_partial: the chain takes the first half of the poses only and leaves the rest - and the group's own
    // solve - to the caller's loop, as it does when a step needs the host's grid filter or a wait on the device runs out
    const size_t n_planned = steps.size(); // every pose without an orientation, then the group
    if (ochip_test_hook("chain_partial") && stepped.size() >= 2)
        steps.resize(stepped.size() / 2);
    ochip_plane_chain *chain = nullptr;
    const int crc = ochip_plane_chain_create(ctx, pe.data(), (uint32_t)pe.size(), rec, n_inliers, cam_pos.data(), cam_q.data(), cam_opt.data(),
                                             (uint32_t)cam_opt.size(), models10.data(), (uint32_t)model_index.size(), 0.15, 1 * M_PI / 180, 1e-3,
                                             steps.data(), (uint32_t)steps.size(), &chain);
    if (crc == OCHIP_EINVAL)
        return 0; // (not for the chain: the host loop)
    if (crc != OCHIP_OK)
    {
        if (error)
            *error = std::string("ochip_plane_chain_create: ") + ochip_last_error(ctx);
        return -1;
    }
    if (timers)
        timers->setup_host += since(t_begin);
    const auto t_run = clk::now();
    std::vector<double> q_out(cam_q.size());
    ochip_plane_chain_result res{};
    const int rrc = ochip_plane_chain_run(chain, ochip_test_hook("chain_stepped") ? 1 : 0, q_out.data(), &res);
    ochip_plane_chain_destroy(chain);
    if (rrc != OCHIP_OK)
    {
        if (error)
            *error = std::string("ochip_plane_chain_run: ") + ochip_last_error(ctx);
        return -1;
    }
    if (ochip_verbose("relax"))
        fprintf(stderr, "[relax chain] %zu poses without an orientation (%s) + the group, %zu edges, %llu inlier matches: %u steps done (status %d), "
                        "%d solves, %d iterations, %d grid syncs on %d workgroups, %.3f ms\n",
                stepped.size(), just_this ? "one at a time" : "with the group", pe.size(), (unsigned long long)n_inliers, res.steps_done, res.status,
                res.solves, res.iterations_total, res.grid_syncs, res.workgroups, since(t_run) * 1e3);
    if (timers)
    {
        timers->device += since(t_run);
        timers->solves += res.solves;
        timers->iterations_total += res.iterations_total;
        timers->last_iterations = res.last_iterations;
        timers->last_initial_cost = res.last_initial_cost;
        timers->last_final_cost = res.last_final_cost;
        timers->last_residual_blocks = res.last_residual_blocks;
    }
    // what relax() wrote back after every finished step
    const bool everything = res.steps_done >= n_planned;
    if (just_this && !everything)
    {
        for (size_t k = 0; k < res.steps_done && k < stepped.size(); k++)
            std::memcpy(nodes[stepped[k]].orientation, &q_out[4 * (size_t)pose_cam[stepped[k]]], 4 * sizeof(double));
    }
    else if (res.steps_done > 0)
        for (size_t i = 0; i < nodes.size(); i++)
            if (pose_cam[i] != UINT32_MAX)
                std::memcpy(nodes[i].orientation, &q_out[4 * (size_t)pose_cam[i]], 4 * sizeof(double));
    *resume_pose = res.steps_done >= stepped.size() ? nodes.size() : stepped[res.steps_done];
    if (everything)
    {
        *all_done = true;
        if (surface)
            for (int i = 0; i < 3; i++)
            {
                surface->corner[corner_of[i]][0] = tri_all[2 * i];
                surface->corner[corner_of[i]][1] = tri_all[2 * i + 1];
                surface->corner[corner_of[i]][2] = res.plane_z[i];
            }
    }
    return 1;
}

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
    // the poses without an orientation: one resident launch walks them on the device (csrc/relax_chain.hip); what it does not
    // take - and OCHIP_TEST_HOOKS=host_bootstrap - goes through the loop below, a problem and two solves per camera
    size_t first = 0;
    const bool sharded = shard && (shard->world > 1 || shard->exchange);
    if (!sharded && !ochip_test_hook("host_bootstrap"))
    {
        size_t resume = 0;
        bool all_done = false;
        const int brc = bootstrap_on_device(ctx, graph, nodes, edges_to_optimize, surface, timers, error, &resume, &all_done);
        if (brc < 0)
            return false;
        if (all_done)
            return true;
        if (brc > 0)
        {
            first = resume;
            if (first > 0 && first <= nodes.size())
                std::memcpy(previous, nodes[first - 1].orientation, sizeof previous);
        }
    }
    for (size_t i = first; i < nodes.size(); i++)
    {
        NodePose &node = nodes[i];
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
