#include "../env.hpp"
#include "relax_mesh.hpp"
#include "triangle_walker.hpp"

#include "invert_distortion.hpp"
#include "relax_util.hpp"

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <limits>
#include <map>
#include <memory>
#include <numeric>

namespace opencalibration_amd
{

namespace
{
using namespace relax_detail;

// ---- exact nearest neighbours in the plane over a bucket grid (what expand_mesh.cpp asks of jk::tree::KDTree<double, 2>:
//      the payload of the nearest point, the squared distance to the second nearest) --------------------------------
class PointGrid2D
{
  public:
    void add(double x, double y, double payload)
    {
        _x.push_back(x);
        _y.push_back(y);
        _payload.push_back(payload);
        _built = false;
    }
    size_t size() const
    {
        return _x.size();
    }
    // nearest point (ties: first inserted): its payload
    double nearest_payload(double qx, double qy)
    {
        double d[2];
        size_t idx[2];
        k_nearest(qx, qy, 1, d, idx);
        return _payload[idx[0]];
    }
    // squared distance to the k-th nearest, k in {1, 2} (clamped to the number of points)
    double kth_distance(double qx, double qy, int k)
    {
        double d[2];
        size_t idx[2];
        const int got = k_nearest(qx, qy, k, d, idx);
        return d[got - 1];
    }

  private:
    void build()
    {
        _lo[0] = _lo[1] = std::numeric_limits<double>::max();
        double hi[2] = {-_lo[0], -_lo[1]};
        for (size_t i = 0; i < _x.size(); i++)
        {
            _lo[0] = std::min(_lo[0], _x[i]);
            _lo[1] = std::min(_lo[1], _y[i]);
            hi[0] = std::max(hi[0], _x[i]);
            hi[1] = std::max(hi[1], _y[i]);
        }
        const double span = std::max(std::max(hi[0] - _lo[0], hi[1] - _lo[1]), 1e-9);
        _n = std::max(1, std::min(1024, (int)std::sqrt((double)_x.size())));
        _cell = span / _n * (1 + 1e-12);
        _start.assign((size_t)_n * _n + 1, 0);
        auto cell_of = [&](size_t i) { return (size_t)cx(_x[i]) * _n + cy(_y[i]); };
        for (size_t i = 0; i < _x.size(); i++)
            _start[cell_of(i) + 1]++;
        for (size_t c = 0; c < (size_t)_n * _n; c++)
            _start[c + 1] += _start[c];
        _items.resize(_x.size());
        std::vector<uint32_t> fill(_start.begin(), _start.end() - 1);
        for (size_t i = 0; i < _x.size(); i++) // ascending i inside a cell
            _items[fill[cell_of(i)]++] = (uint32_t)i;
        _built = true;
    }
    int cx(double x) const
    {
        return std::max(0, std::min(_n - 1, (int)std::floor((x - _lo[0]) / _cell)));
    }
    int cy(double y) const
    {
        return std::max(0, std::min(_n - 1, (int)std::floor((y - _lo[1]) / _cell)));
    }
    int k_nearest(double qx, double qy, int k, double *dist, size_t *idx)
    {
        if (!_built)
            build();
        k = std::min<int>(k, (int)_x.size());
        int have = 0;
        const int c0 = cx(qx), c1 = cy(qy);
        // distance of the query to the grid's cell (c0, c1) when it lies outside of the grid
        for (int ring = 0;; ring++)
        {
            const int x0 = c0 - ring, x1 = c0 + ring, y0 = c1 - ring, y1 = c1 + ring;
            for (int ix = std::max(0, x0); ix <= std::min(_n - 1, x1); ix++)
                for (int iy = std::max(0, y0); iy <= std::min(_n - 1, y1); iy++)
                {
                    if (ix != x0 && ix != x1 && iy != y0 && iy != y1)
                        continue; // interior of the ring was visited before
                    for (uint32_t e = _start[(size_t)ix * _n + iy]; e < _start[(size_t)ix * _n + iy + 1]; e++)
                    {
                        const size_t i = _items[e];
                        const double dx = _x[i] - qx, dy = _y[i] - qy;
                        const double d = dx * dx + dy * dy;
                        // insert into the sorted (d, i) list of the k best
                        int pos = have;
                        while (pos > 0 && (d < dist[pos - 1] || (d == dist[pos - 1] && i < idx[pos - 1])))
                            pos--;
                        if (pos < k)
                        {
                            for (int m = std::min(have, k - 1); m > pos; m--)
                            {
                                dist[m] = dist[m - 1];
                                idx[m] = idx[m - 1];
                            }
                            dist[pos] = d;
                            idx[pos] = i;
                            have = std::min(have + 1, k);
                        }
                    }
                }
            // everything outside of the visited square is at least this far away
            const double bx = std::min(qx - (_lo[0] + x0 * _cell), (_lo[0] + (x1 + 1) * _cell) - qx);
            const double by = std::min(qy - (_lo[1] + y0 * _cell), (_lo[1] + (y1 + 1) * _cell) - qy);
            const double border = std::min(bx, by);
            const bool covers_all = x0 <= 0 && y0 <= 0 && x1 >= _n - 1 && y1 >= _n - 1;
            if (covers_all || (have == k && border > 0 && border * border > dist[k - 1]))
                break;
        }
        return have;
    }
    std::vector<double> _x, _y, _payload;
    std::vector<uint32_t> _start, _items;
    double _lo[2] = {0, 0}, _cell = 1;
    int _n = 1;
    bool _built = false;
};

struct mesh_inputs // what rebuildMesh and buildMinimalMesh both derive first (expand_mesh.cpp:35-121, :252-303)
{
    PointGrid2D vertices, cameras;
    double cam_min[2], cam_max[2];
    std::vector<double> heights, nearest2;
};
void prepare_mesh_inputs(const point_cloud &cams, const std::vector<surface_model> &previous, mesh_inputs &m)
{
    m.cam_min[0] = m.cam_min[1] = std::numeric_limits<double>::max();
    m.cam_max[0] = m.cam_max[1] = -std::numeric_limits<double>::max();
    for (const surface_model &s : previous)
    {
        for (const MeshNode &n : s.mesh.nodes)
            m.vertices.add(n.location[0], n.location[1], n.location[2]);
        for (const point_cloud &c : s.cloud)
            for (const auto &p : c)
                m.vertices.add(p[0], p[1], p[2]);
    }
    for (const auto &p : cams)
    {
        for (int a = 0; a < 2; a++)
        {
            m.cam_min[a] = std::min(m.cam_min[a], p[a]);
            m.cam_max[a] = std::max(m.cam_max[a], p[a]);
        }
        m.cameras.add(p[0], p[1], p[2]);
        if (m.vertices.size() > 0)
        {
            const double agl = p[2] - m.vertices.nearest_payload(p[0], p[1]);
            if (agl > -500 && agl < 5000)
                m.heights.push_back(agl);
        }
    }
    for (const auto &p : cams)
        m.nearest2.push_back(m.cameras.kth_distance(p[0], p[1], 2));
    std::sort(m.nearest2.begin(), m.nearest2.end());
}
double median_height(mesh_inputs &m, double grid_distance)
{
    if (m.heights.empty())
        m.heights.push_back(std::isfinite(grid_distance) ? grid_distance : 10.0);
    std::sort(m.heights.begin(), m.heights.end());
    return m.heights[m.heights.size() / 2];
}

} // namespace

MeshGraph rebuildMesh(const point_cloud &cams, const std::vector<surface_model> &previous)
{
    bool has_previous = false;
    for (const surface_model &s : previous)
        has_previous |= s.mesh.size_nodes() > 0 || !s.cloud.empty();
    if (cams.size() < 2 && !has_previous)
        return MeshGraph();
    mesh_inputs m;
    prepare_mesh_inputs(cams, previous, m);
    double grid = m.nearest2.size() < 2 ? std::numeric_limits<double>::infinity() : std::sqrt(m.nearest2[m.nearest2.size() / 2]);
    const double ex = m.cam_max[0] - m.cam_min[0], ey = m.cam_max[1] - m.cam_min[1];
    const double min_grid = std::sqrt(ex * ex + ey * ey) / 1000.0;
    if (grid < min_grid)
        grid = std::max(1e-3, min_grid);
    const double med = median_height(m, grid);
    const double border = std::max(0.0, std::min(1000.0, med * 2));
    size_t rows = static_cast<size_t>(std::ceil(std::max(0., ey + 2 * border) / grid)) + 1;
    size_t cols = static_cast<size_t>(std::ceil(std::max(0., ex + 2 * border) / grid)) + 1;
    rows = std::min<size_t>(rows, 1000);
    cols = std::min<size_t>(cols, 1000);
    MeshGraph mesh;
    std::vector<size_t> id(rows * cols);
    auto at = [&](size_t r, size_t c) -> size_t & { return id[r * cols + c]; };
    // vertices column by column; every new vertex is tied to its lower, left and lower-left neighbours
    for (size_t c = 0; c < cols; c++)
    {
        const double x = m.cam_min[0] - border + grid * c;
        for (size_t r = 0; r < rows; r++)
        {
            const double y = m.cam_min[1] - border + grid * r;
            const double z = m.vertices.size() > 0 ? m.vertices.nearest_payload(x, y) : m.cameras.nearest_payload(x, y) - med;
            const size_t v = mesh.addNode(x, y, z);
            at(r, c) = v;
            MeshEdge e;
            if (r > 0)
            {
                e.border = c == 0 || c + 1 == cols;
                mesh.addEdge(e, v, at(r - 1, c));
            }
            if (c > 0)
            {
                e.border = r == 0 || r + 1 == rows;
                mesh.addEdge(e, v, at(r, c - 1));
            }
            if (r > 0 && c > 0)
            {
                e.border = false;
                mesh.addEdge(e, v, at(r - 1, c - 1));
            }
        }
    }
    // the vertices opposite every edge (a border edge keeps its single triangle's vertex in slot 0)
    for (size_t c = 0; c < cols; c++)
        for (size_t r = 0; r < rows; r++)
        {
            if (r > 0)
            {
                MeshEdge *e = mesh.getEdge(at(r, c), at(r - 1, c));
                if (c > 0)
                    e->triangleOppositeNodes[0] = at(r - 1, c - 1);
                if (c + 1 < cols)
                {
                    e->triangleOppositeNodes[1] = at(r, c + 1);
                    if (e->border)
                        std::swap(e->triangleOppositeNodes[0], e->triangleOppositeNodes[1]);
                }
            }
            if (c > 0)
            {
                MeshEdge *e = mesh.getEdge(at(r, c), at(r, c - 1));
                if (r > 0)
                    e->triangleOppositeNodes[0] = at(r - 1, c - 1);
                if (r + 1 < rows)
                {
                    e->triangleOppositeNodes[1] = at(r + 1, c);
                    if (e->border)
                        std::swap(e->triangleOppositeNodes[0], e->triangleOppositeNodes[1]);
                }
            }
            if (r > 0 && c > 0)
            {
                MeshEdge *e = mesh.getEdge(at(r, c), at(r - 1, c - 1));
                e->triangleOppositeNodes[0] = at(r, c - 1);
                e->triangleOppositeNodes[1] = at(r - 1, c);
            }
        }
    return mesh;
}

MeshGraph buildMinimalMesh(const point_cloud &cams, const std::vector<surface_model> &previous)
{
    if (cams.size() < 2)
        return MeshGraph();
    mesh_inputs m;
    prepare_mesh_inputs(cams, previous, m);
    const double grid = m.nearest2.size() < 2 ? 10.0 : std::sqrt(m.nearest2[m.nearest2.size() / 2]);
    const double med = median_height(m, grid);
    const double border = std::max(0.0, std::min(1000.0, med * 2));
    const double x0 = m.cam_min[0] - border, x1 = m.cam_max[0] + border, y0 = m.cam_min[1] - border, y1 = m.cam_max[1] + border;
    auto z_at = [&](double x, double y) {
        return m.vertices.size() > 0 ? m.vertices.nearest_payload(x, y) : m.cameras.nearest_payload(x, y) - med;
    };
    // two triangles (0, 1, 3) and (0, 3, 2) over the bounding rectangle
    MeshGraph mesh;
    const size_t v0 = mesh.addNode(x0, y0, z_at(x0, y0)), v1 = mesh.addNode(x1, y0, z_at(x1, y0));
    const size_t v2 = mesh.addNode(x0, y1, z_at(x0, y1)), v3 = mesh.addNode(x1, y1, z_at(x1, y1));
    auto rim = [](size_t opposite) {
        MeshEdge e;
        e.border = true;
        e.triangleOppositeNodes[0] = opposite;
        return e;
    };
    mesh.addEdge(rim(v3), v0, v1);
    mesh.addEdge(rim(v0), v1, v3);
    mesh.addEdge(rim(v0), v2, v3);
    mesh.addEdge(rim(v3), v0, v2);
    MeshEdge diagonal;
    diagonal.triangleOppositeNodes[0] = v1;
    diagonal.triangleOppositeNodes[1] = v2;
    mesh.addEdge(diagonal, v0, v3);
    return mesh;
}

namespace
{

// robustCentroid (relax_cost_function.hpp:73-117) in doubles
v3 robust_centroid(const v3 *pts, int n, double huber_threshold)
{
    v3 c{0, 0, 0};
    for (int i = 0; i < n; i++)
        c = add(c, pts[i]);
    c = v3{c.x / n, c.y / n, c.z / n};
    double w[5];
    for (int stage = 0; stage < 3; stage++)
    {
        double total = 0, mn = std::numeric_limits<double>::max(), mx = 0;
        for (int i = 0; i < n; i++)
        {
            const v3 d = sub(pts[i], c);
            const double err = std::sqrt(dot(d, d));
            double wi = 1.0 / (err + 1e-8);
            if (err > huber_threshold)
                wi *= huber_threshold / err;
            w[i] = wi;
            total += wi;
            mn = std::min(mn, wi);
            mx = std::max(mx, wi);
        }
        v3 s{0, 0, 0};
        for (int i = 0; i < n; i++)
            s = add(s, mul(pts[i], w[i]));
        c = v3{s.x / total, s.y / total, s.z / total};
        if (mn > mx * 0.5)
            break;
    }
    return c;
}

inline uint64_t cell_key(int i, int j) // gridCellKey, grid_filter.hpp:11-14
{
    return (static_cast<uint64_t>(i) << 32) | static_cast<uint32_t>(j);
}

class GroundMeshProblem
{
  public:
    GroundMeshProblem(ochip_ctx *ctx, const MeasurementGraph &graph) : _ctx(ctx), _graph(graph)
    {
    }
    ~GroundMeshProblem()
    {
        if (_dev)
            ochip_relaxg_problem_destroy(_dev);
    }

    // setupGroundMeshProblem (relax_problem.cpp:83-120)
    bool setup(std::vector<NodePose> &poses, std::vector<std::pair<size_t, CameraModel>> &cam_models,
               const std::vector<size_t> &edges_to_optimize, const RelaxConfig &config,
               const std::vector<surface_model> &previous, RelaxMeshStats *stats, std::string *error,
               const RelaxShard *shard = nullptr)
    {
        const bool verbose = ochip_verbose("relax");
        auto tmark = clk::now();
        auto lap = [&](const char *what) {
            if (verbose)
                fprintf(stderr, "[relax mesh setup] %-28s %.3f ms\n", what, since(tmark) * 1e3);
            tmark = clk::now();
        };
        _poses = &poses;
        _cam_models = &cam_models;
        _options = config.options;
        _intrinsics = (config.options & (OPT_FOCAL_LENGTH | OPT_PRINCIPAL_POINT | OPT_LENS_DISTORTIONS_RADIAL)) != 0;
        for (size_t i = 0; i < poses.size(); i++)
            if (_opt_index.emplace(poses[i].node_id, i).second) // the first pose of a node is the one optimised
            {
                _cam_of_node.emplace(poses[i].node_id, (uint32_t)_cam_opt.size());
                _pose_cam.push_back((uint32_t)_cam_opt.size());
                push_camera(poses[i].position, poses[i].orientation, true);
            }
            else
                _pose_cam.push_back(UINT32_MAX);
        initialize_mesh(previous, (config.options & OPT_MINIMAL_MESH) != 0);
        const double frac = config.ground_mesh_grid_fraction;

        // which edges take part, with their poses (nodeid2poseopt, :181-232)
        std::vector<const MeasurementGraph::Edge *> edges;
        std::vector<pose_ref> src, dst;
        for (size_t id : edges_to_optimize)
        {
            const MeasurementGraph::Edge *e = _graph.getEdge(id);
            if (e == nullptr)
                continue;
            edges.push_back(e);
            src.push_back(lookup(e->source));
            dst.push_back(lookup(e->dest));
        }
        lap("mesh + pose lookup");
        build_tracks(edges, src, dst, frac);
        lap("tracks");
        if (stats)
            stats->track_blocks = (int)_blk_n.size();
        const size_t n_track_blocks = _blk_n.size();

        // gridFilterMatchesPerImage (:234-309): the pass stops at the first edge without usable poses
        size_t n_filter = edges.size();
        for (size_t k = 0; k < edges.size(); k++)
            if (src[k].loc == nullptr || dst[k].loc == nullptr)
            {
                n_filter = k;
                break;
            }
        // on the device (ochip_plane_setup_create through relax_util.hpp's device_filter: the scores and the per-cell
        // selection; edges it only flags are decided by the host's grid_filter)
        std::vector<std::vector<uint8_t>> keep(edges.size());
        {
            const double no_triangle[6] = {0, 0, 1, 0, 0, 1}; // (only the plane flavour's block list uses it)
            device_filter df;
            if (!df.run(_ctx, _graph, edges, src, dst, n_filter, _cam_pos, _cam_q, no_triangle, frac, error))
                return false;
            for (size_t j = 0, k = 0; j < df.pe.size(); j++, k++)
            {
                while (edges[k] != df.edge[j]) // (run skips nothing here: every entry of `edges` is an edge)
                    k++;
                keep[k].assign(df.keep.begin() + df.pe[j].inlier_offset, df.keep.begin() + df.pe[j].inlier_offset + df.pe[j].n_inliers);
            }
            if (relax_setup_check_on())
            {
                for (size_t k = 0; k < n_filter; k++)
                    if (keep[k] != grid_filter(_graph, *edges[k], src[k], dst[k], frac))
                    {
                        if (error)
                            *error = "relax mesh set-up: the device's grid filter differs from the host's on edge " + std::to_string(k);
                        return false;
                    }
                relax_setup_check_passed();
            }
        }
        lap("grid filter (device)");
        // addRayTriangleMeasurementCost (:388-560) per edge, every edge with a mesh walker of its own
        std::vector<edge_blocks> per_edge(edges.size());
        const bool mesh_ok = _mesh.size_nodes() > 0 && _mesh.size_edges() > 0;
        for (size_t k = 0; k < edges.size(); k++) // every edge gives its source model an inverse twin (:402-407)
            if (src[k].loc != nullptr && dst[k].loc != nullptr)
                twin_for(model_of(edges[k]->source));
#pragma omp parallel for schedule(dynamic, 4)
        for (size_t k = 0; k < edges.size(); k++)
            if (mesh_ok && src[k].loc != nullptr && dst[k].loc != nullptr)
                two_ray_blocks(*edges[k], src[k], dst[k], keep[k], frac, per_edge[k]);
        for (const edge_blocks &pe : per_edge)
            for (size_t b = 0; b < pe.tri.size() / 3; b++)
            {
                _blk_n.push_back(2);
                _blk_intr.push_back(pe.intr_model != (size_t)-1 ? 1 : 0);
                if (pe.intr_model != (size_t)-1)
                    observe(pe.intr_model);
                _blk_ray_off.push_back((uint32_t)_ray_cam.size() + 2);
                _ray_cam.push_back(pe.cams[2 * b]);
                _ray_cam.push_back(pe.cams[2 * b + 1]);
                _ray_dir.insert(_ray_dir.end(), pe.rays.begin() + 6 * b, pe.rays.begin() + 6 * b + 6);
                _ray_px.insert(_ray_px.end(), pe.px.begin() + 4 * b, pe.px.begin() + 4 * b + 4);
                _blk_tri.insert(_blk_tri.end(), pe.tri.begin() + 3 * b, pe.tri.begin() + 3 * b + 3);
            }
        lap("2-ray blocks");
        if (stats)
        {
            stats->two_ray_blocks = (int)(_blk_n.size() - n_track_blocks);
            stats->mesh_vertices = (int)_mesh.size_nodes();
        }

        // mesh priors: addMeshFlatPrior (:1303-1333), addMeshSmoothPrior (:1335-1366)
        std::vector<uint32_t> diff_v, smooth_v;
        for (const MeshEdge &e : _mesh.edges)
        {
            diff_v.push_back((uint32_t)e.source);
            diff_v.push_back((uint32_t)e.dest);
        }
        for (const MeshEdge &e : _mesh.edges)
            if (!e.border)
            {
                smooth_v.push_back((uint32_t)e.source);
                smooth_v.push_back((uint32_t)e.dest);
                smooth_v.push_back((uint32_t)e.triangleOppositeNodes[0]);
                smooth_v.push_back((uint32_t)e.triangleOppositeNodes[1]);
            }
        std::vector<double> vxy(2 * _mesh.size_nodes()), vz(_mesh.size_nodes());
        std::vector<uint8_t> vopt(_mesh.size_nodes(), 1);
        for (size_t v = 0; v < _mesh.size_nodes(); v++)
        {
            vxy[2 * v] = _mesh.nodes[v].location[0];
            vxy[2 * v + 1] = _mesh.nodes[v].location[1];
            vz[v] = _mesh.nodes[v].location[2];
        }
        ochip_relaxg_desc d{};
        d.n_cams = (uint32_t)_cam_opt.size();
        d.cam_pos = _cam_pos.data();
        d.cam_q = _cam_q.data();
        d.cam_optimize = _cam_opt.data();
        d.n_verts = (uint32_t)_mesh.size_nodes();
        d.vert_xy = vxy.data();
        d.vert_z = vz.data();
        d.vert_optimize = vopt.data();
        d.n_blocks = (uint32_t)_blk_n.size();
        d.blk_n = _blk_n.data();
        d.blk_intr = _blk_intr.data();
        d.blk_ray_off = _blk_ray_off.data();
        d.blk_tri = _blk_tri.data();
        d.ray_cam = _ray_cam.data();
        d.ray_dir = _ray_dir.data();
        d.ray_px = _ray_px.data();
        if (_shared_model != (size_t)-1)
        {
            // the one inverse lens model the FocalRadial blocks share (:483-507, :799-853)
            const InverseCameraModel &m = twin_of(_shared_model)->m;
            d.model[0] = m.focal_length_pixels;
            d.model[1] = m.principle_point[0], d.model[2] = m.principle_point[1];
            for (int i = 0; i < 3; i++)
                d.model[3 + i] = m.radial_distortion[i];
            d.model[6] = m.tangential_distortion[0], d.model[7] = m.tangential_distortion[1];
            d.opt_focal = (_options & OPT_FOCAL_LENGTH) != 0;
            d.opt_principal = (_options & OPT_PRINCIPAL_POINT) != 0;
            // SubsetManifold of the radial block (:533-556): Brown k1 / k1 k2 / k1 k2 k3; with no parameterisation chosen -
            // also when LENS_DISTORTIONS_RADIAL is not among the options at all - the block stays a free Euclidean one
            const bool radial = (_options & OPT_LENS_DISTORTIONS_RADIAL) != 0;
            d.n_radial_free = 3;
            if (radial && !(_options & OPT_LENS_DISTORTIONS_RADIAL_BROWN246_PARAMETERIZATION))
            {
                if (_options & OPT_LENS_DISTORTIONS_RADIAL_BROWN24_PARAMETERIZATION)
                    d.n_radial_free = 2;
                else if (_options & OPT_LENS_DISTORTIONS_RADIAL_BROWN2_PARAMETERIZATION)
                    d.n_radial_free = 1;
            }
            d.mono_observations = (uint32_t)_mono_count; // addMonotonicityCosts (:1381-1388)
            const double hc = m.pixels_cols / 2.0, hr = m.pixels_rows / 2.0;
            d.mono_r_max = std::sqrt(hc * hc + hr * hr) / _mono_focal;
        }
        d.n_down = 0;
        d.n_diff = (uint32_t)(diff_v.size() / 2);
        d.diff_v = diff_v.data();
        d.diff_weight = 1e-4;
        d.anchor_weight = 1e-5;
        d.n_smooth = (uint32_t)(smooth_v.size() / 4);
        d.smooth_v = smooth_v.data();
        d.smooth_weight = 1e-4;
        d.huber_a = 1 * M_PI / 180;
        d.focal_lo = 100.0;
        d.focal_hi = 20000.0;
        if (_several_models)
        {
            if (error)
                *error = "relax: the device path optimises one shared lens model per group; this group holds images of several";
            return false;
        }
        const bool sharded = shard && (shard->world > 1 || shard->exchange);
        if (sharded && _views_without_features.load())
        {
            // a view whose node carries no feature list is skipped (the reference's bounds check), so a rank that holds only
            // its own block's features (och_shard_*: the sharded survey) would build other blocks than its peers and the
            // exchange would meet different record counts: refuse instead of diverging
            if (error)
                *error = "relax: a sharded solve of the mesh flavours needs every rank to hold the feature lists of all the group's images";
            return false;
        }
        if (sharded)
        {
            d.shard_rank = shard->rank;
            d.shard_world = shard->world;
        }
        if (ochip_relaxg_problem_create(_ctx, &d, &_dev) != OCHIP_OK)
        {
            if (error)
                *error = std::string("ochip_relaxg_problem_create: ") + ochip_last_error(_ctx);
            return false;
        }
        if (sharded && ochip_relaxg_set_exchange(_dev, shard->exchange, shard->user) != OCHIP_OK)
        {
            if (error)
                *error = std::string("ochip_relaxg_set_exchange: ") + ochip_last_error(_ctx);
            return false;
        }
        lap("ochip_relaxg_problem_create");
        return true;
    }

    // relaxObservedModelOnly (:931-984): everything but the mesh heights held constant
    bool relax_observed_model_only(RelaxTimers *t, RelaxMeshStats *stats, std::string *error)
    {
        if (ochip_relaxg_set_structure_only(_dev, 1) != OCHIP_OK)
            return fail(error, "ochip_relaxg_set_structure_only");
        const bool ok = solve(t, stats, error);
        if (ochip_relaxg_set_structure_only(_dev, 0) != OCHIP_OK)
            return fail(error, "ochip_relaxg_set_structure_only");
        return ok;
    }

    bool solve(RelaxTimers *t, RelaxMeshStats *stats, std::string *error)
    {
        ochip_relax_options o{100, 1.0, 1e-6, 1e-10, 1e-8}; // relax_problem.cpp:30-37 + Ceres defaults
        ochip_relax_summary s{};
        if (ochip_relaxg_solve(_dev, &o, &s) != OCHIP_OK)
            return fail(error, "ochip_relaxg_solve");
        if (t)
        {
            t->solves++;
            t->iterations_total += s.iterations;
            t->last_iterations = s.iterations;
            t->last_initial_cost = s.initial_cost;
            t->last_final_cost = s.final_cost;
            t->last_residual_blocks = s.num_residual_blocks;
        }
        if (stats)
            stats->unknowns = s.num_parameters;
        std::vector<double> q(_cam_opt.size() * 4), z(_mesh.size_nodes());
        double model[8];
        if (ochip_relaxg_get_state(_dev, q.data(), z.data(), model) != OCHIP_OK)
            return fail(error, "ochip_relaxg_get_state");
        if (_shared_model != (size_t)-1)
        {
            InverseCameraModel &m = twin_of(_shared_model)->m;
            m.focal_length_pixels = model[0];
            m.principle_point[0] = model[1], m.principle_point[1] = model[2];
            for (int i = 0; i < 3; i++)
                m.radial_distortion[i] = model[3 + i];
        }
        // "copy back camera models" (:1415-1419): every model with an inverse twin becomes the forward fit of the twin
        for (const twin &t : _twins)
            for (auto &cm : *_cam_models)
                if (cm.first == t.id)
                    cm.second = convertModel(t.m, t.id);
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
        for (size_t v = 0; v < z.size(); v++)
            _mesh.nodes[v].location[2] = z[v];
        return true;
    }

    // getSurfaceModel (:1422-1507): one robust point per merged track + the mesh
    void surface(surface_model *out)
    {
        out->cloud.clear();
        point_cloud pts;
        for (const auto &tr : _tracks_for_cloud)
        {
            const size_t views = tr.views;
            if (tr.min_error > (views >= 3 ? 10.0 : 1.0))
                continue;
            if (tr.n_points == 1)
                pts.push_back({tr.points[0].x, tr.points[0].y, tr.points[0].z});
            else
            {
                const int n = std::min<int>((int)tr.n_points, 5);
                const v3 c = robust_centroid(tr.points, n, 1.0);
                pts.push_back({c.x, c.y, c.z});
            }
        }
        if (!pts.empty())
            out->cloud.push_back(std::move(pts));
        out->mesh = _mesh;
    }

  private:
    struct edge_blocks
    {
        std::vector<uint32_t> cams, tri;
        std::vector<double> rays, px;
        size_t intr_model = (size_t)-1; // id of the shared model when the edge's blocks use the FocalRadial functor
    };
    struct twin // _inverse_cam_model_to_optimize
    {
        size_t id;
        InverseCameraModel m;
    };
    twin *twin_of(size_t id)
    {
        for (twin &t : _twins)
            if (t.id == id)
                return &t;
        return nullptr;
    }
    twin *twin_for(const CameraModel &forward)
    {
        if (twin *t = twin_of(forward.id))
            return t;
        _twins.push_back(twin{forward.id, convertModel(forward)});
        return &_twins.back();
    }
    void observe(size_t model_id) // trackRadialObservation (:1368-1379) of a block on the shared model
    {
        if (_shared_model == (size_t)-1)
        {
            _shared_model = model_id;
            _mono_focal = twin_of(model_id)->m.focal_length_pixels;
        }
        else if (_shared_model != model_id)
            _several_models = true;
        _mono_count++;
    }
    static bool same_model(const CameraModel &a, const CameraModel &b) // CameraModel::operator== (camera_model.hpp:78-81)
    {
        return a.id == b.id && a.pixels_rows == b.pixels_rows && a.pixels_cols == b.pixels_cols &&
               a.focal_length_pixels == b.focal_length_pixels && a.principle_point[0] == b.principle_point[0] &&
               a.principle_point[1] == b.principle_point[1] && a.radial_distortion[0] == b.radial_distortion[0] &&
               a.radial_distortion[1] == b.radial_distortion[1] && a.radial_distortion[2] == b.radial_distortion[2] &&
               a.tangential_distortion[0] == b.tangential_distortion[0] && a.tangential_distortion[1] == b.tangential_distortion[1];
    }
    struct cloud_track
    {
        v3 points[5]; // the first five finite points (robustCentroid looks at no more)
        size_t n_points = 0;
        double min_error = std::numeric_limits<double>::infinity();
        size_t views = 0;
    };

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
    pose_ref lookup(size_t node_id)
    {
        pose_ref po;
        auto it = _opt_index.find(node_id);
        if (it != _opt_index.end())
        {
            NodePose &np = (*_poses)[it->second];
            po.optimize = true;
            po.loc = np.position;
            po.rot = np.orientation;
            po.cam = _cam_of_node.at(node_id);
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
    const CameraModel &model_of(size_t node_id) const // the group's copy of the node's model if there is one (:207-229)
    {
        const CameraModel &m = *_graph.getNode(node_id)->payload.model;
        for (const auto &cm : *_cam_models)
            if (cm.first == m.id)
                return cm.second;
        return m;
    }

    void initialize_mesh(const std::vector<surface_model> &previous, bool minimal) // initializeGroundMesh (:1244-1288)
    {
        point_cloud cams;
        for (size_t i = 0; i < _poses->size(); i++)
            if (_pose_cam[i] != UINT32_MAX)
                cams.push_back({(*_poses)[i].position[0], (*_poses)[i].position[1], (*_poses)[i].position[2]});
        const MeshGraph *prev = nullptr;
        for (const surface_model &s : previous)
            if (s.mesh.size_nodes() > 0)
            {
                prev = &s.mesh;
                break;
            }
        const bool prev_is_plane_triangle = prev != nullptr && prev->size_nodes() == 3;
        if (prev != nullptr && !(minimal && prev_is_plane_triangle))
            _mesh = *prev;
        else if (minimal)
            _mesh = buildMinimalMesh(cams, previous);
        else
            _mesh = rebuildMesh(cams, previous);
    }

    // collectEdgeTracks (:351-386) + addMultiRayTrackCosts (:608-929).  The reference builds one FeatureTrack per inlier
    // of every edge, merges the ones that share a measurement (union-find over a hash map of measurements), keeps per image
    // and grid cell the longest merged track, and turns the survivors into 3..5-ray blocks.  Same sets, same order (tracks
    // in the order of their first two-view member, rays in order of appearance), on flat arrays: measurements are
    // (node index, feature index) pairs, so "who held this measurement first" is an array per image.
    void build_tracks(const std::vector<const MeasurementGraph::Edge *> &edges, const std::vector<pose_ref> &src,
                      const std::vector<pose_ref> &dst, double frac)
    {
        const size_t n_nodes = _graph.size_nodes();
        const bool verbose = ochip_verbose("relax");
        auto tmark = clk::now();
        auto lap = [&](const char *what) {
            if (verbose)
                fprintf(stderr, "[relax mesh tracks] %-27s %.3f ms\n", what, since(tmark) * 1e3);
            tmark = clk::now();
        };
        struct two_view
        {
            uint32_t na, nb, fa, fb; // node indices and feature indices of the two measurements
            v3 point;
            double error;
        };
        std::vector<size_t> first(edges.size() + 1, 0);
        for (size_t k = 0; k < edges.size(); k++)
            first[k + 1] = first[k] + ((src[k].loc && dst[k].loc) ? edges[k]->payload.inlier_matches.size() : 0);
        std::vector<two_view> tv(first.back());
        if (tv.empty())
            return;
        if (tv.size() >= UINT32_MAX)
            return; // (far beyond what fits the device anyway)
#pragma omp parallel for schedule(dynamic, 4)
        for (size_t k = 0; k < edges.size(); k++)
        {
            if (first[k + 1] == first[k])
                continue;
            const MeasurementGraph::Edge &e = *edges[k];
            const CameraModel &sm = model_of(e.source), &dm = model_of(e.dest);
            const v3 so{src[k].loc[0], src[k].loc[1], src[k].loc[2]}, d_o{dst[k].loc[0], dst[k].loc[1], dst[k].loc[2]};
            const uint32_t na = (uint32_t)_graph.nodeIndex(e.source), nb = (uint32_t)_graph.nodeIndex(e.dest);
            two_view *out = &tv[first[k]];
            for (const feature_match_denormalized &m : e.payload.inlier_matches)
            {
                double r1[3], r2[3];
                image_to_3d(m.pixel_1, sm, r1);
                image_to_3d(m.pixel_2, dm, r2);
                out->na = na, out->nb = nb;
                out->fa = (uint32_t)m.feature_index_1, out->fb = (uint32_t)m.feature_index_2;
                ray_intersection(rotate(src[k].rot, v3{r1[0], r1[1], r1[2]}), so, rotate(dst[k].rot, v3{r2[0], r2[1], r2[2]}), d_o,
                                 &out->point, &out->error);
                out++;
            }
        }
        lap("two-view intersections");
        // Merge: the measurements (image, feature) are the vertices, every two-view track an edge between two of them; a
        // merged track is a connected component.  Lock-free union-find over the measurement ids (the smaller id always
        // becomes the root, so the outcome does not depend on the threads' interleaving).
        const uint32_t M = (uint32_t)tv.size();
        std::vector<uint32_t> meas_off(n_nodes + 1, 0);
        {
            std::vector<uint32_t> n_feat(n_nodes, 0);
            for (const two_view &t : tv)
            {
                n_feat[t.na] = std::max(n_feat[t.na], t.fa + 1);
                n_feat[t.nb] = std::max(n_feat[t.nb], t.fb + 1);
            }
            for (size_t i = 0; i < n_nodes; i++)
                meas_off[i + 1] = meas_off[i] + n_feat[i];
        }
        const uint32_t n_meas = meas_off[n_nodes];
        std::unique_ptr<std::atomic<uint32_t>[]> parent(new std::atomic<uint32_t>[n_meas]);
#pragma omp parallel for schedule(static)
        for (uint32_t i = 0; i < n_meas; i++)
            parent[i].store(i, std::memory_order_relaxed);
        auto find = [&](uint32_t x) {
            while (true)
            {
                uint32_t p = parent[x].load(std::memory_order_relaxed);
                if (p == x)
                    return x;
                const uint32_t gp = parent[p].load(std::memory_order_relaxed);
                if (gp != p)
                    parent[x].compare_exchange_weak(p, gp, std::memory_order_relaxed);
                x = p;
            }
        };
#pragma omp parallel for schedule(static)
        for (uint32_t i = 0; i < M; i++)
        {
            uint32_t a = meas_off[tv[i].na] + tv[i].fa, b = meas_off[tv[i].nb] + tv[i].fb;
            while (true)
            {
                a = find(a);
                b = find(b);
                if (a == b)
                    break;
                if (a < b)
                    std::swap(a, b);
                uint32_t expected = a; // a > b: a (still a root?) goes under b
                if (parent[a].compare_exchange_strong(expected, b, std::memory_order_relaxed))
                    break;
            }
        }
        lap("union-find");
        // tracks in the order of their first two-view member: the component's smallest member index, ranked
        std::vector<uint32_t> root_of(M);
        std::unique_ptr<std::atomic<uint32_t>[]> first_member(new std::atomic<uint32_t>[n_meas]);
#pragma omp parallel for schedule(static)
        for (uint32_t i = 0; i < n_meas; i++)
            first_member[i].store(UINT32_MAX, std::memory_order_relaxed);
#pragma omp parallel for schedule(static)
        for (uint32_t i = 0; i < M; i++)
        {
            const uint32_t r = find(meas_off[tv[i].na] + tv[i].fa);
            root_of[i] = r;
            uint32_t cur = first_member[r].load(std::memory_order_relaxed);
            while (i < cur && !first_member[r].compare_exchange_weak(cur, i, std::memory_order_relaxed))
            {
            }
        }
        std::vector<uint32_t> track_of(M), members;
        uint32_t n_tracks = 0;
        {
            std::vector<uint32_t> rank(M); // rank of member i among the first members, where it is one
            for (uint32_t i = 0; i < M; i++)
                if (first_member[root_of[i]].load(std::memory_order_relaxed) == i)
                    rank[i] = n_tracks++;
            members.assign(n_tracks, 0);
#pragma omp parallel for schedule(static)
            for (uint32_t i = 0; i < M; i++)
                track_of[i] = rank[first_member[root_of[i]].load(std::memory_order_relaxed)];
            for (uint32_t i = 0; i < M; i++)
                members[track_of[i]]++;
        }
        lap("track ids");
        // per track: its rays (usable measurements of distinct images, in order of appearance), the distinct images of all
        // of its measurements, and what the surface model's cloud needs (the first 5 finite points, the smallest error)
        struct view
        {
            uint32_t node, feature;
        };
        std::vector<uint32_t> off(n_tracks + 1, 0);
        for (uint32_t t = 0; t < n_tracks; t++)
            off[t + 1] = off[t] + 2 * members[t];
        std::vector<view> rays(off.back()), imgs(off.back());
        std::vector<uint64_t> ray_cell(off.back()); // the grid cell of every ray's pixel in its image
        std::vector<uint32_t> n_rays(n_tracks, 0), n_imgs(n_tracks, 0);
        std::vector<cloud_track> cloud(n_tracks);
        std::vector<int64_t> pose_of_node(n_nodes, -1); // index into *_poses of the pose that is optimised, per node index
        for (const auto &kv : _opt_index)
            pose_of_node[_graph.nodeIndex(kv.first)] = (int64_t)kv.second;
        const auto &gnodes = _graph.nodes();
        // members grouped by track (ascending inside a track), then the tracks in parallel
        std::vector<uint32_t> moff(n_tracks + 1, 0), mlist(M);
        for (uint32_t t = 0; t < n_tracks; t++)
            moff[t + 1] = moff[t] + members[t];
        {
            std::vector<uint32_t> fill(moff.begin(), moff.end() - 1);
            for (uint32_t i = 0; i < M; i++)
                mlist[fill[track_of[i]]++] = i;
        }
#pragma omp parallel for schedule(dynamic, 256)
        for (uint32_t t = 0; t < n_tracks; t++)
        {
            view *tr = &rays[off[t]], *ti = &imgs[off[t]];
            uint32_t nr = 0, ni = 0;
            cloud_track &ct = cloud[t];
            for (uint32_t e = moff[t]; e < moff[t + 1]; e++)
            {
                const two_view &m = tv[mlist[e]];
                const bool finite_point = std::isfinite(m.point.x) && std::isfinite(m.point.y) && std::isfinite(m.point.z);
                for (int side = 0; side < 2; side++)
                {
                    const view v = side == 0 ? view{m.na, m.fa} : view{m.nb, m.fb};
                    bool present = false;
                    for (uint32_t k = 0; k < nr && !present; k++)
                        present = tr[k].node == v.node;
                    if (!present && pose_of_node[v.node] >= 0 && v.feature < gnodes[v.node].payload.features.size())
                        tr[nr++] = v;
                    else if (!present && pose_of_node[v.node] >= 0 && gnodes[v.node].payload.features.empty())
                        _views_without_features.store(true, std::memory_order_relaxed); // (the node's feature list is not on this rank: a graph from survey_sharded; a single index out of range is skipped on every rank alike, as the reference's bounds check does)
                    if (finite_point)
                    {
                        bool seen = false;
                        for (uint32_t k = 0; k < ni && !seen; k++)
                            seen = ti[k].node == v.node;
                        if (!seen)
                            ti[ni++] = v;
                    }
                }
                if (finite_point)
                {
                    if (ct.n_points < 5)
                        ct.points[ct.n_points] = m.point;
                    ct.n_points++;
                    if (std::isfinite(m.error))
                        ct.min_error = std::min(ct.min_error, m.error);
                }
            }
            n_rays[t] = nr;
            n_imgs[t] = ni;
            if (nr >= 3)
                for (uint32_t k = 0; k < nr; k++)
                {
                    const MeasurementGraph::Node &node = gnodes[tr[k].node];
                    const double *px = node.payload.features[tr[k].feature].location;
                    const CameraModel &model = *node.payload.model;
                    ray_cell[off[t] + k] = cell_key((int)std::floor((px[0] / model.pixels_cols) / frac),
                                                    (int)std::floor((px[1] / model.pixels_rows) / frac));
                }
        }
        lap("rays per track");
        // NOTE: the reference's getSurfaceModel unites only two-view tracks with a finite point, which can split a set that
        // a non-finite one bridges; such tracks have parallel rays and do not occur between overlapping images.
        for (uint32_t t = 0; t < n_tracks; t++)
            if (cloud[t].n_points > 0)
            {
                cloud[t].views = n_imgs[t];
                _tracks_for_cloud.push_back(cloud[t]);
            }
        std::vector<cloud_track>().swap(cloud);

        lap("cloud");
        // per image and grid cell the longest track wins (the first one among equally long ones)
        auto cell_of = [&](const view &v) {
            const MeasurementGraph::Node &node = gnodes[v.node];
            const double *px = node.payload.features[v.feature].location;
            const CameraModel &model = *node.payload.model;
            return cell_key((int)std::floor((px[0] / model.pixels_cols) / frac), (int)std::floor((px[1] / model.pixels_rows) / frac));
        };
        // (a dense cell table per image - pixels / image size lies in [0, 1) for features inside the image - with a map for
        // whatever falls outside of it)
        const int G = (int)std::ceil(1.0 / frac) + 2;
        std::vector<std::pair<uint32_t, uint32_t>> table(n_nodes * (size_t)G * G, std::make_pair(0u, UINT32_MAX)); // (length, track)
        std::map<std::pair<uint32_t, uint64_t>, std::pair<uint32_t, uint32_t>> outside; // (node, cell)
        for (uint32_t t = 0; t < n_tracks; t++)
        {
            if (n_rays[t] < 3)
                continue;
            for (uint32_t k = 0; k < n_rays[t]; k++)
            {
                const view &v = rays[off[t] + k];
                const uint64_t cell = ray_cell[off[t] + k];
                const int ci = (int)(int32_t)(cell >> 32), cj = (int)(int32_t)(uint32_t)cell;
                std::pair<uint32_t, uint32_t> *slot;
                if (ci >= 0 && cj >= 0 && ci < G && cj < G)
                    slot = &table[((size_t)v.node * G + ci) * G + cj];
                else
                    slot = &outside.emplace(std::make_pair(v.node, cell), std::make_pair(0u, UINT32_MAX)).first->second;
                if (slot->second == UINT32_MAX || slot->first < n_rays[t])
                    *slot = std::make_pair(n_rays[t], t);
            }
        }
        std::vector<char> accepted(n_tracks, 0);
        for (const auto &slot : table)
            if (slot.second != UINT32_MAX)
                accepted[slot.second] = 1;
        for (const auto &kv : outside)
            accepted[kv.second.second] = 1;

        lap("cell winners");
        TriangleWalker walker;
        if (!walker.init(_mesh))
            return;
        _track_measurement.assign(n_nodes, {});
        _covered.assign(n_nodes, {});
        struct ray_info
        {
            view v;
            uint32_t cam;
            size_t model_id;
            v3 loc, ray;
            double px[2];
            const double *rot;
        };
        std::vector<ray_info> tr;
        for (uint32_t t = 0; t < n_tracks; t++)
        {
            if (n_rays[t] < 3 || !accepted[t])
                continue;
            tr.clear();
            v3 mean{0, 0, 0};
            for (uint32_t k = 0; k < n_rays[t]; k++)
            {
                ray_info r;
                r.v = rays[off[t] + k];
                const MeasurementGraph::Node &node = gnodes[r.v.node];
                const NodePose &np = (*_poses)[(size_t)pose_of_node[r.v.node]];
                r.cam = _cam_of_node.at(node.id);
                r.loc = v3{np.position[0], np.position[1], np.position[2]};
                double ray[3];
                image_to_3d(node.payload.features[r.v.feature].location, *node.payload.model, ray);
                r.ray = v3{ray[0], ray[1], ray[2]};
                r.model_id = node.payload.model->id;
                r.px[0] = node.payload.features[r.v.feature].location[0];
                r.px[1] = node.payload.features[r.v.feature].location[1];
                r.rot = np.orientation;
                tr.push_back(r);
                mean = add(mean, r.loc);
            }
            const double nr = (double)tr.size();
            mean = v3{mean.x / nr, mean.y / nr, mean.z / nr};
            v3 x01;
            double gap;
            ray_intersection(rotate(tr[0].rot, tr[0].ray), tr[0].loc, rotate(tr[1].rot, tr[1].ray), tr[1].loc, &x01, &gap);
            if (!(std::isfinite(x01.x) && std::isfinite(x01.y) && std::isfinite(x01.z)))
                continue;
            if (walker.find(v3{0, 0, -1}, v3{x01.x, x01.y, mean.z}) != TriangleWalker::INTERSECTION)
                continue;
            const size_t tri[3] = {walker.tri[0], walker.tri[1], walker.tri[2]};
            // where every ray meets the triangle's plane; rays further than 3 x the median from the robust centre go
            v3 c[3];
            for (int i = 0; i < 3; i++)
                c[i] = v3{_mesh.nodes[tri[i]].location[0], _mesh.nodes[tri[i]].location[1], _mesh.nodes[tri[i]].location[2]};
            v3 nrm = cross(sub(c[0], c[1]), sub(c[0], c[2]));
            {
                const double n2 = dot(nrm, nrm);
                if (n2 > 0)
                {
                    const double n = std::sqrt(n2);
                    nrm = v3{nrm.x / n, nrm.y / n, nrm.z / n};
                }
            }
            std::vector<v3> hits(tr.size());
            bool all_valid = true;
            double avg = 0;
            for (size_t i = 0; i < tr.size(); i++)
            {
                const v3 dir = rotate(tr[i].rot, tr[i].ray);
                const double denom = dot(nrm, dir);
                if (std::abs(denom) < 1e-9)
                {
                    all_valid = false;
                    hits[i] = v3{NAN, NAN, NAN};
                }
                else
                    hits[i] = add(tr[i].loc, mul(dir, (dot(nrm, c[0]) - dot(tr[i].loc, nrm)) / denom));
                const v3 dd = sub(hits[i], tr[i].loc);
                avg += std::sqrt(dot(dd, dd));
            }
            if (!all_valid)
                continue;
            avg /= nr;
            const v3 centre = robust_centroid(hits.data(), std::min<int>((int)hits.size(), 5), avg * 0.01);
            std::vector<std::pair<double, size_t>> scores(tr.size());
            for (size_t i = 0; i < tr.size(); i++)
            {
                const v3 dd = sub(hits[i], centre);
                scores[i] = {std::sqrt(dot(dd, dd)) / avg, i};
            }
            std::sort(scores.begin(), scores.end());
            const double threshold = std::max(scores[scores.size() / 2].first * 3.0, 1e-6);
            std::vector<const ray_info *> good;
            for (const auto &es : scores)
                if (es.first <= threshold && good.size() < 5)
                    good.push_back(&tr[es.second]);
            if (good.size() < 3)
                continue;
            bool one_model = true;
            for (const ray_info *r : good)
                one_model &= r->model_id == good[0]->model_id;
            const bool focal_radial = one_model && _intrinsics;
            if (focal_radial)
            {
                twin_for(*gnodes[good[0]->v.node].payload.model);
                observe(good[0]->model_id);
            }
            _blk_n.push_back((uint8_t)good.size());
            _blk_intr.push_back(focal_radial ? 1 : 0);
            _blk_ray_off.push_back((uint32_t)(_ray_cam.size() + good.size()));
            for (const ray_info *r : good)
            {
                _ray_cam.push_back(r->cam);
                _ray_dir.push_back(r->ray.x);
                _ray_dir.push_back(r->ray.y);
                _ray_dir.push_back(r->ray.z);
                _ray_px.push_back(r->px[0]);
                _ray_px.push_back(r->px[1]);
                _track_measurement[r->v.node].emplace(r->v.feature, 1);
                _covered[r->v.node].emplace(cell_of(r->v), 1);
            }
            for (int i = 0; i < 3; i++)
                _blk_tri.push_back((uint32_t)tri[i]);
        }
        lap("track blocks");
    }

    // the 2-ray blocks of one edge: whitelisted inliers that no track block took and whose cells the tracks leave uncovered
    void two_ray_blocks(const MeasurementGraph::Edge &edge, const pose_ref &s, const pose_ref &d, const std::vector<uint8_t> &keep,
                        double frac, edge_blocks &out) const
    {
        const CameraModel &sm = model_of(edge.source), &dm = model_of(edge.dest);
        const v3 so{s.loc[0], s.loc[1], s.loc[2]}, d_o{d.loc[0], d.loc[1], d.loc[2]};
        const size_t si = _graph.nodeIndex(edge.source), di = _graph.nodeIndex(edge.dest);
        static const std::unordered_map<uint64_t, char> none;
        const auto &s_meas = _track_measurement.empty() ? none : _track_measurement[si];
        const auto &d_meas = _track_measurement.empty() ? none : _track_measurement[di];
        const auto &s_cells = _covered.empty() ? none : _covered[si];
        const auto &d_cells = _covered.empty() ? none : _covered[di];
        TriangleWalker walker;
        if (!walker.init(_mesh))
            return;
        const auto &inl = edge.payload.inlier_matches;
        for (size_t idx = 0; idx < inl.size(); idx++)
        {
            if (idx >= keep.size() || keep[idx] == 0)
                continue;
            const feature_match_denormalized &m = inl[idx];
            if ((!s_meas.empty() && s_meas.count((uint64_t)m.feature_index_1)) || (!d_meas.empty() && d_meas.count((uint64_t)m.feature_index_2)))
                continue;
            const bool s_cov = !s_cells.empty() &&
                               s_cells.count(cell_key((int)std::floor((m.pixel_1[0] / (double)sm.pixels_cols) / frac),
                                                      (int)std::floor((m.pixel_1[1] / (double)sm.pixels_rows) / frac)));
            const bool d_cov = !d_cells.empty() &&
                               d_cells.count(cell_key((int)std::floor((m.pixel_2[0] / (double)dm.pixels_cols) / frac),
                                                      (int)std::floor((m.pixel_2[1] / (double)dm.pixels_rows) / frac)));
            if (s_cov && d_cov)
                continue;
            double r1[3], r2[3];
            image_to_3d(m.pixel_1, sm, r1);
            image_to_3d(m.pixel_2, dm, r2);
            v3 mid;
            double gap;
            ray_intersection(rotate(s.rot, v3{r1[0], r1[1], r1[2]}), so, rotate(d.rot, v3{r2[0], r2[1], r2[2]}), d_o, &mid, &gap);
            const double mean_z = (s.loc[2] + d.loc[2]) * 0.5;
            if (walker.find(v3{0, 0, -1}, v3{mid.x, mid.y, mean_z}) != TriangleWalker::INTERSECTION)
                continue;
            out.cams.push_back(s.cam);
            out.cams.push_back(d.cam);
            out.rays.insert(out.rays.end(), r1, r1 + 3);
            out.rays.insert(out.rays.end(), r2, r2 + 3);
            out.px.insert(out.px.end(), m.pixel_1, m.pixel_1 + 2);
            out.px.insert(out.px.end(), m.pixel_2, m.pixel_2 + 2);
            if (_intrinsics && same_model(sm, dm))
                out.intr_model = sm.id;
            for (int i = 0; i < 3; i++)
                out.tri.push_back((uint32_t)walker.tri[i]);
        }
    }

    ochip_ctx *_ctx;
    const MeasurementGraph &_graph;
    std::vector<NodePose> *_poses = nullptr;
    std::vector<std::pair<size_t, CameraModel>> *_cam_models = nullptr;
    std::unordered_map<size_t, size_t> _opt_index;
    std::unordered_map<size_t, uint32_t> _cam_of_node;
    std::vector<uint32_t> _pose_cam; // per pose: its camera, or UINT32_MAX for a repeated node
    std::vector<double> _cam_pos, _cam_q, _ray_dir, _ray_px;
    std::vector<uint8_t> _blk_intr;
    std::vector<twin> _twins;
    uint32_t _options = 0;
    bool _intrinsics = false, _several_models = false;
    std::atomic<bool> _views_without_features{false};
    size_t _shared_model = (size_t)-1, _mono_count = 0;
    double _mono_focal = 1;
    std::vector<uint8_t> _cam_opt, _blk_n;
    std::vector<uint32_t> _blk_ray_off{0}, _ray_cam, _blk_tri;
    std::vector<std::unordered_map<uint64_t, char>> _track_measurement, _covered; // per node index: features / cells in track blocks
    std::vector<cloud_track> _tracks_for_cloud;
    MeshGraph _mesh;
    ochip_relaxg_problem *_dev = nullptr;
};

} // namespace

bool relax(ochip_ctx *ctx, const MeasurementGraph &graph, std::vector<NodePose> &nodes,
           std::vector<std::pair<size_t, CameraModel>> &cam_models, const std::vector<size_t> &edges_to_optimize,
           const RelaxConfig &config, const std::vector<surface_model> &previous, surface_model *surface, RelaxTimers *timers,
           RelaxMeshStats *stats, std::string *error, const RelaxShard *shard)
{
    if (config.options & OPT_GROUND_MESH) // runGroundMesh (relax.cpp:89-102)
    {
        auto t0 = clk::now();
        GroundMeshProblem rp(ctx, graph);
        if (!rp.setup(nodes, cam_models, edges_to_optimize, config, previous, stats, error, shard))
            return false;
        if (timers)
            timers->setup_host += since(t0);
        t0 = clk::now();
        // (the problem writes the poses only after a successful solve; on failure they keep the caller's values)
        std::vector<NodePose> backup = nodes;
        const bool ok = rp.relax_observed_model_only(timers, stats, error) && rp.solve(timers, stats, error);
        if (timers)
            timers->device += since(t0);
        if (!ok)
        {
            nodes = backup;
            return false;
        }
        if (surface)
            rp.surface(surface);
        return true;
    }
    if (config.options & OPT_POINTS_3D) // runPoints (relax.cpp:103-115)
        return relax_points(ctx, graph, nodes, cam_models, edges_to_optimize, config.options, surface, timers, error);
    if (config.options & OPT_GROUND_PLANE) // runGroundPlane (relax.cpp:44-87)
    {
        surface_model_plane plane;
        if (!relax_ground_plane(ctx, graph, nodes, edges_to_optimize, &plane, timers, error, shard))
            return false;
        if (surface)
        {
            *surface = surface_model();
            size_t v[3];
            for (int i = 0; i < 3; i++)
                v[i] = surface->mesh.addNode(plane.corner[i][0], plane.corner[i][1], plane.corner[i][2]);
            for (int i = 0; i < 3; i++)
            {
                MeshEdge e;
                e.border = true;
                e.triangleOppositeNodes[0] = v[(i + 2) % 3];
                surface->mesh.addEdge(e, v[i], v[(i + 1) % 3]);
            }
        }
        return true;
    }
    // runRelativeOrientation (relax.cpp:14-42); its surface model is empty
    if (surface)
        *surface = surface_model();
    return relax_relative_orientation(ctx, graph, nodes, edges_to_optimize, timers, error);
}

} // namespace opencalibration_amd
