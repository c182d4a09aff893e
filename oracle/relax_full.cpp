// ORACLE — test infrastructure only (see oracle.hpp, relax_full.hpp for the list of restated sources).
#include "relax_full.hpp"
#include "relax_functors.hpp"

#include <algorithm>
#include <cstring>

namespace oracle
{
namespace rx
{

namespace
{

inline bool finite3(const Vec3 &v)
{
    return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z);
}
inline bool finiteq(const Quat &q)
{
    return std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z) && std::isfinite(q.w);
}
inline bool hasnanq(const Quat &q)
{
    return std::isnan(q.x) || std::isnan(q.y) || std::isnan(q.z) || std::isnan(q.w);
}
inline bool hasnan3(const Vec3 &v)
{
    return std::isnan(v.x) || std::isnan(v.y) || std::isnan(v.z);
}

Mat3 quat_to_matrix(const Quat &q) // Eigen Quaternion::toRotationMatrix()
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
Vec3 quat_rotate_d(const Quat &q, const Vec3 &v)
{
    const double qq[4] = {q.x, q.y, q.z, q.w};
    const V3<double> r = quat_rotate<double>(qq, V3<double>{v.x, v.y, v.z});
    return Vec3{r.x, r.y, r.z};
}

// src/geometry/intersection.cpp:116-143
std::pair<Vec3, double> rayIntersection(const Vec3 &d1, const Vec3 &o1, const Vec3 &d2, const Vec3 &o2)
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

inline uint64_t gridCellKey(int i, int j) // grid_filter.hpp:11-14
{
    return (static_cast<uint64_t>(i) << 32) | static_cast<uint32_t>(j);
}

// include/opencalibration/relax/grid_filter.hpp:16-62.  _best is a set of VALUES: a value displaced from one cell
// leaves the set even if it is still the best of another cell (kept as in the reference).
template <typename T> class GridFilter
{
  public:
    void setResolution(double r)
    {
        if (_map.empty())
            _res = r;
    }
    void addMeasurement(double x, double y, double score, const T &value)
    {
        const uint64_t index = gridCellKey((int)std::floor(x / _res), (int)std::floor(y / _res));
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
    const std::unordered_set<T> &getBestMeasurementsPerCell() const
    {
        return _best;
    }

  private:
    double _res = 0.075;
    std::unordered_map<uint64_t, std::pair<double, T>> _map;
    std::unordered_set<T> _best;
};

class UnionFind // types/union_find.hpp
{
  public:
    explicit UnionFind(size_t n) : _parent(n), _rank(n, 0)
    {
        for (size_t i = 0; i < n; i++)
            _parent[i] = i;
    }
    size_t find(size_t x)
    {
        // (iterative form of the reference's recursive path compression: same final parents)
        size_t root = x;
        while (_parent[root] != root)
            root = _parent[root];
        while (_parent[x] != root)
        {
            const size_t next = _parent[x];
            _parent[x] = root;
            x = next;
        }
        return root;
    }
    void unite(size_t a, size_t b)
    {
        a = find(a);
        b = find(b);
        if (a == b)
            return;
        if (_rank[a] < _rank[b])
            std::swap(a, b);
        _parent[b] = a;
        if (_rank[a] == _rank[b])
            _rank[a]++;
    }

  private:
    std::vector<size_t> _parent, _rank;
};

struct NodeIdFeatureIndex // types/feature_track.hpp:9-28
{
    size_t node_id, feature_index;
    bool operator==(const NodeIdFeatureIndex &o) const
    {
        return node_id == o.node_id && feature_index == o.feature_index;
    }
};
struct nifi_hash
{
    size_t operator()(const NodeIdFeatureIndex &n) const
    {
        return std::hash<size_t>()(n.node_id * 0x9E3779B97F4A7C15ull + n.feature_index);
    }
};
struct FeatureTrack
{
    Vec3 point{NAN, NAN, NAN};
    double error = NAN;
    std::vector<NodeIdFeatureIndex> measurements;
};

inline bool anticlockwise(const Vec3 &p0, const Vec3 &p1, const Vec3 &p2) // geometry/utils.hpp:10-14
{
    const Vec3 a = p1 - p0, b = p2 - p0;
    const double crossZ = a.x * b.y - a.y * b.x;
    return crossZ < 0;
}

// double instances of the geometry templates (intersection.hpp:26-47)
struct plane_no
{
    Vec3 norm, offset;
};
plane_no cornerPlane2normOffsetPlane_d(const Vec3 c[3])
{
    plane_no out;
    out.offset = c[0];
    out.norm = normalized(cross(c[0] - c[1], c[0] - c[2]));
    return out;
}
bool rayPlaneIntersection_d(const Vec3 &dir, const Vec3 &offset, const plane_no &p, Vec3 &out)
{
    const double denom = dot(p.norm, dir);
    if (std::abs(denom) < 1e-9)
    {
        out = Vec3{NAN, NAN, NAN};
        return false;
    }
    const double t = (dot(p.norm, p.offset) - dot(offset, p.norm)) / denom;
    out = offset + dir * t;
    return true;
}

} // namespace

// ------------------------------------------------------------------------------------------------- intersect.cpp
bool MeshIntersectionSearcher::init(const MeshGraph &meshGraph)
{
    _meshGraph = &meshGraph;
    _info = IntersectionInfo();
    if (meshGraph.size_nodes() == 0 || meshGraph.size_edges() == 0)
    {
        _meshGraph = nullptr;
        return false;
    }
    // the default IntersectionInfo has three equal node indexes: start from the first edge (intersect.cpp:31-37)
    const mesh_edge &edge = meshGraph.edges[0];
    _info.nodeIndexes[0] = edge.source;
    _info.nodeIndexes[1] = edge.dest;
    _info.nodeIndexes[2] = edge.opposite[0];
    for (size_t i = 0; i < 3; i++)
    {
        if (_info.nodeIndexes[i] >= meshGraph.nodes.size())
            return false;
        _info.nodeLocations[i] = &meshGraph.nodes[_info.nodeIndexes[i]].location;
    }
    return true;
}

const MeshIntersectionSearcher::IntersectionInfo &MeshIntersectionSearcher::triangleIntersect(const Vec3 &dir,
                                                                                              const Vec3 &offset)
{
    if (_meshGraph == nullptr)
    {
        _info.type = UNINITIALIZED;
        return _info;
    }
    Vec3 corner[3];
    for (size_t i = 0; i < 3; i++)
        corner[i] = *_info.nodeLocations[i];
    _info.type = PENDING;
    _info.steps = 0;
    constexpr size_t MAX_WALK_STEPS = 100;
    while (true)
    {
        if (anticlockwise(corner[0], corner[1], corner[2]))
        {
            std::swap(corner[0], corner[1]);
            std::swap(_info.nodeIndexes[0], _info.nodeIndexes[1]);
            std::swap(_info.nodeLocations[0], _info.nodeLocations[1]);
        }
        _info.intersectionLocation = Vec3{NAN, NAN, NAN};
        if (!rayPlaneIntersection_d(dir, offset, cornerPlane2normOffsetPlane_d(corner), _info.intersectionLocation) ||
            hasnan3(_info.intersectionLocation))
        {
            _info.type = RAY_PARALLEL_TO_PLANE;
            break;
        }
        size_t edgeIndex = 3;
        for (int i = 0; i < 3; i++)
            if (anticlockwise(_info.intersectionLocation, corner[i], corner[(i + 1) % 3]))
            {
                edgeIndex = i;
                break;
            }
        if (edgeIndex == 3)
        {
            _info.type = INTERSECTION;
            break;
        }
        const size_t keep0 = _info.nodeIndexes[edgeIndex], keep1 = _info.nodeIndexes[(edgeIndex + 1) % 3];
        const mesh_edge *edge = _meshGraph->getEdge(keep0, keep1);
        if (edge == nullptr)
            edge = _meshGraph->getEdge(keep1, keep0);
        if (edge == nullptr)
        {
            _info.type = GRAPH_STRUCTURE_INCONSISTENT;
            break;
        }
        if (edge->border)
        {
            _info.type = OUTSIDE_BORDER;
            break;
        }
        const size_t replacedNode = (edgeIndex + 2) % 3;
        if (edge->opposite[0] == _info.nodeIndexes[replacedNode])
            _info.nodeIndexes[replacedNode] = edge->opposite[1];
        else if (edge->opposite[1] == _info.nodeIndexes[replacedNode])
            _info.nodeIndexes[replacedNode] = edge->opposite[0];
        else
        {
            _info.type = GRAPH_STRUCTURE_INCONSISTENT;
            break;
        }
        if (_info.nodeIndexes[replacedNode] >= _meshGraph->nodes.size())
        {
            _info.type = GRAPH_STRUCTURE_INCONSISTENT; // (the reference would dereference a null node here)
            break;
        }
        const mesh_node &node = _meshGraph->nodes[_info.nodeIndexes[replacedNode]];
        corner[replacedNode] = node.location;
        _info.nodeLocations[replacedNode] = &node.location;
        _info.steps++;
        if (_info.steps > MAX_WALK_STEPS)
        {
            _info.type = INTERSECTION;
            break;
        }
    }
    return _info;
}

// ----------------------------------------------------------------------------------------------- expand_mesh.cpp
namespace
{
// jk::tree::KDTree<double, 2> restricted to what expand_mesh.cpp asks of it: nearest payload and the squared
// distance to the k-th nearest.  Exhaustive search; among exactly equidistant points the first inserted wins
// (the reference's tree order among exact ties depends on its splits).
struct point_index
{
    std::vector<std::array<double, 2>> xy;
    std::vector<double> payload;
    void addPoint(double x, double y, double p)
    {
        xy.push_back({x, y});
        payload.push_back(p);
    }
    size_t size() const
    {
        return xy.size();
    }
    double nearest_payload(double x, double y) const
    {
        double best = std::numeric_limits<double>::infinity();
        size_t bi = 0;
        for (size_t i = 0; i < xy.size(); i++)
        {
            const double dx = xy[i][0] - x, dy = xy[i][1] - y;
            const double d = dx * dx + dy * dy;
            if (d < best)
            {
                best = d;
                bi = i;
            }
        }
        return payload[bi];
    }
    double kth_distance(double x, double y, size_t k) const // squared distance of the k-th nearest (k >= 1)
    {
        std::vector<double> d(xy.size());
        for (size_t i = 0; i < xy.size(); i++)
        {
            const double dx = xy[i][0] - x, dy = xy[i][1] - y;
            d[i] = dx * dx + dy * dy;
        }
        k = std::min(k, d.size());
        std::nth_element(d.begin(), d.begin() + (k - 1), d.end());
        return d[k - 1];
    }
};

struct mesh_setup // the part rebuildMesh and buildMinimalMesh share (expand_mesh.cpp:35-121, :252-303)
{
    point_index vertexTree, cameraTree;
    double cameraMin[2], cameraMax[2];
    std::vector<double> heights, nearestCameraDistances;
};
void collect(const point_cloud &cameraLocations, const std::vector<surface_model> &previousSurfaces, mesh_setup &s)
{
    s.cameraMin[0] = s.cameraMin[1] = std::numeric_limits<double>::max();
    s.cameraMax[0] = s.cameraMax[1] = -std::numeric_limits<double>::max();
    for (const auto &surface : previousSurfaces)
    {
        for (const auto &n : surface.mesh.nodes)
            s.vertexTree.addPoint(n.location.x, n.location.y, n.location.z);
        for (const auto &cloud : surface.cloud)
            for (const auto &p : cloud)
                s.vertexTree.addPoint(p.x, p.y, p.z);
    }
    s.heights.reserve(cameraLocations.size());
    for (const auto &p : cameraLocations)
    {
        s.cameraMin[0] = std::min(s.cameraMin[0], p.x);
        s.cameraMin[1] = std::min(s.cameraMin[1], p.y);
        s.cameraMax[0] = std::max(s.cameraMax[0], p.x);
        s.cameraMax[1] = std::max(s.cameraMax[1], p.y);
        s.cameraTree.addPoint(p.x, p.y, p.z);
        if (s.vertexTree.size() > 0)
        {
            const double agl = p.z - s.vertexTree.nearest_payload(p.x, p.y);
            if (agl > -500 && agl < 5000)
                s.heights.push_back(agl);
        }
    }
    s.nearestCameraDistances.reserve(cameraLocations.size());
    for (const auto &p : cameraLocations)
        s.nearestCameraDistances.push_back(s.cameraTree.kth_distance(p.x, p.y, 2));
    std::sort(s.nearestCameraDistances.begin(), s.nearestCameraDistances.end());
}
} // namespace

MeshGraph rebuildMesh(const point_cloud &cameraLocations, const std::vector<surface_model> &previousSurfaces)
{
    bool hasPreviousData = false;
    for (const auto &s : previousSurfaces)
        if (s.mesh.size_nodes() > 0 || !s.cloud.empty())
        {
            hasPreviousData = true;
            break;
        }
    if (cameraLocations.size() < 2 && !hasPreviousData)
        return MeshGraph();
    constexpr double HEIGHT_MARGIN = 2;
    mesh_setup s;
    collect(cameraLocations, previousSurfaces, s);
    double gridDistance = s.nearestCameraDistances.size() < 2
                              ? std::numeric_limits<double>::infinity()
                              : std::sqrt(s.nearestCameraDistances[s.nearestCameraDistances.size() / 2]);
    const double ex = s.cameraMax[0] - s.cameraMin[0], ey = s.cameraMax[1] - s.cameraMin[1];
    const double minGridDistance = std::sqrt(ex * ex + ey * ey) / 1000.0;
    if (gridDistance < minGridDistance)
        gridDistance = std::max(1e-3, minGridDistance);
    if (s.heights.size() == 0)
        s.heights.push_back(std::isfinite(gridDistance) ? gridDistance : 10.0);
    std::sort(s.heights.begin(), s.heights.end());
    const double medianHeight = s.heights[s.heights.size() / 2];
    const double minBorderWidth = std::max(0.0, std::min(1000.0, medianHeight * HEIGHT_MARGIN));

    MeshGraph newGraph;
    size_t rows = static_cast<size_t>(std::ceil(std::max(0., ey + 2 * minBorderWidth) / gridDistance)) + 1;
    size_t cols = static_cast<size_t>(std::ceil(std::max(0., ex + 2 * minBorderWidth) / gridDistance)) + 1;
    if (rows > 1000 || cols > 1000)
    {
        rows = std::min<size_t>(rows, 1000);
        cols = std::min<size_t>(cols, 1000);
    }
    std::vector<size_t> grid(rows * cols);
    auto G = [&](size_t row, size_t col) -> size_t & { return grid[row * cols + col]; };
    for (size_t col = 0; col < cols; col++)
    {
        const double x = s.cameraMin[0] - minBorderWidth + gridDistance * col;
        for (size_t row = 0; row < rows; row++)
        {
            const double y = s.cameraMin[1] - minBorderWidth + gridDistance * row;
            const double z = s.vertexTree.size() > 0 ? s.vertexTree.nearest_payload(x, y)
                                                     : s.cameraTree.nearest_payload(x, y) - medianHeight;
            const size_t nodeId = newGraph.addNode(Vec3{x, y, z});
            G(row, col) = nodeId;
            if (row > 0)
            {
                mesh_edge e;
                e.border = col == 0 || col + 1 == cols;
                newGraph.addEdge(e, nodeId, G(row - 1, col));
            }
            if (col > 0)
            {
                mesh_edge e;
                e.border = row == 0 || row + 1 == rows;
                newGraph.addEdge(e, nodeId, G(row, col - 1));
            }
            if (row > 0 && col > 0)
            {
                mesh_edge e;
                e.border = false;
                newGraph.addEdge(e, nodeId, G(row - 1, col - 1));
            }
        }
    }
    for (size_t col = 0; col < cols; col++)
        for (size_t row = 0; row < rows; row++)
        {
            if (row > 0)
            {
                mesh_edge *edge = newGraph.getEdge(G(row, col), G(row - 1, col));
                if (col > 0)
                    edge->opposite[0] = G(row - 1, col - 1);
                if (col + 1 < cols)
                {
                    edge->opposite[1] = G(row, col + 1);
                    if (edge->border)
                        std::swap(edge->opposite[0], edge->opposite[1]);
                }
            }
            if (col > 0)
            {
                mesh_edge *edge = newGraph.getEdge(G(row, col), G(row, col - 1));
                if (row > 0)
                    edge->opposite[0] = G(row - 1, col - 1);
                if (row + 1 < rows)
                {
                    edge->opposite[1] = G(row + 1, col);
                    if (edge->border)
                        std::swap(edge->opposite[0], edge->opposite[1]);
                }
            }
            if (row > 0 && col > 0)
            {
                mesh_edge *edge = newGraph.getEdge(G(row, col), G(row - 1, col - 1));
                edge->opposite[0] = G(row, col - 1);
                edge->opposite[1] = G(row - 1, col);
            }
        }
    return newGraph;
}

MeshGraph buildMinimalMesh(const point_cloud &cameraLocations, const std::vector<surface_model> &previousSurfaces)
{
    if (cameraLocations.size() < 2)
        return MeshGraph();
    constexpr double HEIGHT_MARGIN = 2;
    mesh_setup s;
    collect(cameraLocations, previousSurfaces, s);
    const double gridDistance = s.nearestCameraDistances.size() < 2
                                    ? 10.0
                                    : std::sqrt(s.nearestCameraDistances[s.nearestCameraDistances.size() / 2]);
    if (s.heights.size() == 0)
        s.heights.push_back(std::isfinite(gridDistance) ? gridDistance : 10.0);
    std::sort(s.heights.begin(), s.heights.end());
    const double medianHeight = s.heights[s.heights.size() / 2];
    const double minBorderWidth = std::max(0.0, std::min(1000.0, medianHeight * HEIGHT_MARGIN));
    const double xMin = s.cameraMin[0] - minBorderWidth, xMax = s.cameraMax[0] + minBorderWidth;
    const double yMin = s.cameraMin[1] - minBorderWidth, yMax = s.cameraMax[1] + minBorderWidth;
    auto getZ = [&](double x, double y) -> double {
        if (s.vertexTree.size() > 0)
            return s.vertexTree.nearest_payload(x, y);
        return s.cameraTree.nearest_payload(x, y) - medianHeight;
    };
    MeshGraph mesh;
    const size_t v0 = mesh.addNode(Vec3{xMin, yMin, getZ(xMin, yMin)});
    const size_t v1 = mesh.addNode(Vec3{xMax, yMin, getZ(xMax, yMin)});
    const size_t v2 = mesh.addNode(Vec3{xMin, yMax, getZ(xMin, yMax)});
    const size_t v3 = mesh.addNode(Vec3{xMax, yMax, getZ(xMax, yMax)});
    auto border = [](size_t opp) {
        mesh_edge e;
        e.border = true;
        e.opposite[0] = opp;
        return e;
    };
    mesh.addEdge(border(v3), v0, v1); // bottom
    mesh.addEdge(border(v0), v1, v3); // right
    mesh.addEdge(border(v0), v2, v3); // top
    mesh.addEdge(border(v3), v0, v2); // left
    mesh_edge diag;
    diag.border = false;
    diag.opposite[0] = v1;
    diag.opposite[1] = v2;
    mesh.addEdge(diag, v0, v3);
    return mesh;
}

// ------------------------------------------------------------------------------------------- invert_distortion.cpp
namespace
{
inline bool all_zero_distortion(const camera_model &m)
{
    return m.radial_distortion[0] == 0 && m.radial_distortion[1] == 0 && m.radial_distortion[2] == 0 &&
           m.tangential_distortion[0] == 0 && m.tangential_distortion[1] == 0;
}

// ceres::TinySolver [3P, ceres/tiny_solver.h] for N parameters and the options intersection.cpp:175-180 sets.  F(params,
// residuals, jacobian row-major m x N).  Returns the final cost.
struct tiny_options
{
    double gradient_tolerance = 1e-10, parameter_tolerance = 1e-8, function_tolerance = 1e-6;
    double cost_threshold = std::numeric_limits<double>::epsilon(), initial_trust_region_radius = 1e4;
    int max_num_iterations = 50;
};
template <int N, typename F> double tiny_solve_n(F &&func, int m, double *x, const tiny_options &o)
{
    std::vector<double> r(m), J((size_t)m * N), fn(m);
    double jac_scale[N], jtj[N][N], g[N], cost = 0, gmax = 0;
    int iterations = 0;
    auto update = [&](const double *xx) {
        func(xx, r.data(), J.data());
        for (double &v : r)
            v = -v;
        if (iterations == 0)
            for (int c = 0; c < N; c++)
            {
                double s = 0;
                for (int i = 0; i < m; i++)
                    s += J[(size_t)i * N + c] * J[(size_t)i * N + c];
                jac_scale[c] = 1.0 / (1.0 + std::sqrt(s));
            }
        for (int i = 0; i < m; i++)
            for (int c = 0; c < N; c++)
                J[(size_t)i * N + c] *= jac_scale[c];
        for (int a = 0; a < N; a++)
        {
            for (int b = 0; b < N; b++)
            {
                double s = 0;
                for (int i = 0; i < m; i++)
                    s += J[(size_t)i * N + a] * J[(size_t)i * N + b];
                jtj[a][b] = s;
            }
            double s = 0;
            for (int i = 0; i < m; i++)
                s += J[(size_t)i * N + a] * r[i];
            g[a] = s;
        }
        gmax = 0;
        for (int a = 0; a < N; a++)
            gmax = std::max(gmax, std::abs(g[a]));
        double s = 0;
        for (int i = 0; i < m; i++)
            s += r[i] * r[i];
        cost = s / 2;
    };
    update(x);
    if (gmax < o.gradient_tolerance || cost < o.cost_threshold)
        return cost;
    double u = 1.0 / o.initial_trust_region_radius, v = 2;
    for (iterations = 1; iterations < o.max_num_iterations; iterations++)
    {
        double A[N][N], step[N];
        for (int a = 0; a < N; a++)
            for (int b = 0; b < N; b++)
                A[a][b] = jtj[a][b];
        for (int i = 0; i < N; i++)
        {
            const double d = std::sqrt(u * std::min(std::max(jtj[i][i], 1e-6), 1e32));
            A[i][i] += d * d;
        }
        {
            double L[N][N] = {}, D[N];
            bool ok = true;
            for (int j = 0; j < N && ok; j++)
            {
                double d = A[j][j];
                for (int k = 0; k < j; k++)
                    d -= L[j][k] * L[j][k] * D[k];
                D[j] = d;
                ok = d != 0 && std::isfinite(d);
                L[j][j] = 1;
                for (int i = j + 1; i < N && ok; i++)
                {
                    double e = A[i][j];
                    for (int k = 0; k < j; k++)
                        e -= L[i][k] * L[j][k] * D[k];
                    L[i][j] = e / d;
                }
            }
            double y[N];
            for (int i = 0; i < N; i++)
            {
                double e = g[i];
                for (int k = 0; k < i; k++)
                    e -= L[i][k] * y[k];
                y[i] = e;
            }
            for (int i = 0; i < N; i++)
                y[i] /= D[i];
            for (int i = N - 1; i >= 0; i--)
            {
                double e = y[i];
                for (int k = i + 1; k < N; k++)
                    e -= L[k][i] * step[k];
                step[i] = e;
            }
        }
        double dx[N], xn[N], dxn = 0, xnorm = 0;
        for (int i = 0; i < N; i++)
        {
            dx[i] = jac_scale[i] * step[i];
            dxn += dx[i] * dx[i];
            xnorm += x[i] * x[i];
        }
        if (std::sqrt(dxn) < o.parameter_tolerance * (std::sqrt(xnorm) + o.parameter_tolerance))
            break;
        for (int i = 0; i < N; i++)
            xn[i] = x[i] + dx[i];
        func(xn, fn.data(), nullptr);
        double cost_new = 0;
        for (int i = 0; i < m; i++)
            cost_new += fn[i] * fn[i];
        const double cost_change = (2 * cost - cost_new);
        double model_cost_change = 0; // lm_step' (2 g - jtj lm_step)
        for (int a = 0; a < N; a++)
        {
            double t = 2 * g[a];
            for (int b = 0; b < N; b++)
                t -= jtj[a][b] * step[b];
            model_cost_change += step[a] * t;
        }
        const double rho = cost_change / model_cost_change;
        if (rho > 0)
        {
            for (int i = 0; i < N; i++)
                x[i] = xn[i];
            if (std::abs(cost_change) < o.function_tolerance)
            {
                update(x);
                break;
            }
            double tmp = 2 * rho - 1;
            u = u * std::max(1 / 3., 1 - tmp * tmp * tmp);
            v = 2;
            update(x);
            if (gmax < o.gradient_tolerance || cost < o.cost_threshold)
                break;
        }
        else
        {
            u *= v;
            v *= 2;
        }
    }
    return cost;
}

// rayIntersection(model1, model2, pos1, pos2, rot1, rot2, px1, px2) (src/geometry/intersection.cpp:163-186): the midpoint
// of the two rays, refined by TinySolver on the four reprojection residuals; second = the solver's final cost
std::pair<Vec3, double> rayIntersectionRefined(const camera_model &model1, const camera_model &model2, const Vec3 &pos1, const Vec3 &pos2,
                                               const Quat &rot1, const Quat &rot2, const double px1[2], const double px2[2])
{
    std::pair<Vec3, double> guess = rayIntersection(quat_rotate_d(rot1, image_to_3d(px1, model1)), pos1,
                                                    quat_rotate_d(rot2, image_to_3d(px2, model2)), pos2);
    const camera_model *models[2] = {&model1, &model2};
    const Vec3 *pos[2] = {&pos1, &pos2};
    const Quat *rot[2] = {&rot1, &rot2};
    const double *px[2] = {px1, px2};
    auto func = [&](const double *params, double *res, double *jac) {
        for (int c = 0; c < 2; c++)
        {
            PixelErrorCost f{*pos[c], *models[c], {px[c][0], px[c][1]}};
            const double q[4] = {rot[c]->x, rot[c]->y, rot[c]->z, rot[c]->w};
            if (jac)
            {
                using J3 = Jet<3>;
                J3 p[3], qj[4], r[2];
                for (int i = 0; i < 3; i++)
                {
                    p[i] = J3(params[i]);
                    p[i].v[i] = 1;
                }
                for (int i = 0; i < 4; i++)
                    qj[i] = J3(q[i]);
                f.eval<J3>(qj, p, nullptr, nullptr, nullptr, nullptr, r);
                for (int i = 0; i < 2; i++)
                {
                    res[2 * c + i] = r[i].a;
                    for (int k = 0; k < 3; k++)
                        jac[(size_t)(2 * c + i) * 3 + k] = r[i].v[k];
                }
            }
            else
                f.eval<double>(q, params, nullptr, nullptr, nullptr, nullptr, res + 2 * c);
        }
    };
    tiny_options o;
    o.max_num_iterations = 50;
    o.cost_threshold = 1e-7;
    o.parameter_tolerance = 1e-14;
    o.gradient_tolerance = 1e-12;
    o.initial_trust_region_radius = 1e6;
    double x[3] = {guess.first.x, guess.first.y, guess.first.z};
    guess.second = tiny_solve_n<3>(func, 4, x, o);
    guess.first = Vec3{x[0], x[1], x[2]};
    return guess;
}

// ceres::TinySolver<TinySolverAutoDiffFunction<F, Dynamic, 5>> [3P, ceres/tiny_solver.h]: LM with Jacobi scaling from
// the first Jacobian, an LDLT solve of the regularised normal equations, Nielsen's damping update; default options
// (50 iterations, gradient 1e-10, parameter 1e-8, function 1e-6, cost threshold machine epsilon, radius 1e4).
// F(params, residuals, jacobian row-major m x 5) evaluates residuals and Jacobian.
template <typename F> void tiny_solve5(F &&func, int m, double x[5])
{
    constexpr int N = 5;
    const double gradient_tolerance = 1e-10, parameter_tolerance = 1e-8, function_tolerance = 1e-6;
    const double cost_threshold = std::numeric_limits<double>::epsilon();
    const double initial_trust_region_radius = 1e4;
    const int max_num_iterations = 50;
    std::vector<double> r(m), J((size_t)m * N), fn(m), Jn;
    double jac_scale[N], jtj[N][N], g[N], cost = 0, gmax = 0;
    int iterations = 0;
    auto update = [&](const double *xx) {
        func(xx, r.data(), J.data());
        for (double &v : r)
            v = -v; // residuals_ = -residuals_
        if (iterations == 0)
            for (int c = 0; c < N; c++)
            {
                double s = 0;
                for (int i = 0; i < m; i++)
                    s += J[(size_t)i * N + c] * J[(size_t)i * N + c];
                jac_scale[c] = 1.0 / (1.0 + std::sqrt(s));
            }
        for (int i = 0; i < m; i++)
            for (int c = 0; c < N; c++)
                J[(size_t)i * N + c] *= jac_scale[c];
        for (int a = 0; a < N; a++)
        {
            for (int b = 0; b < N; b++)
            {
                double s = 0;
                for (int i = 0; i < m; i++)
                    s += J[(size_t)i * N + a] * J[(size_t)i * N + b];
                jtj[a][b] = s;
            }
            double s = 0;
            for (int i = 0; i < m; i++)
                s += J[(size_t)i * N + a] * r[i];
            g[a] = s;
        }
        gmax = 0;
        for (int a = 0; a < N; a++)
            gmax = std::max(gmax, std::abs(g[a]));
        double s = 0;
        for (int i = 0; i < m; i++)
            s += r[i] * r[i];
        cost = s / 2;
    };
    update(x);
    if (gmax < gradient_tolerance || cost < cost_threshold)
        return;
    double u = 1.0 / initial_trust_region_radius, v = 2;
    for (iterations = 1; iterations < max_num_iterations; iterations++)
    {
        double A[N][N], step[N];
        for (int a = 0; a < N; a++)
            for (int b = 0; b < N; b++)
                A[a][b] = jtj[a][b];
        for (int i = 0; i < N; i++)
        {
            const double d = std::sqrt(u * std::min(std::max(jtj[i][i], 1e-6), 1e32));
            A[i][i] += d * d;
        }
        // symmetric positive definite solve (LDLT without pivoting; Eigen's LDLT pivots on the largest diagonal, the
        // solution agrees to rounding)
        {
            double L[N][N] = {}, D[N];
            for (int j = 0; j < N; j++)
            {
                double d = A[j][j];
                for (int k = 0; k < j; k++)
                    d -= L[j][k] * L[j][k] * D[k];
                D[j] = d;
                for (int i = j + 1; i < N; i++)
                {
                    double s = A[i][j];
                    for (int k = 0; k < j; k++)
                        s -= L[i][k] * L[j][k] * D[k];
                    L[i][j] = s / d;
                }
            }
            double y[N];
            for (int i = 0; i < N; i++)
            {
                double s = g[i];
                for (int k = 0; k < i; k++)
                    s -= L[i][k] * y[k];
                y[i] = s;
            }
            for (int i = N - 1; i >= 0; i--)
            {
                double s = y[i] / D[i];
                for (int k = i + 1; k < N; k++)
                    s -= L[k][i] * step[k];
                step[i] = s;
            }
        }
        double dx[N], xn[N], dxn = 0, xnorm = 0;
        for (int i = 0; i < N; i++)
        {
            dx[i] = jac_scale[i] * step[i];
            dxn += dx[i] * dx[i];
            xnorm += x[i] * x[i];
            xn[i] = x[i] + dx[i];
        }
        if (std::sqrt(dxn) < parameter_tolerance * (std::sqrt(xnorm) + parameter_tolerance))
            break;
        func(xn, fn.data(), nullptr);
        double fn2 = 0;
        for (int i = 0; i < m; i++)
            fn2 += fn[i] * fn[i];
        const double cost_change = 2 * cost - fn2;
        double model_cost_change = 0;
        for (int a = 0; a < N; a++)
        {
            double t = 2 * g[a];
            for (int b = 0; b < N; b++)
                t -= jtj[a][b] * step[b];
            model_cost_change += step[a] * t;
        }
        const double rho = cost_change / model_cost_change;
        if (rho > 0)
        {
            for (int i = 0; i < N; i++)
                x[i] = xn[i];
            if (std::abs(cost_change) < function_tolerance)
                break;
            update(x);
            if (gmax < gradient_tolerance || cost < cost_threshold)
                break;
            const double tmp = 2 * rho - 1;
            u = u * std::max(1 / 3., 1 - tmp * tmp * tmp);
            v = 2;
        }
        else
        {
            if (std::abs(cost_change) < function_tolerance)
                break;
            u *= v;
            v *= 2;
        }
    }
}
} // namespace

Vec3 image_to_3d_inverse_model(const double keypoint[2], const camera_model &m)
{
    inverse_model_t<double> im;
    im.focal_length_pixels = m.focal_length_pixels;
    im.principle_point[0] = m.principle_point[0], im.principle_point[1] = m.principle_point[1];
    for (int i = 0; i < 3; i++)
        im.radial_distortion[i] = m.radial_distortion[i];
    im.tangential_distortion[0] = m.tangential_distortion[0], im.tangential_distortion[1] = m.tangential_distortion[1];
    const V3<double> r = image_to_3d_inverse<double>(keypoint, im);
    return Vec3{r.x, r.y, r.z};
}

camera_model convertModelToInverse(const camera_model &standardModel) // invert_distortion.cpp:105-150
{
    camera_model inverted = standardModel;
    for (int i = 0; i < 3; i++)
        inverted.radial_distortion[i] *= -1;
    inverted.tangential_distortion[0] = inverted.tangential_distortion[1] = 0;
    std::vector<std::pair<Vec3, Vec2>> corr;
    constexpr int grid_divisions = 20;
    const size_t si = standardModel.pixels_cols / grid_divisions, sj = standardModel.pixels_rows / grid_divisions;
    if (si == 0 || sj == 0)
        return inverted; // (the reference would loop forever on an image narrower than 20 px)
    for (size_t i = 0; i < standardModel.pixels_cols; i += si)
        for (size_t j = 0; j < standardModel.pixels_rows; j += sj)
        {
            const double p[2] = {(double)i, (double)j};
            const Vec3 training_point = image_to_3d(p, standardModel);
            const Vec2 p2 = image_from_3d(training_point, standardModel);
            if (!hasnan3(training_point))
                corr.emplace_back(training_point, p2);
        }
    const int m = (int)corr.size() * 3;
    auto func = [&](const double *params, double *res, double *jac) {
        using J5 = Jet<5>;
        for (size_t c = 0; c < corr.size(); c++)
        {
            if (jac)
            {
                inverse_model_t<J5> im;
                im.focal_length_pixels = J5(inverted.focal_length_pixels);
                im.principle_point[0] = J5(inverted.principle_point[0]), im.principle_point[1] = J5(inverted.principle_point[1]);
                for (int i = 0; i < 3; i++)
                    im.radial_distortion[i] = J5(params[i], i);
                for (int i = 0; i < 2; i++)
                    im.tangential_distortion[i] = J5(params[3 + i], 3 + i);
                const J5 px[2] = {J5(corr[c].second.x), J5(corr[c].second.y)};
                const V3<J5> r = image_to_3d_inverse<J5>(px, im);
                const J5 e[3] = {r.x - J5(corr[c].first.x), r.y - J5(corr[c].first.y), r.z - J5(corr[c].first.z)};
                for (int k = 0; k < 3; k++)
                {
                    res[3 * c + k] = e[k].a;
                    for (int q = 0; q < 5; q++)
                        jac[(3 * c + k) * 5 + q] = e[k].v[q];
                }
            }
            else
            {
                camera_model t = inverted;
                for (int i = 0; i < 3; i++)
                    t.radial_distortion[i] = params[i];
                t.tangential_distortion[0] = params[3], t.tangential_distortion[1] = params[4];
                const double px[2] = {corr[c].second.x, corr[c].second.y};
                const Vec3 r = image_to_3d_inverse_model(px, t);
                res[3 * c] = r.x - corr[c].first.x;
                res[3 * c + 1] = r.y - corr[c].first.y;
                res[3 * c + 2] = r.z - corr[c].first.z;
            }
        }
    };
    double params[5] = {0, 0, 0, 0, 0};
    if (m > 0)
        tiny_solve5(func, m, params);
    for (int i = 0; i < 3; i++)
        inverted.radial_distortion[i] = params[i];
    inverted.tangential_distortion[0] = params[3], inverted.tangential_distortion[1] = params[4];
    return inverted;
}

camera_model convertModelToForward(const camera_model &invertedModel) // invert_distortion.cpp:152-191
{
    camera_model standard = invertedModel;
    for (int i = 0; i < 3; i++)
        standard.radial_distortion[i] *= -1;
    standard.tangential_distortion[0] = standard.tangential_distortion[1] = 0;
    std::vector<std::pair<Vec3, Vec2>> corr;
    constexpr int grid_divisions = 20;
    const size_t si = invertedModel.pixels_cols / grid_divisions, sj = invertedModel.pixels_rows / grid_divisions;
    if (si == 0 || sj == 0)
        return standard;
    for (size_t i = 0; i < invertedModel.pixels_cols; i += si)
        for (size_t j = 0; j < invertedModel.pixels_rows; j += sj)
        {
            const double p[2] = {(double)i, (double)j};
            const Vec3 training_point = image_to_3d_inverse_model(p, invertedModel);
            if (!hasnan3(training_point))
                corr.emplace_back(training_point, Vec2{p[0], p[1]});
        }
    const int m = (int)corr.size() * 2;
    auto func = [&](const double *params, double *res, double *jac) {
        using J5 = Jet<5>;
        for (size_t c = 0; c < corr.size(); c++)
        {
            // image_from_3d<T> (distort_keypoints.hpp:44-66) with the distortion coefficients as the unknowns
            if (jac)
            {
                const Vec3 &ray = corr[c].first;
                const double cz = ray.z < 1e-3 ? 1e-3 : ray.z;
                const J5 rp[2] = {J5(ray.x / cz), J5(ray.y / cz)};
                J5 radial[3], tang[2], rd[2];
                for (int i = 0; i < 3; i++)
                    radial[i] = J5(params[i], i);
                for (int i = 0; i < 2; i++)
                    tang[i] = J5(params[3 + i], 3 + i);
                distortProjectedRayT<J5>(rp, radial, tang, rd);
                for (int k = 0; k < 2; k++)
                {
                    const J5 e = rd[k] * J5(standard.focal_length_pixels) + J5(standard.principle_point[k]) -
                                 J5(k == 0 ? corr[c].second.x : corr[c].second.y);
                    res[2 * c + k] = e.a;
                    for (int q = 0; q < 5; q++)
                        jac[(2 * c + k) * 5 + q] = e.v[q];
                }
            }
            else
            {
                camera_model t = standard;
                for (int i = 0; i < 3; i++)
                    t.radial_distortion[i] = params[i];
                t.tangential_distortion[0] = params[3], t.tangential_distortion[1] = params[4];
                const Vec2 px = image_from_3d(corr[c].first, t);
                res[2 * c] = px.x - corr[c].second.x;
                res[2 * c + 1] = px.y - corr[c].second.y;
            }
        }
    };
    double params[5] = {0, 0, 0, 0, 0};
    if (m > 0)
        tiny_solve5(func, m, params);
    for (int i = 0; i < 3; i++)
        standard.radial_distortion[i] = params[i];
    standard.tangential_distortion[0] = params[3], standard.tangential_distortion[1] = params[4];
    return standard;
}

// ------------------------------------------------------------------------------------------------ relax_problem.cpp
namespace
{

struct PoseOpt // relax_problem.hpp OptimizationPackage::PoseOpt
{
    const Vec3 *loc_ptr = nullptr;
    Quat *rot_ptr = nullptr;
    const CameraModel *model_ptr = nullptr;
    bool optimize = true;
    size_t node_id = 0;
};

struct RayInfo // relax_problem.cpp:564-574
{
    size_t node_id, feature_index, camera_model_id;
    Vec3 camera_loc, camera_ray;
    Vec2 pixel;
    Quat orientation;
    double *rot_ptr;
};

template <int N> mc::CostFunction *makeMultiRayCost(const std::vector<RayInfo> &good, const double corner2d[3][2])
{
    auto *f = new NRay<N>();
    for (int i = 0; i < N; i++)
    {
        const Vec3 &l = good[i].camera_loc, &r = good[i].camera_ray;
        f->impl.camera_loc[i][0] = l.x, f->impl.camera_loc[i][1] = l.y, f->impl.camera_loc[i][2] = l.z;
        f->impl.camera_ray[i][0] = r.x, f->impl.camera_ray[i][1] = r.y, f->impl.camera_ray[i][2] = r.z;
        f->impl.camera_pixel[i][0] = good[i].pixel.x, f->impl.camera_pixel[i][1] = good[i].pixel.y;
    }
    std::memcpy(f->impl.plane_point, corner2d, sizeof f->impl.plane_point);
    if (N == 3)
        return new mc::AutoDiffCostFunction<NRay<N>, 3 * N, 1, 1, 1, 4, 4, 4>(f);
    return nullptr;
}
template <> mc::CostFunction *makeMultiRayCost<4>(const std::vector<RayInfo> &good, const double corner2d[3][2])
{
    auto *f = new NRay<4>();
    for (int i = 0; i < 4; i++)
    {
        const Vec3 &l = good[i].camera_loc, &r = good[i].camera_ray;
        f->impl.camera_loc[i][0] = l.x, f->impl.camera_loc[i][1] = l.y, f->impl.camera_loc[i][2] = l.z;
        f->impl.camera_ray[i][0] = r.x, f->impl.camera_ray[i][1] = r.y, f->impl.camera_ray[i][2] = r.z;
        f->impl.camera_pixel[i][0] = good[i].pixel.x, f->impl.camera_pixel[i][1] = good[i].pixel.y;
    }
    std::memcpy(f->impl.plane_point, corner2d, sizeof f->impl.plane_point);
    return new mc::AutoDiffCostFunction<NRay<4>, 12, 1, 1, 1, 4, 4, 4, 4>(f);
}
template <> mc::CostFunction *makeMultiRayCost<5>(const std::vector<RayInfo> &good, const double corner2d[3][2])
{
    auto *f = new NRay<5>();
    for (int i = 0; i < 5; i++)
    {
        const Vec3 &l = good[i].camera_loc, &r = good[i].camera_ray;
        f->impl.camera_loc[i][0] = l.x, f->impl.camera_loc[i][1] = l.y, f->impl.camera_loc[i][2] = l.z;
        f->impl.camera_ray[i][0] = r.x, f->impl.camera_ray[i][1] = r.y, f->impl.camera_ray[i][2] = r.z;
        f->impl.camera_pixel[i][0] = good[i].pixel.x, f->impl.camera_pixel[i][1] = good[i].pixel.y;
    }
    std::memcpy(f->impl.plane_point, corner2d, sizeof f->impl.plane_point);
    return new mc::AutoDiffCostFunction<NRay<5>, 15, 1, 1, 1, 4, 4, 4, 4, 4>(f);
}

template <int N> void fill_focal_radial(NRayFocalRadial<N> *f, const std::vector<RayInfo> &good,
                                        const double corner2d[3][2], const camera_model &inv)
{
    for (int i = 0; i < N; i++)
    {
        const Vec3 &l = good[i].camera_loc;
        f->impl.camera_loc[i][0] = l.x, f->impl.camera_loc[i][1] = l.y, f->impl.camera_loc[i][2] = l.z;
        f->impl.camera_pixel[i][0] = good[i].pixel.x, f->impl.camera_pixel[i][1] = good[i].pixel.y;
        f->impl.camera_ray[i][0] = f->impl.camera_ray[i][1] = f->impl.camera_ray[i][2] = NAN;
    }
    std::memcpy(f->impl.plane_point, corner2d, sizeof f->impl.plane_point);
    f->impl.shared_tangential[0] = inv.tangential_distortion[0];
    f->impl.shared_tangential[1] = inv.tangential_distortion[1];
}
mc::CostFunction *makeMultiRayCostFocalRadial(int N, const std::vector<RayInfo> &good, const double corner2d[3][2],
                                              const camera_model &inv)
{
    if (N == 3)
    {
        auto *f = new NRayFocalRadial<3>();
        fill_focal_radial<3>(f, good, corner2d, inv);
        return new mc::AutoDiffCostFunction<NRayFocalRadial<3>, 9, 1, 1, 1, 1, 2, 3, 4, 4, 4>(f);
    }
    if (N == 4)
    {
        auto *f = new NRayFocalRadial<4>();
        fill_focal_radial<4>(f, good, corner2d, inv);
        return new mc::AutoDiffCostFunction<NRayFocalRadial<4>, 12, 1, 1, 1, 1, 2, 3, 4, 4, 4, 4>(f);
    }
    auto *f = new NRayFocalRadial<5>();
    fill_focal_radial<5>(f, good, corner2d, inv);
    return new mc::AutoDiffCostFunction<NRayFocalRadial<5>, 15, 1, 1, 1, 1, 2, 3, 4, 4, 4, 4, 4>(f);
}

struct InverseModel // InverseDifferentiableCameraModel<double>: the parameter blocks of the intrinsics flavours
{
    camera_model m;
};

class RelaxProblem
{
  public:
    explicit RelaxProblem(relax_stats *stats) : _stats(stats)
    {
        _opt.max_num_iterations = 100; // relax_problem.cpp:30-37
        _opt.initial_trust_region_radius = 1;
    }

    // :61-81
    void setupGroundPlaneProblem(const MeasurementGraph &graph, std::vector<NodePose> &nodes, model_map &cam_models,
                                 const std::vector<size_t> &edges_to_optimize, const RelaxOptionSet &options)
    {
        initialize(nodes, cam_models);
        initializeGroundPlane();
        _loss.reset(new mc::HuberLoss(1 * M_PI / 180));
        gridFilterMatchesPerImage(graph, edges_to_optimize, 0.15);
        for (size_t edge_id : edges_to_optimize)
        {
            const graph_edge *edge = graph.getEdge(edge_id);
            if (edge != nullptr && shouldAddEdgeToOptimization(edge_id))
                addRayTriangleMeasurementCost(graph, edge_id, *edge, options);
        }
        addDownwardsPrior();
    }

    // :83-120
    void setupGroundMeshProblem(const MeasurementGraph &graph, std::vector<NodePose> &nodes, model_map &cam_models,
                                const std::vector<size_t> &edges_to_optimize, const RelaxOptionSet &options,
                                const std::vector<surface_model> &previousSurfaces, double grid_fraction)
    {
        initialize(nodes, cam_models);
        initializeGroundMesh(previousSurfaces, options.get(MINIMAL_MESH));
        _loss.reset(new mc::HuberLoss(1 * M_PI / 180));
        for (size_t edge_id : edges_to_optimize)
        {
            const graph_edge *edge = graph.getEdge(edge_id);
            if (edge != nullptr && shouldAddEdgeToOptimization(edge_id))
                collectEdgeTracks(graph, edge_id, *edge);
        }
        addMultiRayTrackCosts(graph, options, grid_fraction);
        gridFilterMatchesPerImage(graph, edges_to_optimize, grid_fraction);
        for (size_t edge_id : edges_to_optimize)
        {
            const graph_edge *edge = graph.getEdge(edge_id);
            if (edge != nullptr && shouldAddEdgeToOptimization(edge_id))
                addRayTriangleMeasurementCost(graph, edge_id, *edge, options);
        }
        addMeshFlatPrior();
        addMeshSmoothPrior();
        addMonotonicityCosts();
    }

    // :40-59 (tests only in the reference: relax() with {ORIENTATION} alone)
    void setupDecompositionProblem(const MeasurementGraph &graph, std::vector<NodePose> &nodes, const std::vector<size_t> &edges_to_optimize)
    {
        _loss.reset(new mc::HuberLoss(10 * M_PI / 180));
        model_map cam_models;
        initialize(nodes, cam_models);
        _opt.initial_trust_region_radius = 0.1;
        for (size_t edge_id : edges_to_optimize)
        {
            const graph_edge *edge = graph.getEdge(edge_id);
            if (edge != nullptr && shouldAddEdgeToOptimization(edge_id))
                addRelationCost(graph, edge_id, *edge);
        }
        addDownwardsPrior();
    }

    // :122-145 (tests only in the reference: {ORIENTATION, POINTS_3D, ...})
    void setup3dPointProblem(const MeasurementGraph &graph, std::vector<NodePose> &nodes, model_map &cam_models,
                             const std::vector<size_t> &edges_to_optimize, const RelaxOptionSet &options)
    {
        initialize(nodes, cam_models);
        _loss.reset(new mc::HuberLoss(10));
        gridFilterMatchesPerImage(graph, edges_to_optimize, 0.05);
        for (size_t edge_id : edges_to_optimize)
        {
            const graph_edge *edge = graph.getEdge(edge_id);
            if (edge != nullptr && shouldAddEdgeToOptimization(edge_id))
                addPointMeasurementsCost(graph, edge_id, *edge, options);
        }
        addMonotonicityCosts();
        _opt.max_num_iterations = 1000; // (SPARSE_SCHUR: another exact solve of the same normal equations)
    }

    // :311-350
    void addRelationCost(const MeasurementGraph &graph, size_t edge_id, const graph_edge &edge)
    {
        if (edge.payload.inlier_matches.size() == 0)
            return;
        const PoseOpt src = nodeid2poseopt(graph, edge.source, false), dst = nodeid2poseopt(graph, edge.dest, false);
        if (src.loc_ptr == nullptr || dst.loc_ptr == nullptr)
            return;
        if (!finiteq(*src.rot_ptr) || !finiteq(*dst.rot_ptr) || !finite3(*src.loc_ptr) || !finite3(*dst.loc_ptr))
            return;
        double *datas[2] = {&src.rot_ptr->x, &dst.rot_ptr->x};
        _problem.AddResidualBlock(new mc::AutoDiffCostFunction<MultiDecomposedRotationCost, 3, 4, 4>(
                                      new MultiDecomposedRotationCost(edge.payload.relative_poses, *src.loc_ptr, *dst.loc_ptr)),
                                  _loss.get(), {datas[0], datas[1]});
        _problem.SetManifold(datas[0], mc::Manifold::EIGEN_QUATERNION);
        _problem.SetManifold(datas[1], mc::Manifold::EIGEN_QUATERNION);
        if (!src.optimize)
            _problem.SetParameterBlockConstant(datas[0]);
        if (!dst.optimize)
            _problem.SetParameterBlockConstant(datas[1]);
        _edges_used.insert(edge_id);
    }

    // :986-1187
    void addPointMeasurementsCost(const MeasurementGraph &graph, size_t edge_id, const graph_edge &edge, const RelaxOptionSet &options)
    {
        _edge_tracks.emplace_back(edge_id, std::vector<FeatureTrack>());
        auto &points = _edge_tracks.back().second;
        points.reserve(edge.payload.inlier_matches.size());
        const PoseOpt src = nodeid2poseopt(graph, edge.source), dst = nodeid2poseopt(graph, edge.dest);
        if (src.loc_ptr == nullptr || dst.loc_ptr == nullptr)
            return;
        if (options.hasAll(FOCAL_LENGTH) && (src.model_ptr == nullptr || dst.model_ptr == nullptr))
            return;
        CameraModel &source_model = *const_cast<CameraModel *>(src.model_ptr), &dest_model = *const_cast<CameraModel *>(dst.model_ptr);
        const auto &swl = _grid_filter[edge.source][edge_id].getBestMeasurementsPerCell();
        const auto &dwl = _grid_filter[edge.dest][edge_id].getBestMeasurementsPerCell();
        double *orientation_ptrs[2] = {&src.rot_ptr->x, &dst.rot_ptr->x};
        const Vec3 *locs[2] = {src.loc_ptr, dst.loc_ptr};
        CameraModel *models[2] = {&source_model, &dest_model};
        const bool tangential = options.hasAny(LENS_DISTORTIONS_TANGENTIAL) &&
                                options.hasAll(LENS_DISTORTIONS_RADIAL | FOCAL_LENGTH | ORIENTATION | POINTS_3D);
        const bool radial = !tangential && options.hasAny(LENS_DISTORTIONS_RADIAL) && options.hasAll(FOCAL_LENGTH | ORIENTATION | POINTS_3D);
        const bool focal = !tangential && !radial && options.hasAny(FOCAL_LENGTH | PRINCIPAL_POINT) && options.hasAll(ORIENTATION | POINTS_3D);
        const bool plain = !tangential && !radial && !focal && options.hasAll(ORIENTATION | POINTS_3D);
        if (!tangential && !radial && !focal && !plain)
            return; // "No viable bundle options found"
        bool points_added = false;
        for (const auto &inlier : edge.payload.inlier_matches)
        {
            if (swl.find(&inlier) == swl.end() && dwl.find(&inlier) == dwl.end())
                continue;
            const auto intersection = rayIntersectionRefined(source_model, dest_model, *src.loc_ptr, *dst.loc_ptr, *src.rot_ptr, *dst.rot_ptr,
                                                             inlier.pixel_1, inlier.pixel_2);
            FeatureTrack track;
            track.point = intersection.first;
            track.error = intersection.second;
            track.measurements = {NodeIdFeatureIndex{edge.source, inlier.feature_index_1}, NodeIdFeatureIndex{edge.dest, inlier.feature_index_2}};
            points.push_back(track);
            double *point = &points.back().point.x;
            mc::CostFunction *func[2];
            std::vector<double *> args[2];
            for (int i = 0; i < 2; i++)
            {
                const double *px = i == 0 ? inlier.pixel_1 : inlier.pixel_2;
                PixelErrorCost base{*locs[i], static_cast<const camera_model &>(*models[i]), {px[0], px[1]}};
                double *f = &models[i]->focal_length_pixels, *pp = models[i]->principle_point, *k = models[i]->radial_distortion,
                       *tg = models[i]->tangential_distortion;
                if (tangential)
                {
                    auto *fn = new PixelErrorCost_OrientationFocalRadialTangential();
                    static_cast<PixelErrorCost &>(*fn) = base;
                    func[i] = new mc::AutoDiffCostFunction<PixelErrorCost_OrientationFocalRadialTangential, 2, 4, 3, 1, 2, 3, 2>(fn);
                    args[i] = {orientation_ptrs[i], point, f, pp, k, tg};
                }
                else if (radial)
                {
                    auto *fn = new PixelErrorCost_OrientationFocalRadial();
                    static_cast<PixelErrorCost &>(*fn) = base;
                    func[i] = new mc::AutoDiffCostFunction<PixelErrorCost_OrientationFocalRadial, 2, 4, 3, 1, 2, 3>(fn);
                    args[i] = {orientation_ptrs[i], point, f, pp, k};
                }
                else if (focal)
                {
                    auto *fn = new PixelErrorCost_OrientationFocal();
                    static_cast<PixelErrorCost &>(*fn) = base;
                    func[i] = new mc::AutoDiffCostFunction<PixelErrorCost_OrientationFocal, 2, 4, 3, 1, 2>(fn);
                    args[i] = {orientation_ptrs[i], point, f, pp};
                }
                else
                {
                    auto *fn = new PixelErrorCost_Orientation();
                    static_cast<PixelErrorCost &>(*fn) = base;
                    func[i] = new mc::AutoDiffCostFunction<PixelErrorCost_Orientation, 2, 4, 3>(fn);
                    args[i] = {orientation_ptrs[i], point};
                }
            }
            bool all_finite = true;
            for (int i = 0; i < 2; i++)
            {
                double res[2] = {NAN, NAN};
                func[i]->Evaluate(args[i].data(), res, nullptr);
                if (!std::isfinite(res[0]) || !std::isfinite(res[1]))
                    all_finite = false;
            }
            if (!all_finite)
            {
                delete func[0];
                delete func[1];
                continue;
            }
            for (int i = 0; i < 2; i++)
                _problem.AddResidualBlock(func[i], _loss.get(), args[i]);
            if (options.hasAny(LENS_DISTORTIONS_RADIAL))
                for (int i = 0; i < 2; i++)
                    trackRadialObservation(models[i]->radial_distortion, models[i]->pixels_rows, models[i]->pixels_cols,
                                           models[i]->focal_length_pixels);
            points_added = true;
        }
        if (points_added)
        {
            for (int i = 0; i < 2; i++)
                _problem.SetManifold(orientation_ptrs[i], mc::Manifold::EIGEN_QUATERNION);
            if (!src.optimize)
                _problem.SetParameterBlockConstant(orientation_ptrs[0]);
            if (!dst.optimize)
                _problem.SetParameterBlockConstant(orientation_ptrs[1]);
            if (options.hasAny(FOCAL_LENGTH | PRINCIPAL_POINT | LENS_DISTORTIONS_RADIAL | LENS_DISTORTIONS_TANGENTIAL))
            {
                for (int i = 0; i < 2; i++)
                {
                    if (!options.hasAny(FOCAL_LENGTH))
                        _problem.SetParameterBlockConstant(&models[i]->focal_length_pixels);
                    else
                    {
                        _problem.SetParameterLowerBound(&models[i]->focal_length_pixels, 0, 100.0);
                        _problem.SetParameterUpperBound(&models[i]->focal_length_pixels, 0, 20000.0);
                    }
                    if (!options.hasAny(PRINCIPAL_POINT))
                        _problem.SetParameterBlockConstant(models[i]->principle_point);
                }
            }
            if (options.hasAll(LENS_DISTORTIONS_RADIAL))
                for (int i = 0; i < 2; i++)
                {
                    if (options.hasAll(LENS_DISTORTIONS_RADIAL_BROWN246_PARAMETERIZATION))
                        ; // all three coefficients free
                    else if (options.hasAll(LENS_DISTORTIONS_RADIAL_BROWN24_PARAMETERIZATION))
                        _problem.SetSubsetManifold(models[i]->radial_distortion, {2});
                    else if (options.hasAll(LENS_DISTORTIONS_RADIAL_BROWN2_PARAMETERIZATION))
                        _problem.SetSubsetManifold(models[i]->radial_distortion, {1, 2});
                }
        }
        _edges_used.insert(edge_id);
    }

    const std::vector<std::pair<size_t, std::vector<FeatureTrack>>> &tracks() const // TestRelaxProblem::test_get_tracks
    {
        return _edge_tracks;
    }

    // :931-984
    void relaxObservedModelOnly()
    {
        std::vector<double *> params = _problem.GetParameterBlocks();
        std::vector<std::pair<double *, bool>> params_map;
        std::unordered_map<double *, bool> was_const;
        for (double *p : params)
        {
            const bool isConst = _problem.IsParameterBlockConstant(p);
            _problem.SetParameterBlockConstant(p);
            params_map.emplace_back(p, isConst);
            was_const.emplace(p, isConst);
        }
        for (auto &et : _edge_tracks)
            for (auto &t : et.second)
            {
                auto it = was_const.find(&t.point.x);
                if (it != was_const.end() && !it->second)
                    _problem.SetParameterBlockVariable(&t.point.x);
            }
        for (auto &n : _mesh.nodes)
        {
            auto it = was_const.find(&n.location.z);
            if (it != was_const.end() && !it->second)
                _problem.SetParameterBlockVariable(&n.location.z);
        }
        solve();
        for (const auto &pc : params_map)
        {
            if (pc.second)
                _problem.SetParameterBlockConstant(pc.first);
            else
                _problem.SetParameterBlockVariable(pc.first);
        }
    }

    // :1390-1420
    void solve()
    {
        if (_problem.NumParameterBlocks() == 0 || _problem.NumResidualBlocks() == 0)
            return;
        mc::SolverSummary s;
        mc::Solve(_opt, &_problem, &s);
        if (_stats)
        {
            _stats->solves++;
            _stats->iterations_total += (int)s.iterations.size();
            _stats->last_iterations = (int)s.iterations.size();
            _stats->iterations_per_solve.push_back((int)s.iterations.size());
            _stats->last_initial_cost = s.initial_cost;
            _stats->last_final_cost = s.final_cost;
            _stats->last_residual_blocks = _problem.NumResidualBlocks();
            _stats->last_parameter_blocks = _problem.NumParameterBlocks();
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
        // "copy back camera models" (:1415-1419): every model that got an inverse twin is replaced by the forward fit
        // of that twin, whether or not intrinsics were optimised
        for (auto &im : _inverse_cam_model_to_optimize)
        {
            CameraModel *dst = find_model(im.first);
            if (dst)
            {
                const camera_model fwd = convertModelToForward(im.second->m);
                static_cast<camera_model &>(*dst) = fwd;
                dst->id = im.first;
            }
        }
    }

    // :1422-1507
    surface_model getSurfaceModel()
    {
        surface_model s;
        std::vector<const FeatureTrack *> flat_tracks;
        for (const auto &et : _edge_tracks)
            for (const auto &t : et.second)
                flat_tracks.push_back(&t);
        UnionFind uf(flat_tracks.size());
        std::unordered_map<NodeIdFeatureIndex, size_t, nifi_hash> measurement_to_idx;
        for (size_t i = 0; i < flat_tracks.size(); i++)
        {
            const FeatureTrack &t = *flat_tracks[i];
            if (!finite3(t.point))
                continue;
            for (const auto &m : t.measurements)
            {
                auto ins = measurement_to_idx.emplace(m, i);
                if (!ins.second)
                    uf.unite(i, ins.first->second);
            }
        }
        struct MergedTrack
        {
            std::vector<Vec3> points;
            double min_error = std::numeric_limits<double>::infinity();
            std::vector<size_t> unique_nodes;
        };
        std::vector<std::pair<size_t, MergedTrack>> merged; // insertion order
        std::unordered_map<size_t, size_t> merged_index;
        for (size_t i = 0; i < flat_tracks.size(); i++)
        {
            const FeatureTrack &t = *flat_tracks[i];
            if (!finite3(t.point))
                continue;
            const size_t root = uf.find(i);
            auto it = merged_index.find(root);
            if (it == merged_index.end())
            {
                merged_index.emplace(root, merged.size());
                merged.emplace_back(root, MergedTrack());
                it = merged_index.find(root);
            }
            MergedTrack &m = merged[it->second].second;
            m.points.push_back(t.point);
            if (std::isfinite(t.error))
                m.min_error = std::min(m.min_error, t.error);
            for (const auto &meas : t.measurements)
                if (std::find(m.unique_nodes.begin(), m.unique_nodes.end(), meas.node_id) == m.unique_nodes.end())
                    m.unique_nodes.push_back(meas.node_id);
        }
        point_cloud cloud_points;
        for (const auto &rm : merged)
        {
            const MergedTrack &m = rm.second;
            const double max_allowed_error = (m.unique_nodes.size() >= 3) ? 10.0 : 1.0;
            if (m.min_error > max_allowed_error)
                continue;
            Vec3 pt;
            if (m.points.size() == 1)
                pt = m.points[0];
            else
            {
                const int n = std::min((int)m.points.size(), ROBUST_CENTROID_MAX_POINTS);
                V3<double> pts[ROBUST_CENTROID_MAX_POINTS];
                for (int i = 0; i < n; i++)
                    pts[i] = {m.points[i].x, m.points[i].y, m.points[i].z};
                const V3<double> c = robustCentroid<double>(pts, n, 1.0);
                pt = Vec3{c.x, c.y, c.z};
            }
            cloud_points.push_back(pt);
        }
        if (!cloud_points.empty())
            s.cloud.emplace_back(std::move(cloud_points));
        s.mesh = _mesh;
        return s;
    }

    mc::Problem _problem;
    mc::SolverOptions _opt;
    MeshGraph _mesh;

  private:
    CameraModel *find_model(size_t id)
    {
        for (auto &m : _cam_models_to_optimize)
            if (m.first == id)
                return m.second;
        return nullptr;
    }
    NodePose *find_pose(size_t node_id)
    {
        auto it = _node_index.find(node_id);
        return it == _node_index.end() ? nullptr : _nodes_to_optimize[it->second].second;
    }

    void initialize(std::vector<NodePose> &nodes, model_map &cam_models) // :147-161
    {
        for (NodePose &n : nodes)
            if (_node_index.emplace(n.node_id, _nodes_to_optimize.size()).second) // map::emplace keeps the first
                _nodes_to_optimize.emplace_back(n.node_id, &n);
        for (auto &id_model : cam_models)
        {
            bool found = false;
            for (auto &m : _cam_models_to_optimize)
                if (m.first == id_model.first)
                {
                    m.second = &id_model.second;
                    found = true;
                }
            if (!found)
                _cam_models_to_optimize.emplace_back(id_model.first, &id_model.second);
        }
    }

    bool shouldAddEdgeToOptimization(size_t edge_id) // :163-179 (membership in edges_to_optimize holds by construction)
    {
        return _edges_used.find(edge_id) == _edges_used.end();
    }

    PoseOpt nodeid2poseopt(const MeasurementGraph &graph, size_t node_id, bool load_cam_model = true) // :181-232
    {
        PoseOpt po;
        po.node_id = node_id;
        NodePose *np = find_pose(node_id);
        const image_node *node = graph.getNode(node_id);
        if (np != nullptr)
        {
            po.optimize = true;
            po.loc_ptr = &np->position;
            po.rot_ptr = &np->orientation;
        }
        else
        {
            po.optimize = false;
            if (node != nullptr && finiteq(node->orientation) && finite3(node->position))
            {
                po.loc_ptr = &node->position;
                po.rot_ptr = const_cast<Quat *>(&node->orientation);
            }
        }
        if (load_cam_model && node != nullptr)
        {
            const CameraModel *m = find_model(node->model->id);
            po.model_ptr = m ? m : node->model.get();
        }
        return po;
    }

    void gridFilterMatchesPerImage(const MeasurementGraph &graph, const std::vector<size_t> &edges_to_optimize,
                                   double grid_cell_image_fraction) // :234-309
    {
        for (size_t edge_id : edges_to_optimize)
        {
            const graph_edge *edge_ptr = graph.getEdge(edge_id);
            if (edge_ptr == nullptr)
                continue;
            const graph_edge &edge = *edge_ptr;
            const PoseOpt src = nodeid2poseopt(graph, edge.source), dst = nodeid2poseopt(graph, edge.dest);
            if (src.loc_ptr == nullptr || dst.loc_ptr == nullptr)
                return; // sic
            const CameraModel &sm = *graph.getNode(edge.source)->model, &dm = *graph.getNode(edge.dest)->model;
            const Mat3 srot = quat_to_matrix(*src.rot_ptr), drot = quat_to_matrix(*dst.rot_ptr);
            auto &sf = _grid_filter[edge.source][edge_id];
            auto &df = _grid_filter[edge.dest][edge_id];
            sf.setResolution(grid_cell_image_fraction);
            df.setResolution(grid_cell_image_fraction);
            std::vector<std::pair<double, size_t>> scored;
            scored.reserve(edge.payload.inlier_matches.size());
            for (size_t idx = 0; idx < edge.payload.inlier_matches.size(); idx++)
            {
                const auto &inl = edge.payload.inlier_matches[idx];
                const Vec3 sdir = mul(srot, image_to_3d(inl.pixel_1, sm)), ddir = mul(drot, image_to_3d(inl.pixel_2, dm));
                const auto isect = rayIntersection(sdir, *src.loc_ptr, ddir, *dst.loc_ptr);
                const double intersection_score = isect.second < 0 ? 0. : 1. / (1. + isect.second);
                const double cos_angle = dot(sdir, ddir);
                const double angle_score = 1.0 - cos_angle * cos_angle;
                const double descriptor_score = inl.match_index < edge.payload.matches.size()
                                                    ? 1.0 - edge.payload.matches[inl.match_index].distance
                                                    : 1.0;
                const double snx = (inl.pixel_1[0] - sm.principle_point[0]) / sm.focal_length_pixels;
                const double sny = (inl.pixel_1[1] - sm.principle_point[1]) / sm.focal_length_pixels;
                const double dnx = (inl.pixel_2[0] - dm.principle_point[0]) / dm.focal_length_pixels;
                const double dny = (inl.pixel_2[1] - dm.principle_point[1]) / dm.focal_length_pixels;
                double ransac_score = 1.0;
                if (edge.payload.is_homography)
                {
                    const Vec2 h = hnormalized(mul(edge.payload.ransac_relation, Vec3{snx, sny, 1.0}));
                    const double ex = dnx - h.x, ey = dny - h.y;
                    ransac_score = 1.0 / (1.0 + std::sqrt(ex * ex + ey * ey));
                }
                scored.emplace_back(intersection_score * angle_score * descriptor_score * ransac_score, idx);
            }
            std::sort(scored.begin(), scored.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
            for (const auto &si : scored)
                if (si.first > 0)
                {
                    const auto &inl = edge.payload.inlier_matches[si.second];
                    sf.addMeasurement(inl.pixel_1[0] / sm.pixels_cols, inl.pixel_1[1] / sm.pixels_rows, si.first, &inl);
                    df.addMeasurement(inl.pixel_2[0] / dm.pixels_cols, inl.pixel_2[1] / dm.pixels_rows, si.first, &inl);
                }
        }
    }

    void collectEdgeTracks(const MeasurementGraph &graph, size_t edge_id, const graph_edge &edge) // :351-386
    {
        _edge_tracks.emplace_back(edge_id, std::vector<FeatureTrack>());
        auto &points = _edge_tracks.back().second;
        points.reserve(edge.payload.inlier_matches.size());
        const PoseOpt src = nodeid2poseopt(graph, edge.source), dst = nodeid2poseopt(graph, edge.dest);
        if (src.loc_ptr == nullptr || dst.loc_ptr == nullptr)
            return;
        const CameraModel &sm = *src.model_ptr, &dm = *dst.model_ptr;
        for (const auto &inl : edge.payload.inlier_matches)
        {
            const Vec3 sdir = image_to_3d(inl.pixel_1, sm), ddir = image_to_3d(inl.pixel_2, dm);
            const auto isect = rayIntersection(quat_rotate_d(*src.rot_ptr, sdir), *src.loc_ptr,
                                               quat_rotate_d(*dst.rot_ptr, ddir), *dst.loc_ptr);
            FeatureTrack t;
            t.point = isect.first;
            t.error = isect.second;
            t.measurements = {NodeIdFeatureIndex{edge.source, inl.feature_index_1},
                              NodeIdFeatureIndex{edge.dest, inl.feature_index_2}};
            points.push_back(std::move(t));
        }
    }

    InverseModel *inverse_model_for(size_t model_id, const camera_model &forward) // :402-407, :821-827
    {
        for (auto &im : _inverse_cam_model_to_optimize)
            if (im.first == model_id)
                return im.second.get();
        auto p = std::make_unique<InverseModel>();
        p->m = convertModelToInverse(forward);
        _inverse_cam_model_to_optimize.emplace_back(model_id, std::move(p));
        return _inverse_cam_model_to_optimize.back().second.get();
    }

    void set_radial_manifold(const RelaxOptionSet &options, double *radial) // :533-556, :889-901
    {
        if (!options.hasAny(LENS_DISTORTIONS_RADIAL))
            return;
        if (options.hasAll(LENS_DISTORTIONS_RADIAL_BROWN246_PARAMETERIZATION))
            _problem.SetManifold(radial, mc::Manifold::EUCLIDEAN); // SubsetManifold(3) with no constant coordinate
        else if (options.hasAll(LENS_DISTORTIONS_RADIAL_BROWN24_PARAMETERIZATION))
            _problem.SetSubsetManifold(radial, {2});
        else if (options.hasAll(LENS_DISTORTIONS_RADIAL_BROWN2_PARAMETERIZATION))
            _problem.SetSubsetManifold(radial, {1, 2});
    }

    void addRayTriangleMeasurementCost(const MeasurementGraph &graph, size_t edge_id, const graph_edge &edge,
                                       const RelaxOptionSet &options) // :388-560
    {
        const PoseOpt src = nodeid2poseopt(graph, edge.source), dst = nodeid2poseopt(graph, edge.dest);
        if (src.loc_ptr == nullptr || dst.loc_ptr == nullptr)
            return;
        const CameraModel &sm = *src.model_ptr, &dm = *dst.model_ptr;
        InverseModel *inverse = inverse_model_for(sm.id, sm);
        const auto &swl = _grid_filter[edge.source][edge_id].getBestMeasurementsPerCell();
        const auto &dwl = _grid_filter[edge.dest][edge_id].getBestMeasurementsPerCell();
        double *datas[2] = {&src.rot_ptr->x, &dst.rot_ptr->x};
        MeshIntersectionSearcher searcher;
        if (!searcher.init(_mesh))
            return;
        const bool intrinsics = options.hasAny(FOCAL_LENGTH | PRINCIPAL_POINT | LENS_DISTORTIONS_RADIAL);
        bool points_added = false;
        for (const auto &inl : edge.payload.inlier_matches)
        {
            if (swl.find(&inl) == swl.end() && dwl.find(&inl) == dwl.end())
                continue;
            const NodeIdFeatureIndex nifi_src{edge.source, inl.feature_index_1}, nifi_dst{edge.dest, inl.feature_index_2};
            if (_multi_ray_measurements.count(nifi_src) || _multi_ray_measurements.count(nifi_dst))
                continue;
            {
                auto cellKey = [this](double px, double py, double cols, double rows) {
                    const int gi = static_cast<int>(std::floor((px / cols) / _track_grid_fraction));
                    const int gj = static_cast<int>(std::floor((py / rows) / _track_grid_fraction));
                    return gridCellKey(gi, gj);
                };
                auto sc = _multi_ray_covered_cells.find(edge.source), dc = _multi_ray_covered_cells.find(edge.dest);
                const bool src_covered =
                    sc != _multi_ray_covered_cells.end() &&
                    sc->second.count(cellKey(inl.pixel_1[0], inl.pixel_1[1], (double)sm.pixels_cols, (double)sm.pixels_rows));
                const bool dst_covered =
                    dc != _multi_ray_covered_cells.end() &&
                    dc->second.count(cellKey(inl.pixel_2[0], inl.pixel_2[1], (double)dm.pixels_cols, (double)dm.pixels_rows));
                if (src_covered && dst_covered)
                    continue;
            }
            const Vec3 sray = image_to_3d(inl.pixel_1, sm), dray = image_to_3d(inl.pixel_2, dm);
            const auto isect = rayIntersection(quat_rotate_d(*src.rot_ptr, sray), *src.loc_ptr,
                                               quat_rotate_d(*dst.rot_ptr, dray), *dst.loc_ptr);
            const double mean_cam_z = (src.loc_ptr->z + dst.loc_ptr->z) * 0.5;
            const auto &tri = searcher.triangleIntersect(Vec3{0, 0, -1}, Vec3{isect.first.x, isect.first.y, mean_cam_z});
            if (tri.type != MeshIntersectionSearcher::INTERSECTION)
                continue;
            double corner2d[3][2];
            double *zValues[3];
            for (int i = 0; i < 3; i++)
            {
                corner2d[i][0] = tri.nodeLocations[i]->x;
                corner2d[i][1] = tri.nodeLocations[i]->y;
                zValues[i] = const_cast<double *>(&tri.nodeLocations[i]->z);
            }
            if (intrinsics && same_model(sm, dm))
            {
                auto *f = new TwoRayFocalRadial();
                const Vec3 locs[2] = {*src.loc_ptr, *dst.loc_ptr};
                for (int i = 0; i < 2; i++)
                {
                    f->impl.camera_loc[i][0] = locs[i].x, f->impl.camera_loc[i][1] = locs[i].y, f->impl.camera_loc[i][2] = locs[i].z;
                    f->impl.camera_ray[i][0] = f->impl.camera_ray[i][1] = f->impl.camera_ray[i][2] = NAN;
                }
                f->impl.camera_pixel[0][0] = inl.pixel_1[0], f->impl.camera_pixel[0][1] = inl.pixel_1[1];
                f->impl.camera_pixel[1][0] = inl.pixel_2[0], f->impl.camera_pixel[1][1] = inl.pixel_2[1];
                std::memcpy(f->impl.plane_point, corner2d, sizeof corner2d);
                f->impl.shared_tangential[0] = inverse->m.tangential_distortion[0];
                f->impl.shared_tangential[1] = inverse->m.tangential_distortion[1];
                _problem.AddResidualBlock(new mc::AutoDiffCostFunction<TwoRayFocalRadial, 6, 4, 4, 1, 1, 1, 1, 2, 3>(f),
                                          _loss.get(),
                                          {datas[0], datas[1], zValues[0], zValues[1], zValues[2], &inverse->m.focal_length_pixels,
                                           inverse->m.principle_point, inverse->m.radial_distortion});
                _problem.SetParameterLowerBound(&inverse->m.focal_length_pixels, 0, 100.0);
                _problem.SetParameterUpperBound(&inverse->m.focal_length_pixels, 0, 20000.0);
                if (!options.hasAny(FOCAL_LENGTH))
                    _problem.SetParameterBlockConstant(&inverse->m.focal_length_pixels);
                if (!options.hasAny(PRINCIPAL_POINT))
                    _problem.SetParameterBlockConstant(inverse->m.principle_point);
                trackRadialObservation(inverse->m.radial_distortion, sm.pixels_rows, sm.pixels_cols,
                                       inverse->m.focal_length_pixels);
                points_added = true;
            }
            else
            {
                auto *f = new PlaneIntersectionAngleCost();
                const Vec3 locs[2] = {*src.loc_ptr, *dst.loc_ptr}, rays[2] = {sray, dray};
                for (int i = 0; i < 2; i++)
                {
                    f->camera_loc[i][0] = locs[i].x, f->camera_loc[i][1] = locs[i].y, f->camera_loc[i][2] = locs[i].z;
                    f->camera_ray[i][0] = rays[i].x, f->camera_ray[i][1] = rays[i].y, f->camera_ray[i][2] = rays[i].z;
                }
                std::memcpy(f->plane_point, corner2d, sizeof corner2d);
                _problem.AddResidualBlock(new mc::AutoDiffCostFunction<PlaneIntersectionAngleCost, 6, 4, 4, 1, 1, 1>(f),
                                          _loss.get(), {datas[0], datas[1], zValues[0], zValues[1], zValues[2]});
                points_added = true;
            }
            if (_stats)
                _stats->two_ray_blocks++;
        }
        if (points_added)
        {
            _problem.SetManifold(datas[0], mc::Manifold::EIGEN_QUATERNION);
            _problem.SetManifold(datas[1], mc::Manifold::EIGEN_QUATERNION);
            if (!src.optimize)
                _problem.SetParameterBlockConstant(datas[0]);
            if (!dst.optimize)
                _problem.SetParameterBlockConstant(datas[1]);
            // (the radial block only exists in the problem when an intrinsics functor was added)
            if (_problem.HasParameterBlock(inverse->m.radial_distortion))
                set_radial_manifold(options, inverse->m.radial_distortion);
        }
        _edges_used.insert(edge_id);
    }

    void addMultiRayTrackCosts(const MeasurementGraph &graph, const RelaxOptionSet &options, double grid_fraction) // :608-929
    {
        _track_grid_fraction = grid_fraction;
        std::vector<const FeatureTrack *> flat_tracks;
        for (const auto &et : _edge_tracks)
            for (const auto &t : et.second)
                flat_tracks.push_back(&t);
        if (flat_tracks.empty())
            return;
        UnionFind uf(flat_tracks.size());
        std::unordered_map<NodeIdFeatureIndex, size_t, nifi_hash> measurement_to_idx;
        for (size_t i = 0; i < flat_tracks.size(); i++)
            for (const auto &m : flat_tracks[i]->measurements)
            {
                auto ins = measurement_to_idx.emplace(m, i);
                if (!ins.second)
                    uf.unite(i, ins.first->second);
            }
        std::vector<std::pair<size_t, std::vector<RayInfo>>> track_rays; // insertion order of the roots
        std::unordered_map<size_t, size_t> track_index;
        for (size_t i = 0; i < flat_tracks.size(); i++)
        {
            const size_t root = uf.find(i);
            auto it = track_index.find(root);
            if (it == track_index.end())
            {
                track_index.emplace(root, track_rays.size());
                track_rays.emplace_back(root, std::vector<RayInfo>());
                it = track_index.find(root);
            }
            auto &rays = track_rays[it->second].second;
            for (const auto &m : flat_tracks[i]->measurements)
            {
                bool already_present = false;
                for (const auto &existing : rays)
                    if (existing.node_id == m.node_id)
                    {
                        already_present = true;
                        break;
                    }
                if (already_present)
                    continue;
                NodePose *np = find_pose(m.node_id);
                if (np == nullptr)
                    continue;
                const image_node *node = graph.getNode(m.node_id);
                if (node == nullptr || m.feature_index >= node->feature_location.size())
                    continue;
                const CameraModel &model = *node->model;
                const Vec2 &pixel = node->feature_location[m.feature_index];
                const double px[2] = {pixel.x, pixel.y};
                rays.push_back(RayInfo{m.node_id, m.feature_index, model.id, np->position, image_to_3d(px, model), pixel,
                                       np->orientation, &np->orientation.x});
            }
        }
        std::vector<std::pair<size_t, GridFilter<size_t>>> track_grid_filter;
        std::unordered_map<size_t, size_t> filter_index;
        for (auto &tr : track_rays)
        {
            auto &rays = tr.second;
            if (rays.size() < 3)
                continue;
            const double score = static_cast<double>(rays.size());
            for (const auto &r : rays)
            {
                const image_node *node = graph.getNode(r.node_id);
                if (node == nullptr)
                    continue;
                const CameraModel &model = *node->model;
                auto it = filter_index.find(r.node_id);
                if (it == filter_index.end())
                {
                    filter_index.emplace(r.node_id, track_grid_filter.size());
                    track_grid_filter.emplace_back(r.node_id, GridFilter<size_t>());
                    it = filter_index.find(r.node_id);
                }
                auto &filter = track_grid_filter[it->second].second;
                filter.setResolution(grid_fraction);
                filter.addMeasurement(r.pixel.x / model.pixels_cols, r.pixel.y / model.pixels_rows, score, tr.first);
            }
        }
        std::unordered_set<size_t> accepted_tracks;
        for (const auto &nf : track_grid_filter)
            for (size_t root : nf.second.getBestMeasurementsPerCell())
                accepted_tracks.insert(root);

        MeshIntersectionSearcher searcher;
        if (!searcher.init(_mesh))
            return;
        const bool intrinsics = options.hasAny(FOCAL_LENGTH | PRINCIPAL_POINT | LENS_DISTORTIONS_RADIAL);
        for (auto &tr : track_rays)
        {
            auto &rays = tr.second;
            if (rays.size() < 3)
                continue;
            if (!accepted_tracks.count(tr.first))
                continue;
            Vec3 mean_loc{0, 0, 0};
            for (const auto &r : rays)
                mean_loc = mean_loc + r.camera_loc;
            mean_loc = mean_loc / static_cast<double>(rays.size());
            const Vec3 ray0_world = quat_rotate_d(rays[0].orientation, rays[0].camera_ray);
            const Vec3 ray1_world = quat_rotate_d(rays[1].orientation, rays[1].camera_ray);
            const auto intersection_3d = rayIntersection(ray0_world, rays[0].camera_loc, ray1_world, rays[1].camera_loc);
            if (!finite3(intersection_3d.first))
                continue;
            const auto &tri = searcher.triangleIntersect(Vec3{0, 0, -1},
                                                         Vec3{intersection_3d.first.x, intersection_3d.first.y, mean_loc.z});
            if (tri.type != MeshIntersectionSearcher::INTERSECTION)
                continue;
            double corner2d[3][2];
            double *zValues[3];
            Vec3 corner[3];
            for (int i = 0; i < 3; i++)
            {
                corner[i] = *tri.nodeLocations[i];
                corner2d[i][0] = corner[i].x;
                corner2d[i][1] = corner[i].y;
                zValues[i] = const_cast<double *>(&tri.nodeLocations[i]->z);
            }
            std::vector<std::pair<double, size_t>> ray_scores(rays.size());
            {
                const plane_no pno = cornerPlane2normOffsetPlane_d(corner);
                std::vector<Vec3> intersections(rays.size());
                bool all_valid = true;
                double avg_dist = 0;
                for (size_t i = 0; i < rays.size(); i++)
                {
                    const Vec3 dir = quat_rotate_d(rays[i].orientation, rays[i].camera_ray);
                    all_valid &= rayPlaneIntersection_d(dir, rays[i].camera_loc, pno, intersections[i]);
                    avg_dist += norm(intersections[i] - rays[i].camera_loc);
                }
                if (!all_valid)
                    continue;
                avg_dist /= static_cast<double>(rays.size());
                const int n = std::min(static_cast<int>(intersections.size()), ROBUST_CENTROID_MAX_POINTS);
                const double huber_threshold = avg_dist * 0.01;
                V3<double> pts[ROBUST_CENTROID_MAX_POINTS];
                for (int i = 0; i < n; i++)
                    pts[i] = {intersections[i].x, intersections[i].y, intersections[i].z};
                const V3<double> c = robustCentroid<double>(pts, n, huber_threshold);
                const Vec3 centroid{c.x, c.y, c.z};
                for (size_t i = 0; i < rays.size(); i++)
                    ray_scores[i] = {norm(intersections[i] - centroid) / avg_dist, i};
            }
            std::sort(ray_scores.begin(), ray_scores.end());
            const double median_err = ray_scores[ray_scores.size() / 2].first;
            const double threshold = std::max(median_err * 3.0, 1e-6);
            std::vector<RayInfo> good_rays;
            for (const auto &es : ray_scores)
                if (es.first <= threshold && good_rays.size() < 5)
                    good_rays.push_back(rays[es.second]);
            if (good_rays.size() < 3)
                continue;
            const int N = static_cast<int>(good_rays.size());
            bool all_same_model = true;
            for (int i = 1; i < N; i++)
                if (good_rays[i].camera_model_id != good_rays[0].camera_model_id)
                {
                    all_same_model = false;
                    break;
                }
            const bool use_focal_radial = all_same_model && intrinsics;
            std::vector<double *> param_blocks;
            mc::CostFunction *cost = nullptr;
            InverseModel *inv = nullptr;
            if (use_focal_radial)
            {
                const image_node *node = graph.getNode(good_rays[0].node_id);
                inv = inverse_model_for(good_rays[0].camera_model_id, *node->model);
                for (int i = 0; i < 3; i++)
                    param_blocks.push_back(zValues[i]);
                param_blocks.push_back(&inv->m.focal_length_pixels);
                param_blocks.push_back(inv->m.principle_point);
                param_blocks.push_back(inv->m.radial_distortion);
                for (int i = 0; i < N; i++)
                    param_blocks.push_back(good_rays[i].rot_ptr);
                cost = makeMultiRayCostFocalRadial(N, good_rays, corner2d, inv->m);
            }
            else
            {
                for (int i = 0; i < 3; i++)
                    param_blocks.push_back(zValues[i]);
                for (int i = 0; i < N; i++)
                    param_blocks.push_back(good_rays[i].rot_ptr);
                cost = N == 3 ? makeMultiRayCost<3>(good_rays, corner2d)
                              : N == 4 ? makeMultiRayCost<4>(good_rays, corner2d) : makeMultiRayCost<5>(good_rays, corner2d);
            }
            _problem.AddResidualBlock(cost, nullptr, param_blocks);
            if (inv != nullptr)
            {
                _problem.SetParameterLowerBound(&inv->m.focal_length_pixels, 0, 100.0);
                _problem.SetParameterUpperBound(&inv->m.focal_length_pixels, 0, 20000.0);
                if (!options.hasAny(FOCAL_LENGTH))
                    _problem.SetParameterBlockConstant(&inv->m.focal_length_pixels);
                if (!options.hasAny(PRINCIPAL_POINT))
                    _problem.SetParameterBlockConstant(inv->m.principle_point);
                set_radial_manifold(options, inv->m.radial_distortion);
                const image_node *node = graph.getNode(good_rays[0].node_id);
                trackRadialObservation(inv->m.radial_distortion, node->model->pixels_rows, node->model->pixels_cols,
                                       inv->m.focal_length_pixels);
            }
            for (int i = 0; i < N; i++)
            {
                _problem.SetManifold(good_rays[i].rot_ptr, mc::Manifold::EIGEN_QUATERNION);
                _multi_ray_measurements.insert(NodeIdFeatureIndex{good_rays[i].node_id, good_rays[i].feature_index});
                const image_node *node = graph.getNode(good_rays[i].node_id);
                if (node != nullptr)
                {
                    const CameraModel &model = *node->model;
                    const double nx = good_rays[i].pixel.x / model.pixels_cols, ny = good_rays[i].pixel.y / model.pixels_rows;
                    const int gi = static_cast<int>(std::floor(nx / grid_fraction));
                    const int gj = static_cast<int>(std::floor(ny / grid_fraction));
                    _multi_ray_covered_cells[good_rays[i].node_id].insert(gridCellKey(gi, gj));
                }
            }
            if (_stats)
                _stats->track_blocks++;
        }
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
        constexpr double margin = 50;
        height -= margin;
        const double cx = (xmin + xmax) / 2, cy = (ymin + ymax) / 2;
        const double spacing = std::max(xmax - xmin, ymax - ymin) + margin;
        _mesh = MeshGraph();
        size_t ids[3];
        ids[0] = _mesh.addNode(Vec3{-spacing + cx, -spacing + cy, height});
        ids[1] = _mesh.addNode(Vec3{spacing + cx, -spacing + cy, height});
        ids[2] = _mesh.addNode(Vec3{0 + cx, spacing + cy, height});
        for (size_t i = 0; i < 3; i++)
        {
            mesh_edge e;
            e.border = true;
            e.opposite[0] = ids[(i + 2) % 3];
            e.opposite[1] = 0; // {nodeIds[(i + 2) % 3], 0}
            _mesh.addEdge(e, ids[i], ids[(i + 1) % 3]);
        }
    }

    void initializeGroundMesh(const std::vector<surface_model> &previousSurfaces, bool useMinimalMesh) // :1244-1288
    {
        point_cloud cameraLocations;
        for (const auto &kv : _nodes_to_optimize)
            cameraLocations.push_back(kv.second->position);
        const MeshGraph *previousMesh = nullptr;
        for (const auto &s : previousSurfaces)
            if (s.mesh.size_nodes() > 0)
            {
                previousMesh = &s.mesh;
                break;
            }
        const bool previousIsGroundPlaneTriangle = previousMesh != nullptr && previousMesh->size_nodes() == 3;
        const bool shouldReusePreviousMesh = previousMesh != nullptr && !(useMinimalMesh && previousIsGroundPlaneTriangle);
        if (shouldReusePreviousMesh)
            _mesh = *previousMesh;
        else if (useMinimalMesh)
            _mesh = buildMinimalMesh(cameraLocations, previousSurfaces);
        else
            _mesh = rebuildMesh(cameraLocations, previousSurfaces);
    }

    void addDownwardsPrior() // :1290-1301
    {
        for (auto &p : _nodes_to_optimize)
            if (!hasnanq(p.second->orientation))
            {
                double *d = &p.second->orientation.x;
                _problem.AddResidualBlock(
                    new mc::AutoDiffCostFunction<PointsDownwardsPrior, 1, 4>(new PointsDownwardsPrior(1e-3)), nullptr, {d});
                _problem.SetManifold(d, mc::Manifold::EIGEN_QUATERNION);
            }
    }

    void addMeshFlatPrior() // :1303-1333
    {
        for (auto &e : _mesh.edges)
        {
            double *h1 = &_mesh.nodes[e.source].location.z, *h2 = &_mesh.nodes[e.dest].location.z;
            _problem.AddResidualBlock(new mc::AutoDiffCostFunction<DifferenceCost, 1, 1, 1>(new DifferenceCost(1e-4)), nullptr,
                                      {h1, h2});
        }
        _mesh_initial_z.clear();
        _mesh_initial_z.reserve(_mesh.size_nodes());
        for (auto &n : _mesh.nodes)
            _mesh_initial_z.push_back(n.location.z);
        size_t i = 0;
        for (auto &n : _mesh.nodes)
        {
            double *h = &n.location.z;
            _problem.AddResidualBlock(new mc::AutoDiffCostFunction<DifferenceCost, 1, 1, 1>(new DifferenceCost(1e-5)), nullptr,
                                      {h, &_mesh_initial_z[i]});
            _problem.SetParameterBlockConstant(&_mesh_initial_z[i]);
            ++i;
        }
    }

    void addMeshSmoothPrior() // :1335-1366
    {
        for (auto &e : _mesh.edges)
        {
            if (e.border)
                continue;
            const mesh_node &A = _mesh.nodes[e.source], &B = _mesh.nodes[e.dest], &C = _mesh.nodes[e.opposite[0]],
                            &D = _mesh.nodes[e.opposite[1]];
            auto *f = new AdjacentTriangleNormalCost();
            f->xyA[0] = A.location.x, f->xyA[1] = A.location.y;
            f->xyB[0] = B.location.x, f->xyB[1] = B.location.y;
            f->xyC[0] = C.location.x, f->xyC[1] = C.location.y;
            f->xyD[0] = D.location.x, f->xyD[1] = D.location.y;
            f->weight = 1e-4;
            _problem.AddResidualBlock(new mc::AutoDiffCostFunction<AdjacentTriangleNormalCost, 1, 1, 1, 1, 1>(f), nullptr,
                                      {const_cast<double *>(&A.location.z), const_cast<double *>(&B.location.z),
                                       const_cast<double *>(&C.location.z), const_cast<double *>(&D.location.z)});
        }
    }

    void trackRadialObservation(double *radial_data, size_t pixels_rows, size_t pixels_cols, double focal_length) // :1368-1379
    {
        for (auto &info : _radial_monotonicity_info)
            if (info.radial == radial_data)
            {
                info.observation_count++;
                return;
            }
        const double half_cols = pixels_cols / 2.0, half_rows = pixels_rows / 2.0;
        _radial_monotonicity_info.push_back(
            monotonicity_info{radial_data, 1, std::sqrt(half_cols * half_cols + half_rows * half_rows) / focal_length});
    }

    void addMonotonicityCosts() // :1381-1388
    {
        for (auto &info : _radial_monotonicity_info)
        {
            auto *f = new DistortionMonotonicityCost();
            f->r_max = info.r_max;
            f->weight = std::sqrt(info.observation_count / 10.0);
            _problem.AddResidualBlock(new mc::AutoDiffCostFunction<DistortionMonotonicityCost, 10, 3>(f), nullptr, {info.radial});
        }
    }

    struct monotonicity_info
    {
        double *radial;
        size_t observation_count;
        double r_max;
    };

    relax_stats *_stats;
    std::vector<std::pair<size_t, NodePose *>> _nodes_to_optimize; // insertion order
    std::unordered_map<size_t, size_t> _node_index;
    std::vector<std::pair<size_t, CameraModel *>> _cam_models_to_optimize;
    std::vector<std::pair<size_t, std::unique_ptr<InverseModel>>> _inverse_cam_model_to_optimize;
    std::map<size_t, std::map<size_t, GridFilter<const feature_match_denormalized *>>> _grid_filter;
    std::unordered_set<size_t> _edges_used;
    std::vector<std::pair<size_t, std::vector<FeatureTrack>>> _edge_tracks; // insertion order
    std::unordered_set<NodeIdFeatureIndex, nifi_hash> _multi_ray_measurements;
    std::unordered_map<size_t, std::unordered_set<uint64_t>> _multi_ray_covered_cells;
    double _track_grid_fraction = 0.1;
    std::vector<double> _mesh_initial_z;
    std::vector<monotonicity_info> _radial_monotonicity_info;
    std::unique_ptr<mc::LossFunction> _loss;
};

// ------------------------------------------------------------------------------------------------------- relax.cpp
const Quat DOWN_ORIENTED_NORTH{std::sin(M_PI / 2), 0.0, 0.0, std::cos(M_PI / 2)}; // AngleAxis(pi, UnitX)

surface_model runGroundPlane(const MeasurementGraph &graph, std::vector<NodePose> &nodes, model_map &cam_models,
                             const std::vector<size_t> &edges_to_optimize, const RelaxOptionSet &options, relax_stats *stats)
{
    Quat previous = DOWN_ORIENTED_NORTH;
    for (auto &node : nodes)
    {
        if (hasnanq(node.orientation))
        {
            node.orientation = previous;
            if (graph.nodes.size() > 2 * nodes.size())
            {
                std::vector<NodePose> justThis{node};
                RelaxProblem rp(stats);
                rp.setupGroundPlaneProblem(graph, justThis, cam_models, edges_to_optimize, options);
                rp.relaxObservedModelOnly();
                rp.solve();
                node = justThis[0];
            }
            else
            {
                RelaxProblem rp(stats);
                rp.setupGroundPlaneProblem(graph, nodes, cam_models, edges_to_optimize, options);
                rp.relaxObservedModelOnly();
                rp.solve();
            }
        }
        previous = node.orientation;
    }
    if (stats)
        stats->track_blocks = stats->two_ray_blocks = 0;
    RelaxProblem rp(stats);
    rp.setupGroundPlaneProblem(graph, nodes, cam_models, edges_to_optimize, options);
    rp.relaxObservedModelOnly();
    rp.solve();
    return rp.getSurfaceModel();
}

surface_model runGroundMesh(const MeasurementGraph &graph, std::vector<NodePose> &nodes, model_map &cam_models,
                            const std::vector<size_t> &edges_to_optimize, const RelaxConfig &config,
                            const std::vector<surface_model> &previousSurfaces, relax_stats *stats)
{
    if (stats)
        stats->track_blocks = stats->two_ray_blocks = 0;
    RelaxProblem rp(stats);
    rp.setupGroundMeshProblem(graph, nodes, cam_models, edges_to_optimize, config.options, previousSurfaces,
                              config.ground_mesh_grid_fraction);
    rp.relaxObservedModelOnly();
    rp.solve();
    return rp.getSurfaceModel();
}

surface_model runRelativeOrientation(const MeasurementGraph &graph, std::vector<NodePose> &nodes,
                                     const std::vector<size_t> &edges_to_optimize, relax_stats *stats) // relax.cpp:14-42
{
    for (auto &node : nodes)
        if (hasnanq(node.orientation))
        {
            node.orientation = DOWN_ORIENTED_NORTH;
            RelaxProblem rp(stats);
            rp.setupDecompositionProblem(graph, nodes, edges_to_optimize);
            rp.solve();
        }
    RelaxProblem rp(stats);
    rp.setupDecompositionProblem(graph, nodes, edges_to_optimize);
    rp.solve();
    return rp.getSurfaceModel();
}

surface_model runPoints(const MeasurementGraph &graph, std::vector<NodePose> &nodes, model_map &cam_models,
                        const std::vector<size_t> &edges_to_optimize, const RelaxOptionSet &options, relax_stats *stats) // :104-116
{
    RelaxProblem rp(stats);
    rp.setup3dPointProblem(graph, nodes, cam_models, edges_to_optimize, options);
    rp.relaxObservedModelOnly();
    rp.solve();
    return rp.getSurfaceModel();
}

} // namespace

surface_model relax(const MeasurementGraph &graph, std::vector<NodePose> &nodes, model_map &cam_models,
                    const std::vector<size_t> &edges_to_optimize, const RelaxConfig &config,
                    const std::vector<surface_model> &previousSurfaces, relax_stats *stats) // relax.cpp:118-134
{
    if (config.options.get(GROUND_MESH))
        return runGroundMesh(graph, nodes, cam_models, edges_to_optimize, config, previousSurfaces, stats);
    if (config.options.get(POINTS_3D))
        return runPoints(graph, nodes, cam_models, edges_to_optimize, config.options, stats);
    if (config.options.get(GROUND_PLANE))
        return runGroundPlane(graph, nodes, cam_models, edges_to_optimize, config.options, stats);
    return runRelativeOrientation(graph, nodes, edges_to_optimize, stats);
}

// TestRelaxProblem of test/test_relax.cpp:470-483: the 3-D point problem step by step.  mode 0: set-up only, 1: set-up +
// solve, 2: set-up + relaxObservedModelOnly.  points_before / points_after: the tracks' points after the set-up / at the end.
void points_problem_steps(const MeasurementGraph &graph, std::vector<NodePose> &nodes, model_map &cam_models,
                          const std::vector<size_t> &edges_to_optimize, const RelaxOptionSet &options, int mode,
                          std::vector<Vec3> *points_before, std::vector<Vec3> *points_after, relax_stats *stats)
{
    RelaxProblem rp(stats);
    rp.setup3dPointProblem(graph, nodes, cam_models, edges_to_optimize, options);
    auto collect = [&](std::vector<Vec3> *out) {
        if (out)
            for (const auto &et : rp.tracks())
                for (const auto &t : et.second)
                    out->push_back(t.point);
    };
    collect(points_before);
    if (mode == 1)
        rp.solve();
    else if (mode == 2)
        rp.relaxObservedModelOnly();
    collect(points_after);
}

// ------------------------------------------------------------------------------------------------- relax_group.cpp
void RelaxGroup::init(const MeasurementGraph &graph, const std::vector<size_t> &node_ids, const std::vector<size_t> &knn10,
                      size_t graph_connection_depth, const RelaxConfig &config)
{
    _directly_connected.clear();
    _directly_set.clear();
    _edges_to_optimize.clear();
    _edges_set.clear();
    _nodes_to_optimize.clear();
    _local_poses.clear();
    _config = config;
    _nodes_to_optimize.insert(node_ids.begin(), node_ids.end());
    auto add_pose = [&](size_t node_id) {
        const image_node *node = graph.getNode(node_id);
        NodePose pose;
        pose.node_id = node_id;
        pose.orientation = node->orientation;
        pose.position = node->position;
        _local_poses.push_back(pose);
        bool found = false;
        for (auto &m : _camera_models)
            if (m.first == node->model->id)
            {
                m.second = *node->model;
                found = true;
            }
        if (!found)
            _camera_models.emplace_back(node->model->id, *node->model);
        build_optimization_edges(graph, knn10, node_id);
    };
    for (size_t node_id : node_ids)
        add_pose(node_id);
    for (size_t i = 0; i < graph_connection_depth; i++)
    {
        // newly_connected: an insertion-ordered set of the directly connected nodes that are not primary nodes.  It is
        // rebuilt from ALL of _directly_connected every round and _nodes_to_optimize never grows, so nodes found in
        // round 0 are appended to _local_poses again in round 1 (as in the reference, relax_group.cpp:40-66).
        std::vector<size_t> newly_connected;
        for (size_t id : _directly_connected)
            if (_nodes_to_optimize.find(id) == _nodes_to_optimize.end())
                newly_connected.push_back(id);
        for (size_t node_id : newly_connected)
            add_pose(node_id);
    }
    std::sort(_local_poses.begin(), _local_poses.end(), [&graph](const NodePose &a, const NodePose &b) {
        return graph.getNode(a.node_id)->path < graph.getNode(b.node_id)->path;
    });
}

void RelaxGroup::build_optimization_edges(const MeasurementGraph &graph, const std::vector<size_t> &knn10, size_t node_id)
{
    const image_node *node = graph.getNode(node_id);
    std::unordered_set<size_t> ideally_connected_nodes;
    for (size_t k = 0; k < 10; k++)
        if (knn10[node_id * 10 + k] != NONE)
            ideally_connected_nodes.insert(knn10[node_id * 10 + k]);
    ideally_connected_nodes.erase(node_id);
    auto connect = [&](size_t other, size_t edge_id) {
        if (_directly_set.insert(other).second)
            _directly_connected.push_back(other);
        if (_nodes_to_optimize.find(other) != _nodes_to_optimize.end())
            if (_edges_set.insert(edge_id).second)
                _edges_to_optimize.push_back(edge_id);
    };
    for (size_t edge_id : node->edges)
    {
        const graph_edge *edge = graph.getEdge(edge_id);
        if (edge->source == node_id && ideally_connected_nodes.count(edge->dest))
            connect(edge->dest, edge_id);
        else if (edge->dest == node_id && ideally_connected_nodes.count(edge->source))
            connect(edge->source, edge_id);
    }
}

surface_model RelaxGroup::run(const MeasurementGraph &graph, const std::vector<surface_model> &previousSurfaces,
                              relax_stats *stats)
{
    return relax(graph, _local_poses, _camera_models, _edges_to_optimize, _config, previousSurfaces, stats);
}

std::vector<size_t> RelaxGroup::finalize(MeasurementGraph &graph)
{
    std::vector<size_t> optimized_ids;
    const bool model_changed =
        _config.options.hasAny(FOCAL_LENGTH | PRINCIPAL_POINT | LENS_DISTORTIONS_RADIAL | LENS_DISTORTIONS_TANGENTIAL);
    for (const auto &pose : _local_poses)
    {
        image_node *node = graph.getNode(pose.node_id);
        node->orientation = pose.orientation;
        node->position = pose.position;
        if (model_changed)
            for (auto &m : _camera_models)
                if (m.first == node->model->id && !same_model(*node->model, m.second))
                    *node->model = m.second;
        optimized_ids.push_back(pose.node_id);
    }
    // (the re-fit of every edge on its previous inliers, relax_group.cpp:137-177, is restated in oracle/link.cpp:
    //  oc_refit_edge; the caller applies it when model_changed)
    _local_poses.clear();
    return optimized_ids;
}

} // namespace rx
} // namespace oracle

// ---- test hooks: the restated GridFilter / UnionFind against the reference's own headers (oracle/_ref, tests/test_oracle_ref_pins.py)
extern "C"
{
void ocx_grid_filter_values(const double *xy, const double *score, const uint64_t *value, size_t n, double resolution,
                            uint64_t *out_values, size_t *n_out)
{
    oracle::rx::GridFilter<size_t> f;
    f.setResolution(resolution);
    for (size_t i = 0; i < n; i++)
        f.addMeasurement(xy[2 * i], xy[2 * i + 1], score[i], (size_t)value[i]);
    size_t k = 0;
    for (size_t v : f.getBestMeasurementsPerCell())
        out_values[k++] = v;
    *n_out = k;
}
uint64_t ocx_grid_cell_key(int i, int j)
{
    return oracle::rx::gridCellKey(i, j);
}
void ocx_union_find(size_t n, const uint64_t *pairs, size_t n_pairs, uint64_t *roots)
{
    oracle::rx::UnionFind uf(n);
    for (size_t i = 0; i < n_pairs; i++)
        uf.unite(pairs[2 * i], pairs[2 * i + 1]);
    for (size_t i = 0; i < n; i++)
        roots[i] = uf.find(i);
}
}

// ---- test exports: the geometry primitives and the plain cost functors, for the reference's own unit tests restated in
// tests/test_reference_unit_tests.py (test/test_geometry.cpp, test_cost_functions.cpp, test_meshgraph.cpp)
extern "C"
{
// rayIntersection (intersection.cpp:116-143): out4 = point (3), signed squared distance
void ocx_ray_intersection(const double *d1, const double *o1, const double *d2, const double *o2, double *out4)
{
    const auto r = oracle::rx::rayIntersection(oracle::Vec3{d1[0], d1[1], d1[2]}, oracle::Vec3{o1[0], o1[1], o1[2]},
                                               oracle::Vec3{d2[0], d2[1], d2[2]}, oracle::Vec3{o2[0], o2[1], o2[2]});
    out4[0] = r.first.x, out4[1] = r.first.y, out4[2] = r.first.z, out4[3] = r.second;
}
// cornerPlane2normOffsetPlane (intersection.hpp:26-34)
void ocx_corner_plane(const double *c9, double *norm3, double *off3)
{
    const oracle::Vec3 c[3] = {{c9[0], c9[1], c9[2]}, {c9[3], c9[4], c9[5]}, {c9[6], c9[7], c9[8]}};
    const auto p = oracle::rx::cornerPlane2normOffsetPlane_d(c);
    norm3[0] = p.norm.x, norm3[1] = p.norm.y, norm3[2] = p.norm.z;
    off3[0] = p.offset.x, off3[1] = p.offset.y, off3[2] = p.offset.z;
}
// rayPlaneIntersection (intersection.hpp:36-47)
int ocx_ray_plane(const double *dir3, const double *off3, const double *norm3, const double *poff3, double *out3)
{
    oracle::rx::plane_no p;
    p.norm = oracle::Vec3{norm3[0], norm3[1], norm3[2]};
    p.offset = oracle::Vec3{poff3[0], poff3[1], poff3[2]};
    oracle::Vec3 out{NAN, NAN, NAN};
    const bool ok = oracle::rx::rayPlaneIntersection_d(oracle::Vec3{dir3[0], dir3[1], dir3[2]}, oracle::Vec3{off3[0], off3[1], off3[2]}, p, out);
    out3[0] = out.x, out3[1] = out.y, out3[2] = out.z;
    return ok ? 1 : 0;
}
// the cost functors on doubles (relax_cost_function.hpp:16-19, :51-69, :119-155, :157-185)
double ocx_angle_between_unit_vectors(const double *a3, const double *b3)
{
    return oracle::angleBetweenUnitVectors<double>(oracle::V3<double>{a3[0], a3[1], a3[2]}, oracle::V3<double>{b3[0], b3[1], b3[2]});
}
double ocx_difference_cost(double weight, double v1, double v2)
{
    double r = NAN;
    oracle::DifferenceCost f(weight);
    f(&v1, &v2, &r);
    return r;
}
void ocx_distortion_monotonicity(double r_max, double weight, const double *radial3, double *residuals10)
{
    oracle::DistortionMonotonicityCost f;
    f.r_max = r_max, f.weight = weight;
    f(radial3, residuals10);
}
double ocx_adjacent_triangle_normal(const double *xy8, const double *z4, double weight)
{
    oracle::AdjacentTriangleNormalCost f;
    for (int i = 0; i < 2; i++)
        f.xyA[i] = xy8[i], f.xyB[i] = xy8[2 + i], f.xyC[i] = xy8[4 + i], f.xyD[i] = xy8[6 + i];
    f.weight = weight;
    double r = NAN;
    f(&z4[0], &z4[1], &z4[2], &z4[3], &r);
    return r;
}
}

// DecomposedRotationCost (relax_cost_function.hpp:187-251) on doubles: quaternions x y z w, for test/test_relax.cpp:250-296 restated
extern "C" void ocx_decomposed_rotation_cost(const double *rel_rot4, const double *rel_pos3, const double *pos1, const double *pos2,
                                             int score, const double *q1, const double *q2, double *residuals3)
{
    oracle::Quat r;
    r.x = rel_rot4[0], r.y = rel_rot4[1], r.z = rel_rot4[2], r.w = rel_rot4[3];
    const oracle::DecomposedRotationCost f(r, oracle::Vec3{rel_pos3[0], rel_pos3[1], rel_pos3[2]}, oracle::Vec3{pos1[0], pos1[1], pos1[2]},
                                           oracle::Vec3{pos2[0], pos2[1], pos2[2]}, score);
    f(q1, q2, residuals3);
}
