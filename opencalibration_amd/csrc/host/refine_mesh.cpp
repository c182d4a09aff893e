#include "refine_mesh.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace opencalibration_amd
{

namespace
{

constexpr size_t NONE = MeshEdge::NONE;

inline bool alive(const MeshGraph &m, size_t e)
{
    return e < m.edges.size() && m.edges[e].source != NONE;
}

// getTriangleVertices (:111-122); false = the reference's {0, 0, 0} / a missing node
inline bool triangle_vertices(const MeshGraph &m, const TriangleId &t, size_t v[3])
{
    if (!alive(m, t.edgeId))
        return false;
    const MeshEdge &e = m.edges[t.edgeId];
    v[0] = e.source;
    v[1] = e.dest;
    v[2] = e.triangleOppositeNodes[t.side];
    return v[0] < m.nodes.size() && v[1] < m.nodes.size() && v[2] < m.nodes.size();
}

double edge_length_squared(const MeshGraph &m, size_t e) // :15-28, x-y metric
{
    if (!alive(m, e))
        return 0;
    const double *a = m.nodes[m.edges[e].source].location, *b = m.nodes[m.edges[e].dest].location;
    const double dx = a[0] - b[0], dy = a[1] - b[1];
    return dx * dx + dy * dy;
}

size_t find_edge_between(const MeshGraph &m, size_t n1, size_t n2) // :30-46
{
    if (n1 < m.node_edges.size())
        for (size_t eid : m.node_edges[n1])
        {
            const MeshEdge &e = m.edges[eid];
            if ((e.source == n1 && e.dest == n2) || (e.source == n2 && e.dest == n1))
                return eid;
        }
    return NONE;
}

bool point_in_triangle_2d(double px, double py, const double *v0, const double *v1, const double *v2) // :48-63
{
    auto sign = [](double p1x, double p1y, double p2x, double p2y, double p3x, double p3y) {
        return (p1x - p3x) * (p2y - p3y) - (p2x - p3x) * (p1y - p3y);
    };
    const double d1 = sign(px, py, v0[0], v0[1], v1[0], v1[1]);
    const double d2 = sign(px, py, v1[0], v1[1], v2[0], v2[1]);
    const double d3 = sign(px, py, v2[0], v2[1], v0[0], v0[1]);
    const bool neg = (d1 < 0) || (d2 < 0) || (d3 < 0), pos = (d1 > 0) || (d2 > 0) || (d3 > 0);
    return !(neg && pos);
}

int find_triangle_side(const MeshGraph &m, size_t e, size_t opposite) // :65-76
{
    if (!alive(m, e))
        return -1;
    if (m.edges[e].triangleOppositeNodes[0] == opposite)
        return 0;
    if (m.edges[e].triangleOppositeNodes[1] == opposite)
        return 1;
    return -1;
}

TriangleId find_triangle_near_vertices(const MeshGraph &m, const size_t vertices[3], double x, double y) // :81-107
{
    for (int k = 0; k < 3; k++)
    {
        const size_t vtx = vertices[k];
        if (vtx >= m.node_edges.size())
            continue;
        for (size_t eid : m.node_edges[vtx])
            for (int side = 0; side < 2; side++)
            {
                const TriangleId cand{eid, side};
                size_t v[3];
                if (!triangle_vertices(m, cand, v))
                    continue;
                if (point_in_triangle_2d(x, y, m.nodes[v[0]].location, m.nodes[v[1]].location, m.nodes[v[2]].location))
                    return cand;
            }
    }
    return TriangleId();
}

size_t find_longest_edge(const MeshGraph &m, const TriangleId &t) // :124-149
{
    size_t v[3];
    if (!triangle_vertices(m, t, v))
        return NONE;
    std::pair<size_t, double> edges[3];
    edges[0] = {t.edgeId, edge_length_squared(m, t.edgeId)};
    const size_t e1 = find_edge_between(m, v[1], v[2]);
    edges[1] = {e1, e1 != NONE ? edge_length_squared(m, e1) : 0};
    const size_t e2 = find_edge_between(m, v[2], v[0]);
    edges[2] = {e2, e2 != NONE ? edge_length_squared(m, e2) : 0};
    int longest = 0;
    for (int i = 1; i < 3; i++)
        if (edges[i].second > edges[longest].second)
            longest = i;
    return edges[longest].first;
}

// The mesh while it is being refined: edge ids stay put (a removed edge is a tombstone), `order` is the iteration
// order of the reference's edge map, node_edges the vertices' edge sets - both with unordered_dense's erase.
struct Refining
{
    MeshGraph &m;
    std::vector<size_t> order, pos;
    explicit Refining(MeshGraph &mesh) : m(mesh)
    {
        order.resize(m.edges.size());
        pos.resize(m.edges.size());
        for (size_t i = 0; i < order.size(); i++)
            order[i] = pos[i] = i;
    }
    size_t add_edge(MeshEdge e, size_t s, size_t d)
    {
        const size_t id = m.addEdge(e, s, d);
        pos.push_back(order.size());
        order.push_back(id);
        return id;
    }
    static void erase_from(std::vector<size_t> &set, size_t id) // ankerl::unordered_dense::set::erase(key)
    {
        for (size_t i = 0; i < set.size(); i++)
            if (set[i] == id)
            {
                set[i] = set.back();
                set.pop_back();
                return;
            }
    }
    void remove_edge(size_t id) // DirectedGraph::removeEdge (graph.hpp:183-213)
    {
        MeshEdge &e = m.edges[id];
        erase_from(m.node_edges[e.source], id);
        erase_from(m.node_edges[e.dest], id);
        m.forget_edge(e.source, e.dest);
        const size_t p = pos[id], last = order.back();
        order[p] = last;
        pos[last] = p;
        order.pop_back();
        pos[id] = NONE;
        e.source = e.dest = NONE;
    }
    // ids become 0 .. n-1 in iteration order, tombstones go
    void compact()
    {
        std::vector<size_t> new_id(m.edges.size(), NONE);
        std::vector<MeshEdge> edges;
        edges.reserve(order.size());
        for (size_t i = 0; i < order.size(); i++)
        {
            new_id[order[i]] = i;
            edges.push_back(m.edges[order[i]]);
        }
        m.edges.swap(edges);
        for (auto &set : m.node_edges)
            for (size_t &e : set)
                e = new_id[e];
        m.rebuild_lookup();
    }
};

struct BisectionResult
{
    size_t newVertexId = NONE, newEdgeId = NONE;
};

BisectionResult bisect_edge(Refining &r, size_t edgeId) // :195-353
{
    MeshGraph &m = r.m;
    BisectionResult result;
    if (!alive(m, edgeId))
        return result;
    const MeshEdge edge = m.edges[edgeId];
    const size_t src = edge.source, dst = edge.dest;
    const double *a = m.nodes[src].location, *b = m.nodes[dst].location;
    const size_t mid = m.addNode((a[0] + b[0]) / 2.0, (a[1] + b[1]) / 2.0, (a[2] + b[2]) / 2.0);
    result.newVertexId = mid;
    const size_t opp0 = edge.triangleOppositeNodes[0], opp1 = edge.triangleOppositeNodes[1];
    const bool border = edge.border, two_sided = !border && opp1 != NONE;
    const size_t e_src_opp0 = find_edge_between(m, src, opp0), e_dst_opp0 = find_edge_between(m, dst, opp0);
    size_t e_src_opp1 = NONE, e_dst_opp1 = NONE;
    if (two_sided)
    {
        e_src_opp1 = find_edge_between(m, src, opp1);
        e_dst_opp1 = find_edge_between(m, dst, opp1);
    }
    r.remove_edge(edgeId);
    auto make = [&](bool is_border, size_t o0, size_t o1) {
        MeshEdge e;
        e.border = is_border;
        e.triangleOppositeNodes[0] = o0;
        e.triangleOppositeNodes[1] = o1;
        return e;
    };
    r.add_edge(make(border, opp0, two_sided ? opp1 : NONE), src, mid);
    r.add_edge(make(border, opp0, two_sided ? opp1 : NONE), mid, dst);
    const size_t mid_opp0 = r.add_edge(make(false, src, dst), mid, opp0);
    if (two_sided)
        r.add_edge(make(false, src, dst), mid, opp1);
    auto retarget = [&](size_t e, size_t from) { // the neighbouring edges now face the midpoint (:290-346)
        if (e == NONE)
            return;
        for (int i = 0; i < 2; i++)
            if (m.edges[e].triangleOppositeNodes[i] == from)
            {
                m.edges[e].triangleOppositeNodes[i] = mid;
                break;
            }
    };
    retarget(e_src_opp0, dst);
    retarget(e_dst_opp0, src);
    if (two_sided)
    {
        retarget(e_src_opp1, dst);
        retarget(e_dst_opp1, src);
    }
    result.newEdgeId = mid_opp0;
    return result;
}

size_t refine_triangle(Refining &r, const TriangleId &tri, int maxDepth = 100) // :355-450
{
    MeshGraph &m = r.m;
    size_t created = 0;
    TriangleId current = tri;
    for (;;)
    {
        if (maxDepth <= 0)
            return created;
        size_t v[3];
        if (!triangle_vertices(m, current, v))
            return created;
        const size_t longest = find_longest_edge(m, current);
        if (longest == NONE || !alive(m, longest))
            return created;
        if (!m.edges[longest].border)
        {
            int our_side = -1;
            for (int k = 0; k < 3 && our_side < 0; k++)
                our_side = find_triangle_side(m, longest, v[k]);
            if (our_side >= 0)
            {
                const TriangleId neighbor{longest, 1 - our_side};
                const size_t neighbor_longest = find_longest_edge(m, neighbor);
                if (neighbor_longest != NONE && neighbor_longest != longest)
                {
                    const size_t by_recursion = refine_triangle(r, neighbor, maxDepth - 1);
                    created += by_recursion;
                    if (by_recursion == 0)
                        return created;
                    double c[3] = {0, 0, 0}; // Vector3d center, += location, /= 3
                    for (int i = 0; i < 3; i++)
                        for (int k = 0; k < 3; k++)
                            c[k] += m.nodes[v[i]].location[k];
                    const TriangleId relocated = find_triangle_near_vertices(m, v, c[0] / 3.0, c[1] / 3.0);
                    if (relocated.edgeId == NONE)
                        return created;
                    current = relocated;
                    continue;
                }
            }
        }
        const bool was_border = m.edges[longest].border;
        const BisectionResult res = bisect_edge(r, longest);
        if (res.newVertexId != NONE)
            created += was_border ? 2 : 4;
        return created;
    }
}

TriangleId find_triangle_containing_point(const MeshGraph &m, const std::vector<size_t> &order, double x, double y) // :151-193
{
    for (size_t e : order)
        for (int side = 0; side < 2; side++)
        {
            if (side == 1 && m.edges[e].border)
                continue;
            size_t v[3];
            if (!triangle_vertices(m, TriangleId{e, side}, v))
                continue;
            if (point_in_triangle_2d(x, y, m.nodes[v[0]].location, m.nodes[v[1]].location, m.nodes[v[2]].location))
                return TriangleId{e, side};
        }
    return TriangleId();
}

std::vector<size_t> identity_order(const MeshGraph &m)
{
    std::vector<size_t> order;
    for (size_t e = 0; e < m.edges.size(); e++)
        if (alive(m, e))
            order.push_back(e);
    return order;
}

std::vector<std::pair<TriangleId, TrianglePointStats>> count_points(const MeshGraph &mesh, const std::vector<size_t> &order,
                                                                    const std::vector<point_cloud> &points)
{
    TriangleLocator locator(mesh, &order);
    std::vector<const std::array<double, 3> *> all;
    for (const point_cloud &c : points)
        for (const auto &p : c)
            all.push_back(&p);
    // the triangle of every point (independent, in parallel), then the sums in point order
    std::vector<TriangleId> where(all.size());
#pragma omp parallel for schedule(dynamic, 256)
    for (size_t i = 0; i < all.size(); i++)
        where[i] = locator.find((*all[i])[0], (*all[i])[1]);
    struct Acc
    {
        size_t count = 0;
        double sum = 0, sum_sq = 0;
        bool plane = false;
        double n[3], o[3];
    };
    std::vector<size_t> slot(2 * mesh.edges.size(), NONE);
    std::vector<TriangleId> tris;
    std::vector<Acc> acc;
    for (size_t i = 0; i < all.size(); i++)
    {
        const TriangleId t = where[i];
        if (t.edgeId == NONE)
            continue;
        size_t &s = slot[2 * t.edgeId + t.side];
        if (s == NONE)
        {
            s = acc.size();
            tris.push_back(t);
            Acc a;
            size_t v[3];
            if (triangle_vertices(mesh, t, v)) // planeCache (:735-759): normal = ((n1 - n0) x (n2 - n0)).normalized()
            {
                const double *p0 = mesh.nodes[v[0]].location, *p1 = mesh.nodes[v[1]].location, *p2 = mesh.nodes[v[2]].location;
                const double u[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, w[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
                double n[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
                const double n2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                if (n2 > 0)
                {
                    const double nn = std::sqrt(n2);
                    n[0] /= nn, n[1] /= nn, n[2] /= nn;
                }
                for (int k = 0; k < 3; k++)
                    a.n[k] = n[k], a.o[k] = p0[k];
                a.plane = true;
            }
            acc.push_back(a);
        }
        Acc &a = acc[s];
        a.count++;
        if (a.plane)
        {
            const auto &p = *all[i];
            const double dist = (p[0] - a.o[0]) * a.n[0] + (p[1] - a.o[1]) * a.n[1] + (p[2] - a.o[2]) * a.n[2];
            a.sum += dist;
            a.sum_sq += dist * dist;
        }
    }
    std::vector<std::pair<TriangleId, TrianglePointStats>> result;
    for (size_t i = 0; i < acc.size(); i++)
    {
        TrianglePointStats st;
        st.count = acc[i].count;
        if (acc[i].count > 1)
        {
            const double mean = acc[i].sum / acc[i].count;
            st.distanceVariance = acc[i].sum_sq / acc[i].count - mean * mean;
        }
        result.emplace_back(tris[i], st);
    }
    return result;
}

} // namespace

// ------------------------------------------------------------------------------------------------------ TriangleLocator
TriangleLocator::TriangleLocator(const MeshGraph &m, const std::vector<size_t> *edge_order) : _m(m)
{
    _order = edge_order ? *edge_order : identity_order(m);
    for (size_t e : _order)
        for (int side = 0; side < 2; side++)
        {
            if (side == 1 && m.edges[e].border)
                continue;
            size_t v[3];
            if (!triangle_vertices(m, TriangleId{e, side}, v))
                continue;
            _tri.push_back(TriangleId{e, side});
            _cx.push_back((m.nodes[v[0]].location[0] + m.nodes[v[1]].location[0] + m.nodes[v[2]].location[0]) / 3.0);
            _cy.push_back((m.nodes[v[0]].location[1] + m.nodes[v[1]].location[1] + m.nodes[v[2]].location[1]) / 3.0);
        }
    if (_tri.empty())
        return;
    double x1 = -INFINITY, y1 = -INFINITY;
    _x0 = _y0 = INFINITY;
    for (size_t i = 0; i < _tri.size(); i++)
    {
        _x0 = std::min(_x0, _cx[i]), _y0 = std::min(_y0, _cy[i]);
        x1 = std::max(x1, _cx[i]), y1 = std::max(y1, _cy[i]);
    }
    const double span = std::max(std::max(x1 - _x0, y1 - _y0), 1e-9);
    const int n = std::max(1, std::min(1024, (int)std::sqrt((double)_tri.size())));
    _cell = span / n * (1 + 1e-12);
    _nx = _ny = n;
    _start.assign((size_t)n * n + 1, 0);
    auto cell_of = [&](size_t i) {
        const int cx = std::min(n - 1, std::max(0, (int)std::floor((_cx[i] - _x0) / _cell)));
        const int cy = std::min(n - 1, std::max(0, (int)std::floor((_cy[i] - _y0) / _cell)));
        return (size_t)cy * n + cx;
    };
    for (size_t i = 0; i < _tri.size(); i++)
        _start[cell_of(i) + 1]++;
    for (size_t c = 0; c < (size_t)n * n; c++)
        _start[c + 1] += _start[c];
    _items.resize(_tri.size());
    std::vector<uint32_t> fill(_start.begin(), _start.end() - 1);
    for (size_t i = 0; i < _tri.size(); i++)
        _items[fill[cell_of(i)]++] = (uint32_t)i;
}

bool TriangleLocator::vertices(const TriangleId &t, size_t v[3]) const
{
    return triangle_vertices(_m, t, v);
}

TriangleId TriangleLocator::brute_force(double x, double y) const
{
    return find_triangle_containing_point(_m, _order, x, y);
}

TriangleId TriangleLocator::find(double x, double y) const
{
    if (_tri.empty())
        return TriangleId();
    // nearest centroid (the reference's KD-tree query; equidistant centroids: the one added first): rings of grid cells
    size_t best = 0;
    double bd = INFINITY;
    {
        const long cx = (long)std::floor((x - _x0) / _cell), cy = (long)std::floor((y - _y0) / _cell);
        const long far = std::max(std::max(std::labs(cx), std::labs(cx - (_nx - 1))), std::max(std::labs(cy), std::labs(cy - (_ny - 1))));
        auto visit = [&](long gx, long gy) {
            if (gx < 0 || gy < 0 || gx >= _nx || gy >= _ny)
                return;
            const size_t c = (size_t)gy * _nx + gx;
            for (uint32_t it = _start[c]; it < _start[c + 1]; it++)
            {
                const size_t i = _items[it];
                const double dx = _cx[i] - x, dy = _cy[i] - y, d = dx * dx + dy * dy;
                if (d < bd || (d == bd && i < best))
                {
                    bd = d;
                    best = i;
                }
            }
        };
        for (long r = 0; r <= far; r++)
        {
            if (r == 0)
                visit(cx, cy);
            else
            {
                for (long gx = std::max(0L, cx - r); gx <= std::min((long)_nx - 1, cx + r); gx++)
                {
                    visit(gx, cy - r);
                    visit(gx, cy + r);
                }
                for (long gy = std::max(0L, cy - r + 1); gy <= std::min((long)_ny - 1, cy + r - 1); gy++)
                {
                    visit(cx - r, gy);
                    visit(cx + r, gy);
                }
            }
            if (bd < (double)r * _cell * (double)r * _cell)
                break;
        }
    }
    TriangleId current = _tri[best];
    for (int step = 0; step < 100; step++)
    {
        size_t v[3];
        if (!triangle_vertices(_m, current, v))
            return TriangleId();
        const double *p0 = _m.nodes[v[0]].location, *p1 = _m.nodes[v[1]].location, *p2 = _m.nodes[v[2]].location;
        auto sign = [](double px, double py, double ax, double ay, double bx, double by) {
            return (px - bx) * (ay - by) - (ax - bx) * (py - by);
        };
        const double d[3] = {sign(x, y, p0[0], p0[1], p1[0], p1[1]), sign(x, y, p1[0], p1[1], p2[0], p2[1]),
                             sign(x, y, p2[0], p2[1], p0[0], p0[1])};
        const bool neg = d[0] < 0 || d[1] < 0 || d[2] < 0, pos = d[0] > 0 || d[1] > 0 || d[2] > 0;
        if (!(neg && pos))
            return current;
        const bool expect_positive = ((d[0] < 0) + (d[1] < 0) + (d[2] < 0)) < 2;
        double worst = 0;
        int leave = -1;
        for (int i = 0; i < 3; i++)
        {
            if (d[i] == 0)
            {
                worst = 0.000001;
                leave = i;
            }
            else if ((d[i] > 0) != expect_positive && std::abs(d[i]) > worst)
            {
                worst = std::abs(d[i]);
                leave = i;
            }
        }
        if (leave < 0)
            return TriangleId();
        TriangleId next;
        if (leave == 0)
        {
            if (!_m.edges[current.edgeId].border)
                next = TriangleId{current.edgeId, 1 - current.side};
        }
        else
        {
            const size_t va = v[leave], vb = v[(leave + 1) % 3], opp = v[(leave + 2) % 3];
            const size_t ce = find_edge_between(_m, va, vb);
            if (ce != NONE && !_m.edges[ce].border)
            {
                const int side = find_triangle_side(_m, ce, opp);
                if (side >= 0)
                    next = TriangleId{ce, 1 - side};
            }
        }
        if (next.edgeId == NONE)
            return TriangleId();
        current = next;
    }
    return brute_force(x, y);
}

// ---------------------------------------------------------------------------------------------------------- refinement
std::vector<std::pair<TriangleId, TrianglePointStats>> countPointsPerTriangle(const MeshGraph &mesh, const std::vector<point_cloud> &points)
{
    return count_points(mesh, identity_order(mesh), points);
}

size_t refineByPointDensity(MeshGraph &mesh, const std::vector<point_cloud> &points, size_t maxPointsPerTriangle, double minDistanceVariance,
                            int maxIterations, double minTriangleSizeMeters)
{
    Refining r(mesh);
    size_t total = 0;
    for (int iter = 0; iter < maxIterations; iter++)
    {
        const auto stats = count_points(mesh, r.order, points);
        std::vector<TriangleId> to_refine;
        for (const auto &[tri, s] : stats)
        {
            if (!(s.count > maxPointsPerTriangle && s.distanceVariance > minDistanceVariance))
                continue;
            if (minTriangleSizeMeters > 0.0)
            {
                size_t v[3];
                if (triangle_vertices(mesh, tri, v))
                {
                    auto len = [&](size_t a, size_t b) {
                        const double dx = mesh.nodes[a].location[0] - mesh.nodes[b].location[0], dy = mesh.nodes[a].location[1] - mesh.nodes[b].location[1];
                        return std::sqrt(dx * dx + dy * dy);
                    };
                    if (std::max({len(v[0], v[1]), len(v[1], v[2]), len(v[2], v[0])}) < minTriangleSizeMeters)
                        continue;
                }
            }
            to_refine.push_back(tri);
        }
        if (to_refine.empty())
            break;
        size_t created = 0;
        for (const TriangleId &tri : to_refine)
        {
            if (!alive(mesh, tri.edgeId)) // invalidated by an earlier refinement
                continue;
            created += refine_triangle(r, tri);
        }
        if (created == 0)
            break;
        total += created;
    }
    r.compact();
    return total;
}

size_t refineAtPoint(MeshGraph &mesh, double x, double y, int levels)
{
    Refining r(mesh);
    size_t total = 0;
    for (int level = 0; level < levels; level++)
    {
        const TriangleId tri = find_triangle_containing_point(mesh, r.order, x, y);
        if (tri.edgeId == NONE)
            break;
        const size_t created = refine_triangle(r, tri);
        if (created == 0)
            break;
        total += created;
    }
    r.compact();
    return total;
}

// ---------------------------------------------------------------------------------------------- the MESH_REFINEMENT state
bool mesh_refinement_step(ochip_ctx *ctx, MeasurementGraph &graph, std::vector<surface_model> &surfaces, RelaxStage &stage,
                          MeshRefinementState &state, Transition *transition, std::string *error)
{
    constexpr size_t maxPointsPerTriangle = 20;
    constexpr double varianceGsdMultiplier = 2.0, baseGridFraction = 0.1;
    constexpr int maxIterations = 20; // MESH_REFINEMENT_MAX_ITERATIONS (pipeline.cpp:38)
    *transition = Transition::NEXT;
    auto done = [&](Transition t) {
        *transition = t;
        state.run_count = (t == Transition::REPEAT) ? state.run_count + 1 : 0;
        return true;
    };
    if (state.run_count == 0)
    {
        state.grid_level = 0;
        state.level_triangles = 0;
        point_cloud cams;
        for (const auto &n : graph.nodes())
            if (std::isfinite(n.payload.position[0]) && std::isfinite(n.payload.position[1]) && std::isfinite(n.payload.position[2]))
                cams.push_back({n.payload.position[0], n.payload.position[1], n.payload.position[2]});
        surface_model initial;
        initial.mesh = buildMinimalMesh(cams, surfaces);
        surfaces.clear();
        surfaces.push_back(std::move(initial));
        stage.setSurfaceModels(surfaces);
    }
    const double gridFraction = baseGridFraction / std::pow(2.0, state.grid_level);
    state.grid_fraction = gridFraction;
    RelaxConfig config;
    config.options = OPT_ORIENTATION | OPT_GROUND_MESH;
    config.ground_mesh_grid_fraction = gridFraction;
    stage.init(graph, {}, true, false, config);
    auto runners = stage.get_runners(ctx, graph);
    run_parallel(runners, stage.runner_contexts());
    stage.finalize(graph);
    if (!stage.error().empty())
    {
        if (error)
            *error = stage.error();
        return false;
    }
    surfaces = stage.getSurfaceModels();
    if (surfaces.empty())
        return done(Transition::NEXT);

    double meanSurfaceZ = 0;
    size_t surfNodeCount = 0;
    for (const auto &s : surfaces)
        for (const auto &n : s.mesh.nodes)
        {
            meanSurfaceZ += n.location[2];
            surfNodeCount++;
        }
    if (surfNodeCount > 0)
        meanSurfaceZ /= surfNodeCount;
    double meanCameraZ = 0, meanArcPerPixel = 0, meanImageSize = 0;
    size_t camCount = 0;
    for (const auto &n : graph.nodes())
    {
        const image &p = n.payload;
        if (!p.model || p.model->focal_length_pixels <= 0 ||
            !(std::isfinite(p.position[0]) && std::isfinite(p.position[1]) && std::isfinite(p.position[2])))
            continue;
        meanCameraZ += p.position[2];
        meanArcPerPixel += 1.0 / p.model->focal_length_pixels;
        meanImageSize += static_cast<double>(std::max(p.model->pixels_cols, p.model->pixels_rows));
        camCount++;
    }
    double gsd = 0.01, reducedGsd = 0.0;
    if (camCount > 0)
    {
        meanCameraZ /= camCount;
        meanArcPerPixel /= camCount;
        meanImageSize /= camCount;
        gsd = std::max(0.001, std::abs(meanCameraZ - meanSurfaceZ) * meanArcPerPixel);
        reducedGsd = std::sqrt(static_cast<double>(maxPointsPerTriangle) / 8.0) * gridFraction * meanImageSize * gsd;
    }
    const double minDistanceStddev = varianceGsdMultiplier * gsd, minDistanceVariance = minDistanceStddev * minDistanceStddev;
    state.gsd = gsd;
    state.reduced_gsd = reducedGsd;

    size_t above = 0, maxPoints = 0;
    for (const auto &s : surfaces)
    {
        if (s.mesh.size_nodes() == 0)
            continue;
        for (const auto &[tri, st] : countPointsPerTriangle(s.mesh, s.cloud))
        {
            maxPoints = std::max(maxPoints, st.count);
            if (st.count > maxPointsPerTriangle && st.distanceVariance > minDistanceVariance)
                above++;
        }
    }
    state.triangles_above_threshold = above;
    state.max_points = maxPoints;
    state.refined = 0;
    bool levelConverged = (above == 0);
    if (!levelConverged && state.run_count >= (uint64_t)(maxIterations - 1))
        levelConverged = true;
    if (!levelConverged)
    {
        size_t totalRefined = 0;
        for (auto &s : surfaces)
        {
            if (s.mesh.size_nodes() == 0)
                continue;
            totalRefined += refineByPointDensity(s.mesh, s.cloud, maxPointsPerTriangle, minDistanceVariance, 1, reducedGsd);
        }
        state.refined = totalRefined;
        if (totalRefined == 0)
            levelConverged = true;
        else
        {
            state.level_triangles += totalRefined;
            stage.setSurfaceModels(surfaces);
            return done(Transition::REPEAT);
        }
    }
    if (state.level_triangles == 0)
        return done(Transition::NEXT);
    state.grid_level++;
    state.level_triangles = 0;
    stage.setSurfaceModels(surfaces);
    return done(Transition::REPEAT);
}

} // namespace opencalibration_amd
