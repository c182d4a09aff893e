// ORACLE — test infrastructure only (used by tests/; never by the product path).
//
// Second restatement of the reference's mesh refinement, written from the reference's text and independent of
// opencalibration_amd/csrc/host/refine_mesh.cpp:
//   /root/reference/src/surface/refine_mesh.cpp:15-123   edge / triangle helpers
//                                              :125-193  findLongestEdge, findTriangleContainingPoint
//                                              :195-351  bisectEdge
//                                              :353-449  refineTriangle (conforming longest-edge bisection, neighbour first)
//                                              :451-475  refineAtPoint
//                                              :569-712  TriangleLocator (nearest centroid, then a walk of <= 100 steps)
//                                              :713-825  countPointsPerTriangle
//                                              :827-909  refineByPointDensity
// The mesh lives in an emulation of the reference's DirectedGraph<MeshNode, MeshEdge> (include/opencalibration/types/
// graph.hpp): identifiers are drawn, 0 means "none", and every container is an ankerl::unordered_dense one, i.e. it is
// iterated in insertion order and an erase moves the last element into the hole (external/unordered_dense,
// :1150-1166; pinned against the real header by tests/test_oracle_ref_pins.py).  The refinement's choices depend on
// those orders.  Identifiers here are a counter from 1 (the reference draws random 64-bit ids; nothing depends on their
// values, only on which is which).  The locator's nearest-centroid query is exhaustive (first inserted wins a tie; the
// reference's KD-tree returns one of the nearest); countPointsPerTriangle runs on one thread, so its map is in
// first-point order (the reference merges per-thread maps in completion order).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace
{

// ankerl::unordered_dense::set / map as far as the reference uses them: dense storage in insertion order, erase by
// moving the last element into the hole
template <typename K, typename V> struct dense_map
{
    std::vector<std::pair<K, V>> items;
    std::unordered_map<K, size_t> index;
    V *find(const K &k)
    {
        auto it = index.find(k);
        return it == index.end() ? nullptr : &items[it->second].second;
    }
    const V *find(const K &k) const
    {
        auto it = index.find(k);
        return it == index.end() ? nullptr : &items[it->second].second;
    }
    V &emplace(const K &k, V v)
    {
        auto it = index.find(k);
        if (it != index.end())
            return items[it->second].second;
        index.emplace(k, items.size());
        items.emplace_back(k, std::move(v));
        return items.back().second;
    }
    bool erase(const K &k)
    {
        auto it = index.find(k);
        if (it == index.end())
            return false;
        const size_t at = it->second;
        index.erase(it);
        if (at + 1 != items.size())
        {
            items[at] = std::move(items.back());
            index[items[at].first] = at;
        }
        items.pop_back();
        return true;
    }
};
struct dense_set
{
    dense_map<size_t, char> m;
    void insert(size_t k)
    {
        m.emplace(k, 0);
    }
    void erase(size_t k)
    {
        m.erase(k);
    }
};

struct vec3
{
    double x = 0, y = 0, z = 0;
};
struct Node
{
    vec3 location;
    dense_set edges; // Node::getEdges()
};
struct Edge
{
    size_t source = 0, dest = 0;
    bool border = false;
    std::array<size_t, 2> triangleOppositeNodes{{0, 0}};
};
struct TriangleId
{
    size_t edgeId = 0;
    int side = 0;
    bool operator==(const TriangleId &o) const
    {
        return edgeId == o.edgeId && side == o.side;
    }
};
struct TriangleIdHash
{
    size_t operator()(const TriangleId &t) const
    {
        return std::hash<size_t>()(t.edgeId * 2 + (size_t)t.side);
    }
};

struct Mesh // DirectedGraph<MeshNode, MeshEdge>
{
    dense_map<size_t, Node> nodes;
    dense_map<size_t, Edge> edges;
    size_t next_id = 1;
    size_t addNode(const vec3 &p)
    {
        const size_t id = next_id++;
        Node n;
        n.location = p;
        nodes.emplace(id, std::move(n));
        return id;
    }
    size_t addEdge(Edge e, size_t source, size_t dest)
    {
        const size_t id = next_id++;
        e.source = source;
        e.dest = dest;
        edges.emplace(id, e);
        nodes.find(source)->edges.insert(id);
        nodes.find(dest)->edges.insert(id);
        return id;
    }
    bool removeEdge(size_t id) // graph.hpp:183-213
    {
        const Edge *e = edges.find(id);
        if (!e)
            return false;
        const size_t s = e->source, d = e->dest;
        if (Node *n = nodes.find(s))
            n->edges.erase(id);
        if (Node *n = nodes.find(d))
            n->edges.erase(id);
        edges.erase(id);
        return true;
    }
    const Node *getNode(size_t id) const
    {
        return nodes.find(id);
    }
    const Edge *getEdge(size_t id) const
    {
        return edges.find(id);
    }
    Edge *getEdge(size_t id)
    {
        return edges.find(id);
    }
};

using point_cloud = std::vector<vec3>;

// ---- refine_mesh.cpp:15-123 --------------------------------------------------------------------------------------
double edgeLengthSquared(const Mesh &mesh, size_t edgeId)
{
    const Edge *edge = mesh.getEdge(edgeId);
    if (!edge)
        return 0;
    const Node *a = mesh.getNode(edge->source), *b = mesh.getNode(edge->dest);
    if (!a || !b)
        return 0;
    const double dx = a->location.x - b->location.x, dy = a->location.y - b->location.y;
    return dx * dx + dy * dy; // (head<2>() difference).squaredNorm()
}

size_t findEdgeBetween(const Mesh &mesh, size_t node1, size_t node2)
{
    const Node *node = mesh.getNode(node1);
    if (node)
        for (const auto &kv : node->edges.m.items)
        {
            const Edge *e = mesh.getEdge(kv.first);
            if (e && ((e->source == node1 && e->dest == node2) || (e->source == node2 && e->dest == node1)))
                return kv.first;
        }
    return 0;
}

bool pointInTriangle2D(double px, double py, const vec3 &v0, const vec3 &v1, const vec3 &v2)
{
    auto sign = [](double p1x, double p1y, double p2x, double p2y, double p3x, double p3y) {
        return (p1x - p3x) * (p2y - p3y) - (p2x - p3x) * (p1y - p3y);
    };
    const double d1 = sign(px, py, v0.x, v0.y, v1.x, v1.y);
    const double d2 = sign(px, py, v1.x, v1.y, v2.x, v2.y);
    const double d3 = sign(px, py, v2.x, v2.y, v0.x, v0.y);
    const bool hasNeg = (d1 < 0) || (d2 < 0) || (d3 < 0);
    const bool hasPos = (d1 > 0) || (d2 > 0) || (d3 > 0);
    return !(hasNeg && hasPos);
}

int findTriangleSide(const Mesh &mesh, size_t edgeId, size_t oppositeVertex)
{
    const Edge *edge = mesh.getEdge(edgeId);
    if (!edge)
        return -1;
    if (edge->triangleOppositeNodes[0] == oppositeVertex)
        return 0;
    if (edge->triangleOppositeNodes[1] == oppositeVertex)
        return 1;
    return -1;
}

std::array<size_t, 3> getTriangleVertices(const Mesh &mesh, const TriangleId &tri)
{
    const Edge *edge = mesh.getEdge(tri.edgeId);
    if (!edge)
        return {0, 0, 0};
    return {edge->source, edge->dest, edge->triangleOppositeNodes[tri.side]};
}

TriangleId findTriangleNearVertices(const Mesh &mesh, const std::array<size_t, 3> &vertices, double x, double y)
{
    for (size_t vtxId : vertices)
    {
        const Node *vtx = mesh.getNode(vtxId);
        if (!vtx)
            continue;
        for (const auto &kv : vtx->edges.m.items)
            for (int side = 0; side < 2; side++)
            {
                const TriangleId candidate{kv.first, side};
                const auto cv = getTriangleVertices(mesh, candidate);
                if (cv[0] == 0 && cv[1] == 0 && cv[2] == 0)
                    continue;
                const Node *n0 = mesh.getNode(cv[0]), *n1 = mesh.getNode(cv[1]), *n2 = mesh.getNode(cv[2]);
                if (!n0 || !n1 || !n2)
                    continue;
                if (pointInTriangle2D(x, y, n0->location, n1->location, n2->location))
                    return candidate;
            }
    }
    return {0, 0};
}

// ---- :125-193 ----------------------------------------------------------------------------------------------------
size_t findLongestEdge(const Mesh &mesh, const TriangleId &tri)
{
    const auto v = getTriangleVertices(mesh, tri);
    if (v[0] == 0 && v[1] == 0 && v[2] == 0)
        return 0;
    std::array<std::pair<size_t, double>, 3> edges;
    edges[0] = {tri.edgeId, edgeLengthSquared(mesh, tri.edgeId)};
    const size_t e1 = findEdgeBetween(mesh, v[1], v[2]);
    edges[1] = {e1, e1 ? edgeLengthSquared(mesh, e1) : 0};
    const size_t e2 = findEdgeBetween(mesh, v[2], v[0]);
    edges[2] = {e2, e2 ? edgeLengthSquared(mesh, e2) : 0};
    size_t longest = 0;
    for (size_t i = 1; i < 3; i++)
        if (edges[i].second > edges[longest].second)
            longest = i;
    return edges[longest].first;
}

TriangleId findTriangleContainingPoint(const Mesh &mesh, double x, double y)
{
    for (const auto &kv : mesh.edges.items)
    {
        const size_t edgeId = kv.first;
        for (int side = 0; side < 2; side++)
        {
            if (side == 1 && kv.second.border)
                continue;
            const auto v = getTriangleVertices(mesh, {edgeId, side});
            if (v[0] != 0 || v[1] != 0 || v[2] != 0)
            {
                const Node *n0 = mesh.getNode(v[0]), *n1 = mesh.getNode(v[1]), *n2 = mesh.getNode(v[2]);
                if (n0 && n1 && n2 && pointInTriangle2D(x, y, n0->location, n1->location, n2->location))
                    return {edgeId, side};
            }
        }
    }
    return {0, 0};
}

// ---- :195-351 ----------------------------------------------------------------------------------------------------
struct BisectionResult
{
    size_t newVertexId = 0, newEdgeId = 0;
};

BisectionResult bisectEdge(Mesh &mesh, size_t edgeId)
{
    BisectionResult result;
    Edge *edge = mesh.getEdge(edgeId);
    if (!edge)
        return result;
    const size_t srcId = edge->source, dstId = edge->dest;
    const Node *src = mesh.getNode(srcId), *dst = mesh.getNode(dstId);
    if (!src || !dst)
        return result;
    vec3 mid;
    mid.x = (src->location.x + dst->location.x) / 2.0;
    mid.y = (src->location.y + dst->location.y) / 2.0;
    mid.z = (src->location.z + dst->location.z) / 2.0;
    const size_t opp0 = edge->triangleOppositeNodes[0], opp1 = edge->triangleOppositeNodes[1];
    const bool isBorder = edge->border;
    const size_t midId = mesh.addNode(mid); // (invalidates `edge`, `src`, `dst`: not used below)
    result.newVertexId = midId;

    const size_t edgeSrcOpp0 = findEdgeBetween(mesh, srcId, opp0), edgeDstOpp0 = findEdgeBetween(mesh, dstId, opp0);
    size_t edgeSrcOpp1 = 0, edgeDstOpp1 = 0;
    const bool two = !isBorder && opp1 != 0;
    if (two)
    {
        edgeSrcOpp1 = findEdgeBetween(mesh, srcId, opp1);
        edgeDstOpp1 = findEdgeBetween(mesh, dstId, opp1);
    }
    mesh.removeEdge(edgeId);

    Edge e;
    e.border = isBorder;
    const size_t srcMidId = mesh.addEdge(e, srcId, midId);
    const size_t midDstId = mesh.addEdge(e, midId, dstId);
    Edge inner;
    inner.border = false;
    const size_t midOpp0Id = mesh.addEdge(inner, midId, opp0);
    size_t midOpp1Id = 0;
    if (two)
        midOpp1Id = mesh.addEdge(inner, midId, opp1);

    for (size_t id : {srcMidId, midDstId})
    {
        Edge *p = mesh.getEdge(id);
        p->triangleOppositeNodes[0] = opp0;
        if (two)
            p->triangleOppositeNodes[1] = opp1;
    }
    {
        Edge *p = mesh.getEdge(midOpp0Id);
        p->triangleOppositeNodes[0] = srcId;
        p->triangleOppositeNodes[1] = dstId;
    }
    if (two && midOpp1Id != 0)
    {
        Edge *p = mesh.getEdge(midOpp1Id);
        p->triangleOppositeNodes[0] = srcId;
        p->triangleOppositeNodes[1] = dstId;
    }
    auto repoint = [&](size_t id, size_t from) {
        if (!id)
            return;
        if (Edge *p = mesh.getEdge(id))
            for (int i = 0; i < 2; i++)
                if (p->triangleOppositeNodes[i] == from)
                {
                    p->triangleOppositeNodes[i] = midId;
                    break;
                }
    };
    repoint(edgeSrcOpp0, dstId);
    repoint(edgeDstOpp0, srcId);
    if (two)
    {
        repoint(edgeSrcOpp1, dstId);
        repoint(edgeDstOpp1, srcId);
    }
    result.newEdgeId = midOpp0Id;
    return result;
}

// ---- :353-449 ----------------------------------------------------------------------------------------------------
size_t refineTriangle(Mesh &mesh, const TriangleId &tri, int maxDepth = 10)
{
    size_t created = 0;
    TriangleId current = tri;
    for (;;)
    {
        if (maxDepth <= 0)
            return created;
        const auto vertices = getTriangleVertices(mesh, current);
        if (vertices[0] == 0 && vertices[1] == 0 && vertices[2] == 0)
            return created;
        const size_t longestEdgeId = findLongestEdge(mesh, current);
        if (longestEdgeId == 0)
            return created;
        const Edge *longestEdge = mesh.getEdge(longestEdgeId);
        if (!longestEdge)
            return created;
        if (!longestEdge->border)
        {
            int ourSide = -1;
            for (int v = 0; v < 3; v++)
            {
                const int side = findTriangleSide(mesh, longestEdgeId, vertices[v]);
                if (side >= 0)
                {
                    ourSide = side;
                    break;
                }
            }
            if (ourSide >= 0)
            {
                const TriangleId neighbor{longestEdgeId, 1 - ourSide};
                const size_t neighborLongest = findLongestEdge(mesh, neighbor);
                if (neighborLongest != 0 && neighborLongest != longestEdgeId)
                {
                    const size_t byRecursion = refineTriangle(mesh, neighbor, maxDepth - 1);
                    created += byRecursion;
                    if (byRecursion == 0)
                        return created;
                    vec3 center;
                    for (int i = 0; i < 3; i++)
                        if (const Node *n = mesh.getNode(vertices[i]))
                        {
                            center.x += n->location.x;
                            center.y += n->location.y;
                            center.z += n->location.z;
                        }
                    center.x /= 3.0;
                    center.y /= 3.0;
                    const TriangleId newTri = findTriangleNearVertices(mesh, vertices, center.x, center.y);
                    if (newTri.edgeId == 0)
                        return created;
                    current = newTri;
                    continue;
                }
            }
        }
        const bool wasBorder = longestEdge->border;
        const BisectionResult r = bisectEdge(mesh, longestEdgeId);
        if (r.newVertexId != 0)
            created += wasBorder ? 2 : 4;
        return created;
    }
}

size_t refineAtPoint(Mesh &mesh, double x, double y, int levels)
{
    size_t total = 0;
    for (int level = 0; level < levels; level++)
    {
        const TriangleId tri = findTriangleContainingPoint(mesh, x, y);
        if (tri.edgeId == 0)
            break;
        const size_t created = refineTriangle(mesh, tri);
        if (created == 0)
            break;
        total += created;
    }
    return total;
}

// ---- :569-712 ----------------------------------------------------------------------------------------------------
struct TriangleLocator
{
    const Mesh &mesh;
    std::vector<std::array<double, 2>> centroid;
    std::vector<TriangleId> payload;
    explicit TriangleLocator(const Mesh &m) : mesh(m)
    {
        for (const auto &kv : mesh.edges.items)
            for (int side = 0; side < 2; side++)
            {
                if (side == 1 && kv.second.border)
                    continue;
                const TriangleId tri{kv.first, side};
                const auto v = getTriangleVertices(mesh, tri);
                if (v[0] == 0 && v[1] == 0 && v[2] == 0)
                    continue;
                const Node *n0 = mesh.getNode(v[0]), *n1 = mesh.getNode(v[1]), *n2 = mesh.getNode(v[2]);
                if (!n0 || !n1 || !n2)
                    continue;
                centroid.push_back({(n0->location.x + n1->location.x + n2->location.x) / 3.0,
                                    (n0->location.y + n1->location.y + n2->location.y) / 3.0});
                payload.push_back(tri);
            }
    }
    TriangleId find(double x, double y) const
    {
        if (centroid.empty())
            return {0, 0};
        size_t best = 0;
        double bestd = INFINITY;
        for (size_t i = 0; i < centroid.size(); i++)
        {
            const double dx = x - centroid[i][0], dy = y - centroid[i][1];
            double d = 0; // SquaredL2 of jk::KDTree accumulates per dimension
            d += dx * dx;
            d += dy * dy;
            if (d < bestd)
            {
                bestd = d;
                best = i;
            }
        }
        TriangleId current = payload[best];
        for (int step = 0; step < 100; step++)
        {
            const auto v = getTriangleVertices(mesh, current);
            if (v[0] == 0 && v[1] == 0 && v[2] == 0)
                return {0, 0};
            const Node *n0 = mesh.getNode(v[0]), *n1 = mesh.getNode(v[1]), *n2 = mesh.getNode(v[2]);
            if (!n0 || !n1 || !n2)
                return {0, 0};
            const vec3 &p0 = n0->location, &p1 = n1->location, &p2 = n2->location;
            auto sign = [](double px, double py, double ax, double ay, double bx, double by) {
                return (px - bx) * (ay - by) - (ax - bx) * (py - by);
            };
            const double d0 = sign(x, y, p0.x, p0.y, p1.x, p1.y);
            const double d1 = sign(x, y, p1.x, p1.y, p2.x, p2.y);
            const double d2 = sign(x, y, p2.x, p2.y, p0.x, p0.y);
            const bool hasNeg = (d0 < 0) || (d1 < 0) || (d2 < 0);
            const bool hasPos = (d0 > 0) || (d1 > 0) || (d2 > 0);
            if (!(hasNeg && hasPos))
                return current;
            const int negCount = (d0 < 0) + (d1 < 0) + (d2 < 0);
            const bool expectPositive = negCount < 2;
            double worstVal = 0;
            int worstEdge = -1;
            auto checkEdge = [&](int idx, double d) {
                if (d == 0)
                {
                    worstVal = 0.000001;
                    worstEdge = idx;
                    return;
                }
                if ((d > 0) != expectPositive && std::abs(d) > worstVal)
                {
                    worstVal = std::abs(d);
                    worstEdge = idx;
                }
            };
            checkEdge(0, d0);
            checkEdge(1, d1);
            checkEdge(2, d2);
            if (worstEdge < 0)
                return {0, 0};
            TriangleId neighbor{0, 0};
            if (worstEdge == 0)
            {
                const Edge *edge = mesh.getEdge(current.edgeId);
                if (edge && !edge->border)
                    neighbor = {current.edgeId, 1 - current.side};
            }
            else
            {
                const size_t va = v[worstEdge], vb = v[(worstEdge + 1) % 3];
                const size_t crossEdgeId = findEdgeBetween(mesh, va, vb);
                if (crossEdgeId != 0)
                {
                    const Edge *crossEdge = mesh.getEdge(crossEdgeId);
                    if (crossEdge && !crossEdge->border)
                    {
                        const int currentSide = findTriangleSide(mesh, crossEdgeId, v[(worstEdge + 2) % 3]);
                        if (currentSide >= 0)
                            neighbor = {crossEdgeId, 1 - currentSide};
                    }
                }
            }
            if (neighbor.edgeId == 0)
                return {0, 0};
            current = neighbor;
        }
        return findTriangleContainingPoint(mesh, x, y);
    }
};

// ---- :713-825 ----------------------------------------------------------------------------------------------------
struct TrianglePointStats
{
    size_t count = 0;
    double distanceVariance = 0;
};
using stats_map = dense_map<TriangleId, TrianglePointStats>;
} // namespace
namespace std
{
template <> struct hash<TriangleId>
{
    size_t operator()(const TriangleId &t) const
    {
        return TriangleIdHash()(t);
    }
};
} // namespace std
namespace
{

stats_map countPointsPerTriangle(const Mesh &mesh, const std::vector<point_cloud> &points)
{
    struct Accumulator
    {
        size_t count = 0;
        double sumDist = 0, sumDistSq = 0;
    };
    struct TrianglePlane
    {
        vec3 normal, origin;
    };
    TriangleLocator locator(mesh);
    dense_map<TriangleId, TrianglePlane> planeCache;
    for (const auto &kv : mesh.edges.items)
        for (int side = 0; side < 2; side++)
        {
            if (side == 1 && kv.second.border)
                continue;
            const TriangleId tri{kv.first, side};
            const auto v = getTriangleVertices(mesh, tri);
            if (v[0] == 0 && v[1] == 0 && v[2] == 0)
                continue;
            const Node *n0 = mesh.getNode(v[0]), *n1 = mesh.getNode(v[1]), *n2 = mesh.getNode(v[2]);
            if (n0 && n1 && n2)
            {
                const vec3 a{n1->location.x - n0->location.x, n1->location.y - n0->location.y, n1->location.z - n0->location.z};
                const vec3 b{n2->location.x - n0->location.x, n2->location.y - n0->location.y, n2->location.z - n0->location.z};
                vec3 c{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
                // Eigen's normalized(): divide by sqrt(squaredNorm()) when that is positive
                const double n2sq = c.x * c.x + c.y * c.y + c.z * c.z;
                if (n2sq > 0)
                {
                    const double nn = std::sqrt(n2sq);
                    c.x /= nn;
                    c.y /= nn;
                    c.z /= nn;
                }
                planeCache.emplace(tri, TrianglePlane{c, n0->location});
            }
        }
    dense_map<TriangleId, Accumulator> acc;
    for (const point_cloud &cloud : points)
        for (const vec3 &p : cloud)
        {
            const TriangleId tri = locator.find(p.x, p.y);
            if (tri.edgeId == 0)
                continue;
            Accumulator &a = acc.emplace(tri, Accumulator());
            a.count++;
            if (const TrianglePlane *pl = planeCache.find(tri))
            {
                const double dist = (p.x - pl->origin.x) * pl->normal.x + (p.y - pl->origin.y) * pl->normal.y +
                                    (p.z - pl->origin.z) * pl->normal.z;
                a.sumDist += dist;
                a.sumDistSq += dist * dist;
            }
        }
    stats_map result;
    for (const auto &kv : acc.items)
    {
        TrianglePointStats s;
        s.count = kv.second.count;
        if (kv.second.count > 1)
        {
            const double mean = kv.second.sumDist / kv.second.count;
            s.distanceVariance = kv.second.sumDistSq / kv.second.count - mean * mean;
        }
        result.emplace(kv.first, s);
    }
    return result;
}

// ---- :827-909 ----------------------------------------------------------------------------------------------------
size_t refineByPointDensity(Mesh &mesh, const std::vector<point_cloud> &points, size_t maxPointsPerTriangle,
                            double minDistanceVariance, int maxIterations, double minTriangleSizeMeters)
{
    size_t total = 0;
    for (int iter = 0; iter < maxIterations; iter++)
    {
        const stats_map stats = countPointsPerTriangle(mesh, points);
        std::vector<TriangleId> toRefine;
        for (const auto &kv : stats.items)
        {
            const TrianglePointStats &s = kv.second;
            if (s.count > maxPointsPerTriangle && s.distanceVariance > minDistanceVariance)
            {
                if (minTriangleSizeMeters > 0.0)
                {
                    const auto v = getTriangleVertices(mesh, kv.first);
                    const Node *n0 = mesh.getNode(v[0]), *n1 = mesh.getNode(v[1]), *n2 = mesh.getNode(v[2]);
                    if (n0 && n1 && n2)
                    {
                        auto len = [](const vec3 &a, const vec3 &b) {
                            const double dx = a.x - b.x, dy = a.y - b.y;
                            return std::sqrt(dx * dx + dy * dy);
                        };
                        const double maxEdge = std::max({len(n0->location, n1->location), len(n1->location, n2->location),
                                                         len(n2->location, n0->location)});
                        if (maxEdge < minTriangleSizeMeters)
                            continue;
                    }
                }
                toRefine.push_back(kv.first);
            }
        }
        if (toRefine.empty())
            break;
        size_t createdThisIter = 0;
        for (const TriangleId &tri : toRefine)
        {
            const auto v = getTriangleVertices(mesh, tri);
            if (v[0] == 0 && v[1] == 0 && v[2] == 0)
                continue;
            createdThisIter += refineTriangle(mesh, tri);
        }
        if (createdThisIter == 0)
            break;
        total += createdThisIter;
    }
    return total;
}

// ---- a mesh handle for the tests: vertices are reported by their creation number (0, 1, ...)
struct rmesh
{
    Mesh mesh;
    std::vector<size_t> vertex_ids;                 // creation order
    std::unordered_map<size_t, uint64_t> vertex_no; // id -> creation number
    void note_new_vertices()
    {
        for (const auto &kv : mesh.nodes.items) // nodes are never erased: insertion order = creation order
            if (!vertex_no.count(kv.first))
            {
                vertex_no.emplace(kv.first, vertex_ids.size());
                vertex_ids.push_back(kv.first);
            }
    }
};

std::vector<point_cloud> clouds_from(size_t n_clouds, const uint64_t *sizes, const double *xyz)
{
    std::vector<point_cloud> clouds(n_clouds);
    size_t k = 0;
    for (size_t c = 0; c < n_clouds; c++)
    {
        clouds[c].resize(sizes[c]);
        for (size_t i = 0; i < sizes[c]; i++, k++)
            clouds[c][i] = vec3{xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2]};
    }
    return clouds;
}

} // namespace

extern "C"
{

// edges5 rows {source, dest, border, opposite 0, opposite 1} with vertex numbers, UINT64_MAX = none: the mesh is built by
// addNode / addEdge in the given order (what buildMinimalMesh / rebuildMesh do), so its containers have the order the
// reference's would have
void *ocx_rmesh_create(const double *vertices_xyz, size_t n_vertices, const uint64_t *edges5, size_t n_edges)
{
    auto *m = new rmesh();
    for (size_t v = 0; v < n_vertices; v++)
        m->mesh.addNode(vec3{vertices_xyz[3 * v], vertices_xyz[3 * v + 1], vertices_xyz[3 * v + 2]});
    m->note_new_vertices();
    for (size_t e = 0; e < n_edges; e++)
    {
        const uint64_t *r = edges5 + 5 * e;
        Edge ed;
        ed.border = r[2] != 0;
        ed.triangleOppositeNodes[0] = r[3] == UINT64_MAX ? 0 : m->vertex_ids[r[3]];
        ed.triangleOppositeNodes[1] = r[4] == UINT64_MAX ? 0 : m->vertex_ids[r[4]];
        m->mesh.addEdge(ed, m->vertex_ids[r[0]], m->vertex_ids[r[1]]);
    }
    return m;
}

void ocx_rmesh_destroy(void *h)
{
    delete (rmesh *)h;
}

void ocx_rmesh_counts(void *h, uint64_t *n_vertices, uint64_t *n_edges)
{
    auto *m = (rmesh *)h;
    *n_vertices = m->mesh.nodes.items.size();
    *n_edges = m->mesh.edges.items.size();
}

// vertices in creation order; edges in the container's iteration order
void ocx_rmesh_get(void *h, double *vertices_xyz, uint64_t *edges5)
{
    auto *m = (rmesh *)h;
    for (size_t v = 0; v < m->vertex_ids.size(); v++)
    {
        const Node *n = m->mesh.getNode(m->vertex_ids[v]);
        vertices_xyz[3 * v] = n->location.x;
        vertices_xyz[3 * v + 1] = n->location.y;
        vertices_xyz[3 * v + 2] = n->location.z;
    }
    size_t e = 0;
    for (const auto &kv : m->mesh.edges.items)
    {
        uint64_t *r = edges5 + 5 * (e++);
        r[0] = m->vertex_no.at(kv.second.source);
        r[1] = m->vertex_no.at(kv.second.dest);
        r[2] = kv.second.border ? 1 : 0;
        for (int i = 0; i < 2; i++)
            r[3 + i] = kv.second.triangleOppositeNodes[i] == 0 ? UINT64_MAX : m->vertex_no.at(kv.second.triangleOppositeNodes[i]);
    }
}

void ocx_rmesh_set_heights(void *h, const double *z)
{
    auto *m = (rmesh *)h;
    for (size_t v = 0; v < m->vertex_ids.size(); v++)
        m->mesh.nodes.find(m->vertex_ids[v])->location.z = z[v];
}

// rows (three vertex numbers; count, variance) in the order the triangles first receive a point; returns the rows
size_t ocx_rmesh_count_points(void *h, size_t n_clouds, const uint64_t *sizes, const double *xyz, uint64_t *tri3, double *stats2,
                              size_t cap)
{
    auto *m = (rmesh *)h;
    const stats_map stats = countPointsPerTriangle(m->mesh, clouds_from(n_clouds, sizes, xyz));
    size_t k = 0;
    for (const auto &kv : stats.items)
    {
        if (k >= cap)
            break;
        const auto v = getTriangleVertices(m->mesh, kv.first);
        for (int i = 0; i < 3; i++)
            tri3[3 * k + i] = m->vertex_no.at(v[i]);
        stats2[2 * k] = (double)kv.second.count;
        stats2[2 * k + 1] = kv.second.distanceVariance;
        k++;
    }
    return k;
}

size_t ocx_rmesh_refine_by_point_density(void *h, size_t n_clouds, const uint64_t *sizes, const double *xyz,
                                         size_t max_points_per_triangle, double min_distance_variance, int max_iterations,
                                         double min_triangle_size)
{
    auto *m = (rmesh *)h;
    const size_t created = refineByPointDensity(m->mesh, clouds_from(n_clouds, sizes, xyz), max_points_per_triangle,
                                                min_distance_variance, max_iterations, min_triangle_size);
    m->note_new_vertices();
    return created;
}

size_t ocx_rmesh_refine_at_point(void *h, double x, double y, int levels)
{
    auto *m = (rmesh *)h;
    const size_t created = refineAtPoint(m->mesh, x, y, levels);
    m->note_new_vertices();
    return created;
}

} // extern "C"
