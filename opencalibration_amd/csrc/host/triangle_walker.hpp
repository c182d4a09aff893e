// MeshIntersectionSearcher of the reference as a header (used by the mesh relax set-up and by dense guided matching).
#pragma once

#include "relax_mesh.hpp"
#include "relax_util.hpp"

namespace opencalibration_amd
{
namespace relax_detail
{

// MeshIntersectionSearcher (src/surface/intersect.cpp:10-163): the triangle of the mesh a ray hits, found by walking
// from the triangle of the previous query across the edge that separates it from the hit point, at most 100 steps (a
// walk that runs out of steps reports the triangle it stands on, as the reference does).
class TriangleWalker
{
  public:
    enum Result
    {
        INTERSECTION,
        OUTSIDE_BORDER,
        RAY_PARALLEL_TO_PLANE,
        GRAPH_STRUCTURE_INCONSISTENT
    };
    bool init(const MeshGraph &mesh)
    {
        _mesh = &mesh;
        if (mesh.size_nodes() == 0 || mesh.size_edges() == 0)
        {
            _mesh = nullptr;
            return false;
        }
        const MeshEdge &e = mesh.edges[0];
        tri[0] = e.source;
        tri[1] = e.dest;
        tri[2] = e.triangleOppositeNodes[0];
        for (int i = 0; i < 3; i++)
            if (tri[i] >= mesh.nodes.size())
                return false;
        return true;
    }
    // ray from `origin` in direction `dir`
    Result find(const v3 &dir, const v3 &origin)
    {
        v3 c[3];
        for (int i = 0; i < 3; i++)
            c[i] = corner(tri[i]);
        for (size_t steps = 0;;)
        {
            if (turns_clockwise(c[0], c[1], c[2]))
            {
                std::swap(c[0], c[1]);
                std::swap(tri[0], tri[1]);
            }
            // plane of the triangle: normal (c0 - c1) x (c0 - c2), normalised as Eigen does
            v3 nrm = cross(sub(c[0], c[1]), sub(c[0], c[2]));
            const double n2 = dot(nrm, nrm);
            if (n2 > 0)
            {
                const double nn = std::sqrt(n2);
                nrm = v3{nrm.x / nn, nrm.y / nn, nrm.z / nn};
            }
            const double denom = dot(nrm, dir);
            if (std::abs(denom) < 1e-9)
                return RAY_PARALLEL_TO_PLANE;
            const double t = (dot(nrm, c[0]) - dot(origin, nrm)) / denom;
            hit = add(origin, mul(dir, t));
            if (std::isnan(hit.x) || std::isnan(hit.y) || std::isnan(hit.z))
                return RAY_PARALLEL_TO_PLANE;
            int leave = -1;
            for (int i = 0; i < 3 && leave < 0; i++)
                if (turns_clockwise(hit, c[i], c[(i + 1) % 3]))
                    leave = i;
            if (leave < 0)
                return INTERSECTION;
            const size_t k0 = tri[leave], k1 = tri[(leave + 1) % 3];
            const MeshEdge *e = _mesh->getEdge(k0, k1);
            if (!e)
                e = _mesh->getEdge(k1, k0);
            if (!e)
                return GRAPH_STRUCTURE_INCONSISTENT;
            if (e->border)
                return OUTSIDE_BORDER;
            const int far = (leave + 2) % 3;
            if (e->triangleOppositeNodes[0] == tri[far])
                tri[far] = e->triangleOppositeNodes[1];
            else if (e->triangleOppositeNodes[1] == tri[far])
                tri[far] = e->triangleOppositeNodes[0];
            else
                return GRAPH_STRUCTURE_INCONSISTENT;
            if (tri[far] >= _mesh->nodes.size())
                return GRAPH_STRUCTURE_INCONSISTENT;
            c[far] = corner(tri[far]);
            if (++steps > 100)
                return INTERSECTION;
        }
    }
    size_t tri[3] = {0, 0, 0};
    v3 hit{NAN, NAN, NAN};

  private:
    v3 corner(size_t i) const
    {
        const double *l = _mesh->nodes[i].location;
        return v3{l[0], l[1], l[2]};
    }
    static bool turns_clockwise(const v3 &a, const v3 &b, const v3 &c) // geometry/utils.hpp: cross(b - a, c - a).z < 0
    {
        return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x) < 0;
    }
    const MeshGraph *_mesh = nullptr;
};

} // namespace relax_detail
} // namespace opencalibration_amd
