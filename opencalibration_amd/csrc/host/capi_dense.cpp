// liboc_host.so: C ABI of dense guided matching (dense_stereo.hpp).
#include "../../../include/oc_host.h"

#include "capi_graph.hpp"
#include "dense_stereo.hpp"

using namespace opencalibration_amd;

extern "C"
{

int och_densify_mesh(och_graph *g, ochip_ctx *ctx, och_surface *surface, double *stats10, uint64_t *match_pairs, size_t match_cap)
{
    std::vector<surface_model> surfaces{std::move(surface->s)};
    DenseStats st;
    std::string why;
    std::vector<std::pair<size_t, size_t>> matches;
    const bool ok = densifyMesh(ctx, g->graph, surfaces, &st, &why, match_pairs ? &matches : nullptr);
    surface->s = std::move(surfaces[0]);
    if (stats10)
    {
        const double v[10] = {(double)st.images, (double)st.dense_features, (double)st.queries, (double)st.matches, (double)st.tracks,
                              (double)st.points, st.index_seconds, st.rays_seconds, st.device_seconds, st.tracks_seconds};
        for (int i = 0; i < 10; i++)
            stats10[i] = v[i];
    }
    if (match_pairs)
        for (size_t i = 0; i < matches.size() && i < match_cap; i++)
            match_pairs[2 * i] = matches[i].first, match_pairs[2 * i + 1] = matches[i].second;
    if (!ok)
    {
        g->error = why;
        return -1;
    }
    return 0;
}

uint32_t och_hilbert_xy2d(int order, int x, int y)
{
    return hilbert_xy2d(order, x, y);
}

} // extern "C"

// ---- mesh refinement (refine_mesh.hpp)
#include "refine_mesh.hpp"

extern "C"
{

size_t och_refine_by_point_density(och_surface *s, size_t max_points_per_triangle, double min_distance_variance, int max_iterations,
                                   double min_triangle_size)
{
    return refineByPointDensity(s->s.mesh, s->s.cloud, max_points_per_triangle, min_distance_variance, max_iterations, min_triangle_size);
}

size_t och_refine_at_point(och_surface *s, double x, double y, int levels)
{
    return refineAtPoint(s->s.mesh, x, y, levels);
}

/* per triangle that holds points (cap rows): tri3 = its vertices, stats2 = {count, distance variance}; returns how many */
size_t och_count_points_per_triangle(const och_surface *s, uint64_t *tri3, double *stats2, size_t cap)
{
    const auto stats = countPointsPerTriangle(s->s.mesh, s->s.cloud);
    TriangleLocator loc(s->s.mesh);
    for (size_t i = 0; i < stats.size() && i < cap; i++)
    {
        size_t v[3] = {0, 0, 0};
        loc.vertices(stats[i].first, v);
        for (int k = 0; k < 3; k++)
            tri3[3 * i + k] = v[k];
        stats2[2 * i] = (double)stats[i].second.count;
        stats2[2 * i + 1] = stats[i].second.distanceVariance;
    }
    return stats.size();
}

/* the triangle under (x, y): its three vertices (UINT64_MAX x 3 when outside) */
void och_surface_locate(const och_surface *s, const double *xy, size_t n, uint64_t *tri3)
{
    TriangleLocator loc(s->s.mesh);
    for (size_t i = 0; i < n; i++)
    {
        size_t v[3];
        const TriangleId t = loc.find(xy[2 * i], xy[2 * i + 1]);
        const bool ok = t.edgeId != MeshEdge::NONE && loc.vertices(t, v);
        for (int k = 0; k < 3; k++)
            tri3[3 * i + k] = ok ? v[k] : UINT64_MAX;
    }
}

/* Pipeline::Impl::mesh_refinement (src/pipeline/pipeline.cpp:666-819) repeated until it leaves the state or max_steps runs
 * were made.  surface: in = the pipeline's surface (may be empty), out = the first surface afterwards.  log8 (max_steps
 * rows): grid level, grid fraction, gsd, triangles above threshold, max points per triangle, triangles created, mesh
 * vertices after the step, 1 if the step asked to repeat.  Returns the number of steps, or -1 + och_last_error(g). */
int och_mesh_refinement_run(och_graph *g, ochip_ctx *ctx, och_surface *surface, int max_steps, double *log8)
{
    std::vector<surface_model> surfaces;
    if (surface->s.mesh.size_nodes() > 0 || !surface->s.cloud.empty())
        surfaces.push_back(surface->s);
    RelaxStage stage;
    MeshRefinementState state;
    int steps = 0;
    while (steps < max_steps)
    {
        Transition t;
        std::string why;
        const int level = state.grid_level;
        if (!mesh_refinement_step(ctx, g->graph, surfaces, stage, state, &t, &why))
        {
            g->error = why;
            return -1;
        }
        if (log8)
        {
            double *row = log8 + 8 * steps;
            row[0] = level, row[1] = state.grid_fraction, row[2] = state.gsd, row[3] = (double)state.triangles_above_threshold;
            row[4] = (double)state.max_points, row[5] = (double)state.refined;
            row[6] = surfaces.empty() ? 0.0 : (double)surfaces[0].mesh.size_nodes();
            row[7] = t == Transition::REPEAT ? 1.0 : 0.0;
        }
        steps++;
        if (t == Transition::NEXT)
            break;
    }
    surface->s = surfaces.empty() ? surface_model() : surfaces[0];
    return steps;
}

} // extern "C"
