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
