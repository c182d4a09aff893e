#include "match_features.hpp"

#include <algorithm>
#include <limits>

namespace opencalibration_amd
{

// Reference: src/match/match_features.cpp:8-52.  The reference asks a KD-tree for the nearest
// accepted feature; the decision only needs "is any accepted feature within `spacing`", which a
// uniform bucket grid over the image answers exactly.  Squared distances are formed exactly as the
// reference's SquaredL2 functor does (external/jk-tree/include/jk/KDTree.h:681-691) so the boundary
// case d^2 == spacing^2 falls on the same side.
std::vector<size_t> spatially_subsample_feature_indices(const std::vector<feature_2d> &features, double spacing_pixels,
                                                        size_t count)
{
    if (count == 0)
        count = features.size();
    if (count == 0)
        return {};
    count = std::min(count, features.size());

    // strengths and locations side by side in flat arrays: the sort and the bucket walk below would otherwise
    // stride through the 88-byte feature records (same comparator answers, hence the same permutation)
    std::vector<float> strength(count);
    std::vector<double> lx(count), ly(count);
    double min_x = std::numeric_limits<double>::infinity(), min_y = min_x, max_x = -min_x, max_y = -min_x;
    for (size_t i = 0; i < count; i++)
    {
        strength[i] = features[i].strength;
        lx[i] = features[i].location[0];
        ly[i] = features[i].location[1];
        min_x = std::min(min_x, lx[i]);
        max_x = std::max(max_x, lx[i]);
        min_y = std::min(min_y, ly[i]);
        max_y = std::max(max_y, ly[i]);
    }
    std::vector<size_t> sorted_indices(count);
    for (size_t i = 0; i < count; i++)
        sorted_indices[i] = i;
    std::sort(sorted_indices.begin(), sorted_indices.end(),
              [&strength](size_t a, size_t b) { return strength[a] > strength[b]; });

    const double cell = spacing_pixels > 0 ? spacing_pixels : 1.0;
    const double limit = spacing_pixels * spacing_pixels;
    size_t gw = (size_t)std::floor((max_x - min_x) / cell) + 1, gh = (size_t)std::floor((max_y - min_y) / cell) + 1;
    if (!(gw * gh <= 4 * count + 1024)) // degenerate extent: coarsen the buckets (still exact, just slower)
    {
        gw = gh = 1;
    }
    const double inv_w = gw == 1 ? 0.0 : 1.0 / cell, inv_h = gh == 1 ? 0.0 : 1.0 / cell;
    std::vector<int32_t> head(gw * gh, -1), next;
    std::vector<size_t> indices;
    indices.reserve(features.size() / 4);
    next.reserve(features.size() / 4);

    for (size_t idx : sorted_indices)
    {
        const double x = lx[idx], y = ly[idx];
        const long cx = gw == 1 ? 0 : (long)((x - min_x) * inv_w), cy = gh == 1 ? 0 : (long)((y - min_y) * inv_h);
        bool accept = true;
        const long x0 = gw == 1 ? 0 : std::max(cx - 1, 0L), x1 = gw == 1 ? 0 : std::min(cx + 1, (long)gw - 1);
        const long y0 = gh == 1 ? 0 : std::max(cy - 1, 0L), y1 = gh == 1 ? 0 : std::min(cy + 1, (long)gh - 1);
        for (long yy = y0; yy <= y1 && accept; yy++)
            for (long xx = x0; xx <= x1 && accept; xx++)
                for (int32_t e = head[(size_t)yy * gw + xx]; e >= 0; e = next[e])
                {
                    const size_t o = indices[e];
                    const double dx = x - lx[o], dy = y - ly[o];
                    double d = 0;
                    d += dx * dx;
                    d += dy * dy;
                    if (!(d > limit)) // nn[0].distance > spacing^2 must hold for every accepted feature
                    {
                        accept = false;
                        break;
                    }
                }
        if (accept)
        {
            const size_t c = (size_t)std::min(std::max(cy, 0L), (long)gh - 1) * gw +
                             (size_t)std::min(std::max(cx, 0L), (long)gw - 1);
            next.push_back(head[c]);
            head[c] = (int32_t)indices.size();
            indices.push_back(idx);
        }
    }
    return indices;
}

std::vector<feature_match> matches_from_device(const ochip_match *raw, const std::vector<size_t> &indices_1,
                                               const std::vector<size_t> &indices_2)
{
    std::vector<feature_match> results;
    results.reserve(indices_1.size());
    if (indices_2.empty())
        return results; // best stays +inf: `inf < 0.8*inf` is false for every query
    const double inf = std::numeric_limits<double>::infinity();
    for (size_t a = 0; a < indices_1.size(); a++)
    {
        const ochip_match &m = raw[a];
        const double best = (size_t)m.best_count * (1.0 / feature_2d::DESCRIPTOR_BITS);
        const double second =
            m.second_count == OCHIP_NO_SECOND ? inf : (size_t)m.second_count * (1.0 / feature_2d::DESCRIPTOR_BITS);
        if (best < 0.8 * second)
            results.push_back(feature_match{indices_1[a], indices_2[m.best_k], best});
    }
    std::sort(results.begin(), results.end(),
              [](const feature_match &f1, const feature_match &f2) -> bool { return f1.distance > f2.distance; });
    return results;
}

} // namespace opencalibration_amd
