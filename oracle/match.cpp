// ORACLE — test infrastructure only (see oracle.hpp).
// Restates src/match/match_features.cpp:8-103.
#include "oracle.hpp"

#include <algorithm>
#include <unordered_map>

namespace oracle
{

// match_features.cpp:8-52.  The reference asks a jk::tree::KDTree for the single nearest accepted
// feature and accepts when nn[0].distance (SquaredL2, KDTree.h:681-691: dx*dx + dy*dy of
// query - stored) is > spacing^2.  Only the exact minimum matters, so any exact search gives the same
// answer; here: a hash grid with cell = spacing, 3x3 neighbourhood.
std::vector<size_t> spatially_subsample_feature_indices(const std::vector<feature_2d> &features, double spacing_pixels,
                                                        size_t count)
{
    if (count == 0)
        count = features.size();
    if (count == 0)
        return {};

    std::vector<size_t> sorted_indices(count);
    for (size_t i = 0; i < count; i++)
        sorted_indices[i] = i;
    std::sort(sorted_indices.begin(), sorted_indices.end(),
              [&features](size_t a, size_t b) { return features[a].strength > features[b].strength; });

    std::vector<size_t> indices;
    indices.reserve(features.size() / 4);

    struct cell_key
    {
        int64_t cx, cy;
        bool operator==(const cell_key &o) const
        {
            return cx == o.cx && cy == o.cy;
        }
    };
    struct cell_hash
    {
        size_t operator()(const cell_key &k) const
        {
            return std::hash<int64_t>()(k.cx * 1000003 + k.cy);
        }
    };
    std::unordered_map<cell_key, std::vector<size_t>, cell_hash> grid;
    const double s2 = spacing_pixels * spacing_pixels;
    const double cell = spacing_pixels > 0 ? spacing_pixels : 1.0;

    for (size_t idx : sorted_indices)
    {
        const auto &f = features[idx];
        const int64_t cx = (int64_t)std::floor(f.location[0] / cell), cy = (int64_t)std::floor(f.location[1] / cell);
        bool accept = true;
        if (!indices.empty())
        {
            double best = std::numeric_limits<double>::infinity();
            for (int64_t dy = -1; dy <= 1; dy++)
                for (int64_t dx = -1; dx <= 1; dx++)
                {
                    auto it = grid.find(cell_key{cx + dx, cy + dy});
                    if (it == grid.end())
                        continue;
                    for (size_t other : it->second)
                    {
                        const double ex = f.location[0] - features[other].location[0];
                        const double ey = f.location[1] - features[other].location[1];
                        double d = 0;
                        d += ex * ex;
                        d += ey * ey;
                        if (d < best)
                            best = d;
                    }
                }
            // anything outside the 3x3 block is farther than `spacing`, so best is exact whenever it
            // could fail the test below
            accept = best > s2;
        }
        if (accept)
        {
            grid[cell_key{cx, cy}].push_back(idx);
            indices.push_back(idx);
        }
    }
    return indices;
}

// match_features.cpp:54-103: nearest and second-nearest reference by Hamming distance for every query of the first
// subset (strict '<': the lowest reference position wins a tie, and a tie with the nearest makes the runner-up equal to
// it), Lowe's 0.8 ratio on count / 486 in f64, survivors ordered by distance, largest first, with std::sort.
std::vector<feature_match> match_features_subset(const std::vector<feature_2d> &set_1,
                                                 const std::vector<feature_2d> &set_2,
                                                 const std::vector<size_t> &indices_1,
                                                 const std::vector<size_t> &indices_2)
{
    typedef std::bitset<feature_2d::DESCRIPTOR_BITS> bits_t;
    const double unit = 1.0 / feature_2d::DESCRIPTOR_BITS, inf = std::numeric_limits<double>::infinity();
    const size_t n_ref = indices_2.size();
    std::vector<bits_t> refs; // the reference descriptors, gathered once in subset order
    refs.reserve(n_ref);
    for (size_t j : indices_2)
        refs.push_back(set_2[j].descriptor);

    std::vector<feature_match> out;
    out.reserve(indices_1.size());
    for (size_t q : indices_1)
    {
        const bits_t &query = set_1[q].descriptor;
        double nearest = inf, runner_up = inf;
        size_t nearest_ref = 0;
        for (size_t k = 0; k < n_ref; ++k)
        {
            const double d = (query ^ refs[k]).count() * unit;
            if (!(d < runner_up))
                continue;
            if (d < nearest)
            {
                runner_up = nearest;
                nearest = d;
                nearest_ref = indices_2[k];
            }
            else
                runner_up = d;
        }
        if (nearest < 0.8 * runner_up)
            out.push_back(feature_match{q, nearest_ref, nearest});
    }
    std::sort(out.begin(), out.end(), [](const feature_match &a, const feature_match &b) -> bool { return a.distance > b.distance; });
    return out;
}

} // namespace oracle
