// Flat C driver API over the C++ host side (liboc_host.so) so that tests/ and bench.py (Python,
// ctypes) can drive the same code a C++ application would link.  Declared in include/oc_host.h.
#include "../../../include/oc_host.h"

#include "match_features.hpp"

#include <cstring>

using namespace opencalibration_amd;

static std::vector<feature_2d> features_from(const double *loc, const float *strength, const uint64_t *desc, size_t n)
{
    std::vector<feature_2d> f(n);
    for (size_t i = 0; i < n; i++)
    {
        if (loc)
        {
            f[i].location[0] = loc[2 * i];
            f[i].location[1] = loc[2 * i + 1];
        }
        if (strength)
            f[i].strength = strength[i];
        if (desc)
            std::memcpy(f[i].descriptor, desc + 8 * i, 64);
    }
    return f;
}

extern "C"
{

size_t och_subsample(const double *loc, const float *strength, size_t n, double spacing, size_t count, uint64_t *out)
{
    auto f = features_from(loc, strength, nullptr, n);
    auto idx = spatially_subsample_feature_indices(f, spacing, count);
    for (size_t i = 0; i < idx.size(); i++)
        out[i] = idx[i];
    return idx.size();
}

size_t och_matches_from_device(const ochip_match *raw, const uint64_t *idx1, size_t n1, const uint64_t *idx2,
                               size_t n2, uint64_t *out_i1, uint64_t *out_i2, double *out_dist)
{
    std::vector<size_t> i1(idx1, idx1 + n1), i2(idx2, idx2 + n2);
    auto m = matches_from_device(raw, i1, i2);
    for (size_t i = 0; i < m.size(); i++)
    {
        out_i1[i] = m[i].feature_index_1;
        out_i2[i] = m[i].feature_index_2;
        out_dist[i] = m[i].distance;
    }
    return m.size();
}

} // extern "C"
