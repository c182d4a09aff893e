// Host half of the match step (same entry points as include/opencalibration/match/match_features.hpp).
#pragma once

#include "types.hpp"

#include "../../../include/ochip.h"

namespace opencalibration_amd
{

// match_features.hpp:10 — greedy strength-ordered Poisson-disk subsample (src/match/match_features.cpp:8-52)
std::vector<size_t> spatially_subsample_feature_indices(const std::vector<feature_2d> &features, double spacing_pixels,
                                                        size_t count = 0);

// Second half of match_features_subset (src/match/match_features.cpp:94-101) applied to the device
// kernel's per-query (best_k, best_count, second_count): Lowe ratio test in f64, remap through
// indices_2, libstdc++ std::sort by distance descending (tie order preserved because the input
// order and the sort are the reference's).
std::vector<feature_match> matches_from_device(const ochip_match *raw, const std::vector<size_t> &indices_1,
                                               const std::vector<size_t> &indices_2);

} // namespace opencalibration_amd
