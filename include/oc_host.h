/* oc_host.h — C driver API of liboc_host.so, the C++17 host side of the MI355X hot path.
 *
 * liboc_host.so holds the host code that sits above libochip.so (include/ochip.h): the reference's
 * stage classes and value types re-stated for flat-array device calls (opencalibration_amd/csrc/host).
 * A C++ application links the classes directly; this flat API exists so the same code can be driven
 * from Python (tests/, bench.py) without a C++ test harness.
 */
#ifndef OC_HOST_H
#define OC_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "ochip.h"

#ifdef __cplusplus
extern "C"
{
#endif

    /* spatially_subsample_feature_indices (src/match/match_features.cpp:8-52); out sized >= n */
    size_t och_subsample(const double *loc, const float *strength, size_t n, double spacing, size_t count,
                         uint64_t *out);

    /* ratio test + remap + std::sort of match_features_subset (src/match/match_features.cpp:94-101)
     * over the device kernel's raw output for one pair; outputs sized >= n1 */
    size_t och_matches_from_device(const ochip_match *raw, const uint64_t *idx1, size_t n1, const uint64_t *idx2,
                                   size_t n2, uint64_t *out_i1, uint64_t *out_i2, double *out_dist);

#ifdef __cplusplus
}
#endif
#endif
