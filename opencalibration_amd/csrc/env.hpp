// The two environment switches every part of the library shares (README.md, "Knobs"); plain C++, read on every call.
//   OCHIP_VERBOSE    = 1 | all | comma list of relax, link, extract, dense      progress lines on stderr
//   OCHIP_TEST_HOOKS = comma list of the alternative routes the tests compare the default ones with:
//                      host_sort, host_subset, host_tail, host_nms (the round-2 host code of the link / extract tails),
//                      popcount_match (no matrix-core matcher), chol_verify (factor every system both ways and compare),
//                      back_solve_x_global (step vector in HBM even when it fits LDS), no_dissect (one band, no regions),
//                      jacobian_fp32 (profiles/r04_jacobian_precision_sweep_c5.json),
//                      tile_levels, tile_det / strip_levels, strip_det (the LDS-tile kernels of the scale space / the determinant
//                      for every level / round 5's register strips for every level they can take, instead of the choice by
//                      level size), sort_per_level (a launch per introsort level instead of a workgroup per segment),
//                      host_triangulation (densifyMesh's track points by the host loop instead of ochip_dense_triangulate),
//                      dense_predict_unstaged (the nearest-camera scan and the tracks' member -> image search read the device's records instead of
//                      their LDS copies, as they do above 2 048 cameras),
//                      host_bootstrap (runGroundPlane's cameras without an orientation by round 5's host loop - a problem and two
//                      solves each - instead of the resident launch of csrc/relax_chain.hip), chain_stepped (that launch one phase
//                      at a time: the same code, a launch per phase), chain_partial (it takes the first half of the cameras and
//                      hands the rest to the host loop, as it does when it gives up),
//                      sup_generic, sup_small_lists (AKAZE's suppression: the mask probes written for any radius instead of the
//                      branch-free ones for the default scale space's radii; 64 instead of 2 048 list entries in LDS)
// Other switches (read where they apply): OCHIP_CHAIN_WORKGROUPS (grid of the resident launch; default half the compute units),
// OCHIP_IP_PRIORITY=0 (och_initial_processing_step: link and relax streams at the default priority), OCHIP_STRIP_MIN_PIXELS
// (pixels per launch from which a level takes the register-strip kernels), OCHIP_EXTRACT_GATE=0 (two surveys may extract at once),
// OCHIP_REQUIRE_REF=1 (tests: the pins against the reference's own headers must run)
#pragma once

#include <cstdlib>
#include <cstring>

inline bool ochip_env_list_has(const char *var, const char *name, bool one_means_all)
{
    const char *e = std::getenv(var);
    if (!e || !*e)
        return false;
    if (one_means_all && (std::strcmp(e, "1") == 0 || std::strcmp(e, "all") == 0))
        return true;
    const size_t len = std::strlen(name);
    for (const char *p = e; *p;)
    {
        const char *q = std::strchr(p, ',');
        const size_t n = q ? (size_t)(q - p) : std::strlen(p);
        if (n == len && std::strncmp(p, name, len) == 0)
            return true;
        p += n + (q ? 1 : 0);
    }
    return false;
}
inline bool ochip_verbose(const char *what)
{
    return ochip_env_list_has("OCHIP_VERBOSE", what, true);
}
inline bool ochip_test_hook(const char *name)
{
    return ochip_env_list_has("OCHIP_TEST_HOOKS", name, false);
}
