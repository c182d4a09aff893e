// ORACLE — test infrastructure only (see oracle.hpp).
// Restates the per-pair closure of src/pipeline/link_stage.cpp:75-112.
#include "oracle.hpp"

#include <algorithm>

namespace oracle
{

camera_relations link_pair(const std::vector<feature_2d> &f1, const std::vector<feature_2d> &f2,
                           const std::vector<size_t> &idx1, const std::vector<size_t> &idx2, const camera_model &m1,
                           const camera_model &m2)
{
    camera_relations relations;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            relations.ransac_relation.m[i][j] = NAN;

    std::vector<feature_match> coarse_matches = match_features_subset(f1, f2, idx1, idx2); // :83-84
    std::vector<correspondence> coarse_correspondences = distort_keypoints(f1, f2, coarse_matches, m1, m2); // :87-88

    homography_model h; // :91-93
    std::vector<bool> coarse_inliers;
    ransac_trace trace;
    relations.ransac_score = ransac(coarse_correspondences, h, coarse_inliers, &trace);
    relations.ransac_iterations = trace.iterations;
    relations.ransac_improvements = trace.improvements;

    relations.ransac_relation = h.homography; // :95

    const bool can_decompose = h.decompose(coarse_correspondences, coarse_inliers, relations.relative_poses); // :98
    const size_t num_coarse_inliers = std::count(coarse_inliers.begin(), coarse_inliers.end(), true);

    relations.num_coarse_matches = coarse_matches.size();
    relations.coarse_inliers = coarse_inliers;
    relations.can_decompose = can_decompose;

    if (can_decompose && num_coarse_inliers > homography_model::MINIMUM_POINTS * 1.5) // :104
    {
        relations.matches = coarse_matches;
        assembleInliers(relations.matches, coarse_inliers, f1, f2, relations.inlier_matches);
    }
    return relations;
}

} // namespace oracle
