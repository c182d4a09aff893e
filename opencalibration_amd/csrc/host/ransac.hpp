// Host half of the RANSAC step (reference: include/opencalibration/model_inliers/ransac.hpp,
// homography_model.hpp).  The hypothesis loop itself runs on the device
// (ochip_ransac_homography_batch); the host contributes the two libstdc++-defined orders and the
// homography decomposition.
#pragma once

#include "types.hpp"

#include <mutex>

namespace opencalibration_amd
{

// iota(M) after std::shuffle with std::default_random_engine(42) (ransac.cpp:92-98,158) and the engine
// state right after the shuffle.  Depends on M only, so it is cached.
struct eval_order_entry
{
    std::vector<uint32_t> order;
    uint32_t rng_state;
};
class EvalOrderCache
{
  public:
    const eval_order_entry &get(size_t M);

  private:
    std::mutex _mutex;
    std::unordered_map<size_t, std::unique_ptr<eval_order_entry>> _cache;
};

// PROSAC order (ransac.cpp:83-90): iota(M) std::sort-ed by quality ascending; quality[i] is the match
// distance.  Empty if no quality is non-zero (has_quality == false): the device then samples uniformly.
std::vector<uint32_t> prosac_sorted_idx(const std::vector<feature_match> &matches);

struct homography_model // include/opencalibration/model_inliers/homography_model.hpp:16-37 (host-visible part)
{
    static constexpr size_t MINIMUM_POINTS = 4;
    double inlier_threshold = 0.005;
    double homography[9] = {NAN, NAN, NAN, NAN, NAN, NAN, NAN, NAN, NAN}; // row-major

    // homography_model.cpp:138-185: cv::decomposeHomographyMat(H, I) + cheirality vote over the inlier
    // rays + std::stable_sort by score.
    bool decompose(const std::vector<correspondence> &corrs, const std::vector<bool> &inliers,
                   std::array<decomposed_pose, 4> &poses) const;
    // the same on the inlier correspondences only, packed as n x {measurement1(3), measurement2(3)}: the vote only
    // ever looks at inliers, so the batch runner does not materialise a correspondence per match
    bool decompose_inlier_rays(const double *m1m2, size_t n_inliers, std::array<decomposed_pose, 4> &poses) const;
    // the same without packing anything: ray_pair(j, m1, m2) points at the two rays of the j-th inlier.  One pass over
    // the inliers serves all (up to four) solutions; the votes are integers, so the order of evaluation is free.
    struct vote_plan
    {
        size_t solutions = 0;
        double N[4][3], RN[4][3];
        double q[4][4], t[4][3];
    };
    vote_plan plan_votes() const;
    static bool finish_votes(const vote_plan &plan, const int votes[4], std::array<decomposed_pose, 4> &poses);
    template <class RayPair> bool decompose_with(size_t n_inliers, RayPair ray_pair, std::array<decomposed_pose, 4> &poses) const
    {
        const vote_plan plan = plan_votes();
        int votes[4] = {0, 0, 0, 0};
        for (size_t j = 0; j < n_inliers; j++)
        {
            const double *m1, *m2;
            ray_pair(j, m1, m2);
            for (size_t i = 0; i < plan.solutions; i++)
            {
                const double dot1 = plan.N[i][0] * m1[0] + plan.N[i][1] * m1[1] + plan.N[i][2] * m1[2];
                const double dot2 = plan.RN[i][0] * m2[0] + plan.RN[i][1] * m2[1] + plan.RN[i][2] * m2[2];
                votes[i] += (dot1 >= 0 && dot2 >= 0) ? 1 : 0;
            }
        }
        return finish_votes(plan, votes, poses);
    }
};

// ransac.cpp:263-282
void assembleInliers(const std::vector<feature_match> &matches, const std::vector<bool> &inliers,
                     const std::vector<feature_2d> &source_features, const std::vector<feature_2d> &dest_features,
                     std::vector<feature_match_denormalized> &inlier_list);

// distort_keypoints.cpp:68-103 incl. the TinySolver lens-model inversion (csrc/undistort.hpp); host copy: hands the
// cheirality vote the same unit rays the device computed, and the relax its camera-frame rays
void image_to_3d(const double keypoint[2], const CameraModel &model, double ray[3]);

} // namespace opencalibration_amd
