#include "ransac.hpp"
#include "../decompose.hpp"

#include "../undistort.hpp"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <sstream>

namespace opencalibration_amd
{

const eval_order_entry &EvalOrderCache::get(size_t M)
{
    std::lock_guard<std::mutex> lock(_mutex);
    auto it = _cache.find(M);
    if (it != _cache.end())
        return *it->second;
    // ransac.cpp:92-98,158 — same engine, same seed, same libstdc++ std::shuffle
    std::vector<size_t> eval_order(M);
    std::iota(eval_order.begin(), eval_order.end(), 0);
    std::default_random_engine generator(42);
    std::shuffle(eval_order.begin(), eval_order.end(), generator);
    auto e = std::make_unique<eval_order_entry>();
    e->order.assign(eval_order.begin(), eval_order.end());
    std::ostringstream os;
    os << generator; // minstd_rand0 streams its single state word
    e->rng_state = (uint32_t)std::stoul(os.str());
    return *_cache.emplace(M, std::move(e)).first->second;
}

std::vector<uint32_t> prosac_sorted_idx(const std::vector<feature_match> &matches)
{
    bool has_quality = false;
    for (const auto &m : matches)
        if (m.distance != 0)
        {
            has_quality = true;
            break;
        }
    std::vector<uint32_t> out;
    if (!has_quality)
        return out;
    std::vector<size_t> sorted_idx(matches.size());
    std::iota(sorted_idx.begin(), sorted_idx.end(), 0);
    std::sort(sorted_idx.begin(), sorted_idx.end(),
              [&matches](size_t a, size_t b) { return matches[a].distance < matches[b].distance; });
    out.assign(sorted_idx.begin(), sorted_idx.end());
    return out;
}

void image_to_3d(const double keypoint[2], const CameraModel &model, double ray[3])
{
    const double model8[8] = {model.focal_length_pixels,      model.principle_point[0],      model.principle_point[1],
                              model.radial_distortion[0],     model.radial_distortion[1],    model.radial_distortion[2],
                              model.tangential_distortion[0], model.tangential_distortion[1]};
    ochip_ud::image_to_3d(keypoint, model8, ray);
}

void assembleInliers(const std::vector<feature_match> &matches, const std::vector<bool> &inliers,
                     const std::vector<feature_2d> &source_features, const std::vector<feature_2d> &dest_features,
                     std::vector<feature_match_denormalized> &inlier_list)
{
    inlier_list.reserve(std::count(inliers.begin(), inliers.end(), true));
    for (size_t i = 0; i < matches.size(); i++)
    {
        if (i + 8 < matches.size()) // the feature records are 88 bytes apart in strength order: random lines of two big arrays
        {
            __builtin_prefetch(&source_features[matches[i + 8].feature_index_1]);
            __builtin_prefetch(&dest_features[matches[i + 8].feature_index_2]);
        }
        if (!inliers[i])
            continue;
        feature_match_denormalized fmd;
        const feature_2d &s = source_features[matches[i].feature_index_1], &d = dest_features[matches[i].feature_index_2];
        fmd.pixel_1[0] = s.location[0];
        fmd.pixel_1[1] = s.location[1];
        fmd.pixel_2[0] = d.location[0];
        fmd.pixel_2[1] = d.location[1];
        fmd.feature_index_1 = matches[i].feature_index_1;
        fmd.feature_index_2 = matches[i].feature_index_2;
        fmd.match_index = i;
        inlier_list.push_back(fmd);
    }
}

// cv::decomposeHomographyMat(H, K = I) and the ordering of the poses: csrc/decompose.hpp, shared with the device

bool homography_model::decompose(const std::vector<correspondence> &corrs, const std::vector<bool> &inliers,
                                 std::array<decomposed_pose, 4> &poses) const
{
    std::vector<double> packed;
    packed.reserve(corrs.size() * 6);
    for (size_t j = 0; j < corrs.size(); j++)
        if (inliers[j])
        {
            packed.insert(packed.end(), corrs[j].measurement1, corrs[j].measurement1 + 3);
            packed.insert(packed.end(), corrs[j].measurement2, corrs[j].measurement2 + 3);
        }
    return decompose_inlier_rays(packed.data(), packed.size() / 6, poses);
}

bool homography_model::decompose_inlier_rays(const double *m1m2, size_t n_inliers, std::array<decomposed_pose, 4> &poses) const
{
    return decompose_with(n_inliers, [m1m2](size_t j, const double *&m1, const double *&m2) {
        m1 = m1m2 + 6 * j;
        m2 = m1 + 3;
    }, poses);
}

homography_model::vote_plan homography_model::plan_votes() const
{
    ochip_dc::vote_plan dp;
    ochip_dc::plan_votes(homography, &dp);
    vote_plan plan;
    plan.solutions = (size_t)dp.solutions;
    std::memcpy(plan.N, dp.N, sizeof plan.N);
    std::memcpy(plan.RN, dp.RN, sizeof plan.RN);
    std::memcpy(plan.q, dp.q, sizeof plan.q);
    std::memcpy(plan.t, dp.t, sizeof plan.t);
    return plan;
}

bool homography_model::finish_votes(const vote_plan &plan, const int votes[4], std::array<decomposed_pose, 4> &poses)
{
    for (size_t i = 0; i < plan.solutions; i++)
    {
        poses[i].score = votes[i];
        for (int c = 0; c < 4; c++)
            poses[i].orientation[c] = plan.q[i][c];
        for (int c = 0; c < 3; c++)
            poses[i].position[c] = plan.t[i][c];
    }
    for (size_t i = plan.solutions; i < poses.size(); i++)
        poses[i].score = -1;
    std::stable_sort(poses.begin(), poses.end(),
                     [](const decomposed_pose &p1, const decomposed_pose &p2) { return p1.score >= p2.score; });
    return poses[0].score > 0;
}

} // namespace opencalibration_amd
