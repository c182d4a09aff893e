// Host half of the extract step (reference: include/opencalibration/extract/extract_features.hpp:10-18).
#pragma once

#include "types.hpp"

#include "../../../include/ochip.h"

#include <functional>
#include <string>

namespace opencalibration_amd
{

struct extracted_features // extract_features.hpp:10-15
{
    std::vector<feature_2d> features;
    size_t num_sparse_features = 0;
    // spatially_subsample_feature_indices(features, coarse_spacing, num_sparse_features) when the device computed it with the
    // list (csrc/features.hip); coarse_spacing == 0: not computed
    std::vector<uint32_t> coarse_subset;
    double coarse_spacing = 0;
};

// extract_features(const cv::Mat&) for a batch of equally sized BGR images (n x height x width x 3 bytes):
// grey + INTER_AREA downscale + AKAZE on the device (ochip_akaze_batch), then the host tail of
// src/extract/extract_features.cpp:38-87: rescale to full-resolution pixels, std::sort by response, greedy
// 8 px NMS, [sparse..., dense...].  On a device error returns an empty vector and sets *error.
// images_on_device: images_bgr is a device pointer (the images are already resident in HBM).
std::vector<extracted_features> extract_features_batch(ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images,
                                                       int width, int height, uint32_t max_keypoints, std::string *error,
                                                       bool images_on_device = false);

// The same, streaming: `on_chunk(first, count, features)` is called (from the calling thread, chunks in completion
// order) as soon as the features of images [first, first + count) are final, while later chunks are still on the
// device; `features` points at the `count` results, which the callback may move from.  host_threads caps the OpenMP
// team of the host tail (0 = the whole team).  Returns false and sets *error on a device error.
bool extract_features_stream(ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images, int width, int height,
                             uint32_t max_keypoints, bool images_on_device, int host_threads,
                             const std::function<void(uint32_t, uint32_t, extracted_features *)> &on_chunk,
                             std::string *error);

} // namespace opencalibration_amd
