// Pieces of the overlapped load + link stage (host/load_link.cpp) shared with the one-survey-over-several-ranks driver
// (host/shard_link.cpp).
#pragma once

#include "capi_graph.hpp"

namespace opencalibration_amd
{

struct owned_pair // a directed pair and the link (= image, in batch order) whose batch runs it
{
    size_t owner;
    LinkStage::link_pair pair;
};
std::vector<owned_pair> pair_owners(const std::vector<NodeLinks> &links);

// One node per image of a survey, in image order (load_stage.cpp:89-103 without the features, which arrive later).
std::vector<size_t> add_survey_nodes(och_graph *g, uint32_t n_images, uint32_t model, const double *positions,
                                     const double *orientations, uint64_t *node_ids_out);

// Extracts the images [first, first + count) of `ids` (images_bgr points at image `first`) chunk by chunk and runs
// `pairs` on link runners (their own device contexts) as soon as every image of this block a batch of them touches has
// its features; images outside the block must have been prepared before the call.  link.init() and
// link.prepare_index() have been called; finalize() is left to the caller.  False + g->error on failure.
bool load_link_stream(och_graph *g, ochip_ctx *ctx, LinkStage &link, const std::vector<size_t> &ids, uint32_t first,
                      uint32_t count, const uint8_t *images_bgr, int width, int height, uint32_t max_keypoints,
                      bool images_on_device, const std::vector<owned_pair> &pairs, double *total_out, double *sparse_out,
                      double *t_extract_done);

} // namespace opencalibration_amd
