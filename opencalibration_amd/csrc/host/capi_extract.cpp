// Flat C driver API for the extract step (declared in include/oc_host.h).
#include "../../../include/oc_host.h"

#include "capi_graph.hpp"
#include "extract_features.hpp"

#include <cstring>

using namespace opencalibration_amd;

static std::string g_extract_error;

extern "C"
{

const char *och_extract_last_error(void)
{
    return g_extract_error.c_str();
}

// images: n x height x width x 3 BGR.  Per image up to max_out features are written at stride max_out:
// loc (x,y full-resolution pixels), strength, desc (8 u64); counts[i] = features of image i,
// num_sparse[i] = how many of them passed the 8 px NMS (they come first).  Returns 0 or -1.
int och_extract_features_batch(ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images, int width, int height,
                               uint32_t max_keypoints, uint32_t max_out, double *loc, float *strength, uint64_t *desc,
                               uint32_t *counts, uint32_t *num_sparse, int images_on_device)
{
    auto ex = extract_features_batch(ctx, images_bgr, n_images, width, height, max_keypoints, &g_extract_error,
                                     images_on_device != 0);
    if (ex.size() != n_images)
        return -1;
    for (uint32_t b = 0; b < n_images; b++)
    {
        const size_t n = ex[b].features.size();
        if (n > max_out)
        {
            g_extract_error = "image " + std::to_string(b) + " produced " + std::to_string(n) + " features > max_out";
            return -1;
        }
        counts[b] = (uint32_t)n;
        num_sparse[b] = (uint32_t)ex[b].num_sparse_features;
        for (size_t i = 0; i < n; i++)
        {
            const feature_2d &f = ex[b].features[i];
            loc[((size_t)b * max_out + i) * 2] = f.location[0];
            loc[((size_t)b * max_out + i) * 2 + 1] = f.location[1];
            strength[(size_t)b * max_out + i] = f.strength;
            std::memcpy(desc + ((size_t)b * max_out + i) * 8, f.descriptor, 64);
        }
    }
    return 0;
}

int och_graph_load_images(och_graph *g, ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images, int width, int height,
                          uint32_t max_keypoints, int images_on_device, uint32_t model, const double *positions,
                          uint64_t *node_ids_out, double *totals2)
{
    if (!g || !ctx || (n_images && (!images_bgr || !positions)) || model >= g->models.size())
    {
        if (g)
            g->error = "och_graph_load_images: bad argument";
        return -1;
    }
    auto ex = extract_features_batch(ctx, images_bgr, n_images, width, height, max_keypoints, &g->error, images_on_device != 0);
    if (ex.size() != n_images)
        return -1;
    double total = 0, sparse = 0;
    for (uint32_t b = 0; b < n_images; b++)
    {
        image img;
        total += (double)ex[b].features.size();
        sparse += (double)ex[b].num_sparse_features;
        img.features = std::move(ex[b].features);
        img.num_sparse_features = ex[b].num_sparse_features;
        img.coarse_subset = std::move(ex[b].coarse_subset);
        img.coarse_spacing = ex[b].coarse_spacing;
        img.model = g->models[model];
        for (int i = 0; i < 3; i++)
            img.position[i] = positions[3 * (size_t)b + i];
        img.path = "image_" + std::to_string(g->graph.size_nodes());
        const size_t id = g->graph.addNode(std::move(img));
        if (node_ids_out)
            node_ids_out[b] = id;
    }
    if (totals2)
    {
        totals2[0] = total;
        totals2[1] = sparse;
    }
    return 0;
}

} // extern "C"
