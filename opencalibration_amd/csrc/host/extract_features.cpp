#include "extract_features.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <tuple>

namespace opencalibration_amd
{

namespace
{
// nearest kept feature by exact search over a bucket grid (cell = NMS radius in full-resolution pixels);
// only the exact minimum squared distance enters the decision, as with the reference's KD-tree
struct nn_grid
{
    double cell, minx, miny;
    size_t gw, gh;
    std::vector<int32_t> head, next;
    std::vector<std::pair<double, double>> pts;
    nn_grid(double cell_, double minx_, double miny_, double maxx, double maxy) : cell(cell_), minx(minx_), miny(miny_)
    {
        gw = (size_t)std::floor((maxx - minx) / cell) + 1;
        gh = (size_t)std::floor((maxy - miny) / cell) + 1;
        head.assign(gw * gh, -1);
    }
    void add(double x, double y)
    {
        const size_t cx = (size_t)((x - minx) / cell), cy = (size_t)((y - miny) / cell);
        next.push_back(head[cy * gw + cx]);
        head[cy * gw + cx] = (int32_t)pts.size();
        pts.emplace_back(x, y);
    }
    // true if some kept point lies within sqrt(limit2) (i.e. squared distance <= limit2)
    bool any_within(double x, double y, double limit2) const
    {
        const long cx = (long)((x - minx) / cell), cy = (long)((y - miny) / cell);
        for (long yy = std::max(cy - 1, 0L); yy <= std::min(cy + 1, (long)gh - 1); yy++)
            for (long xx = std::max(cx - 1, 0L); xx <= std::min(cx + 1, (long)gw - 1); xx++)
                for (int32_t e = head[(size_t)yy * gw + xx]; e >= 0; e = next[e])
                {
                    const double dx = x - pts[e].first, dy = y - pts[e].second;
                    double d = 0;
                    d += dx * dx;
                    d += dy * dy;
                    if (!(d > limit2))
                        return true;
                }
        return false;
    }
};
} // namespace

std::vector<extracted_features> extract_features_batch(ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images,
                                                       int width, int height, uint32_t max_keypoints, std::string *error,
                                                       bool images_on_device)
{
    std::vector<extracted_features> out(n_images);
    if (n_images == 0 || width <= 0 || height <= 0) // image.empty(): {results, 0}, extract_features.cpp:20-23
        return out;
    const int max_length_pixels = 1600;
    const double nms_pixel_radius = 8;
    std::vector<float> kp((size_t)n_images * max_keypoints * 6);
    std::vector<uint64_t> desc((size_t)n_images * max_keypoints * 8);
    std::vector<uint32_t> counts(n_images);
    int wh[2];
    const int rc = images_on_device ? ochip_akaze_batch_dev(ctx, images_bgr, n_images, width, height, max_keypoints,
                                                            kp.data(), desc.data(), counts.data(), wh)
                                    : ochip_akaze_batch(ctx, images_bgr, n_images, width, height, max_keypoints, kp.data(),
                                                        desc.data(), counts.data(), wh);
    if (rc != OCHIP_OK)
    {
        if (error)
            *error = std::string("ochip_akaze_batch: ") + ochip_last_error(ctx);
        return {};
    }
    const double scale = std::min(1.f, float(max_length_pixels) / (float)std::max(width, height));
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t b = 0; b < n_images; b++)
    {
        const uint32_t n = counts[b];
        const float *k6 = kp.data() + (size_t)b * max_keypoints * 6;
        const uint64_t *dd = desc.data() + (size_t)b * max_keypoints * 8;
        // device order is arbitrary: restore detection order (level, y, x) before the (unstable) strength sort
        std::vector<uint32_t> order(n);
        for (uint32_t i = 0; i < n; i++)
            order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t c) {
            return std::make_tuple(k6[6 * a + 5], k6[6 * a + 1], k6[6 * a]) < std::make_tuple(k6[6 * c + 5], k6[6 * c + 1], k6[6 * c]);
        });
        std::vector<feature_2d> oc_keypoints(n);
        for (uint32_t i = 0; i < n; i++)
        {
            const uint32_t s = order[i];
            feature_2d &p = oc_keypoints[i];
            p.location[0] = k6[6 * s] / scale; // keypoints[i].pt.x / scale, extract_features.cpp:44-45
            p.location[1] = k6[6 * s + 1] / scale;
            p.strength = k6[6 * s + 4];
            std::memcpy(p.descriptor, dd + 8 * s, 64);
        }
        std::sort(oc_keypoints.begin(), oc_keypoints.end(),
                  [](const feature_2d &a, const feature_2d &c) -> bool { return a.strength > c.strength; });
        // non-maximal suppression, extract_features.cpp:58-83 (the seeded first keypoint is visited again by the
        // loop and therefore also heads the dense list)
        std::vector<feature_2d> results, dense;
        if (!oc_keypoints.empty())
        {
            double minx = std::numeric_limits<double>::infinity(), miny = minx, maxx = -minx, maxy = -minx;
            for (const auto &f : oc_keypoints)
            {
                minx = std::min(minx, f.location[0]);
                maxx = std::max(maxx, f.location[0]);
                miny = std::min(miny, f.location[1]);
                maxy = std::max(maxy, f.location[1]);
            }
            nn_grid grid(nms_pixel_radius / scale, minx, miny, maxx, maxy);
            const double limit2 = (nms_pixel_radius * nms_pixel_radius) / (scale * scale); // d2 * scale^2 > r^2
            grid.add(oc_keypoints[0].location[0], oc_keypoints[0].location[1]);
            results.push_back(oc_keypoints[0]);
            for (const feature_2d &f : oc_keypoints)
            {
                // nn[0].distance * sqr(scale) > sqr(nms_pixel_radius)
                bool close = false;
                {
                    // exact form of the reference comparison on the nearest neighbour
                    const long cx = (long)((f.location[0] - grid.minx) / grid.cell), cy = (long)((f.location[1] - grid.miny) / grid.cell);
                    double best = std::numeric_limits<double>::infinity();
                    for (long yy = std::max(cy - 1, 0L); yy <= std::min(cy + 1, (long)grid.gh - 1); yy++)
                        for (long xx = std::max(cx - 1, 0L); xx <= std::min(cx + 1, (long)grid.gw - 1); xx++)
                            for (int32_t e = grid.head[(size_t)yy * grid.gw + xx]; e >= 0; e = grid.next[e])
                            {
                                const double dx = f.location[0] - grid.pts[e].first, dy = f.location[1] - grid.pts[e].second;
                                double d = 0;
                                d += dx * dx;
                                d += dy * dy;
                                best = std::min(best, d);
                            }
                    close = !(best * (scale * scale) > nms_pixel_radius * nms_pixel_radius);
                    (void)limit2;
                }
                if (!close)
                {
                    grid.add(f.location[0], f.location[1]);
                    results.push_back(f);
                }
                else
                    dense.push_back(f);
            }
        }
        out[b].num_sparse_features = results.size();
        results.insert(results.end(), dense.begin(), dense.end());
        out[b].features = std::move(results);
    }
    return out;
}

} // namespace opencalibration_amd
