// libochip.so — the tail of match_features_subset on the device (gfx950): Lowe's ratio test and the std::sort of the matches
// by descending distance (src/match/match_features.cpp:94-101), on the 2-NN records ochip_match_launch left in HBM.
//
// Hamming counts tie all the time, and the order std::sort leaves equal distances in is part of the reference's result: it
// is the correspondences' order, which PROSAC's order, the evaluation order and the inlier lists are built on.  The sort
// is std_sort.hip (libstdc++'s introsort restated as parallel partitions); the records it sorts are (count << 32 | query),
// written per pair in query order - the order the reference pushes its matches in.  The sorted records stay in HBM for
// ochip_ransac_homography_batch_sorted; the host only learns how many matches each pair has.
#include "ctx.hpp"

#include <algorithm>
#include <vector>

using namespace ochip;

namespace
{

// one wavefront per pair: queries that pass `best < 0.8 * second` (distances = count / 486 as doubles, as the reference
// computes them), compacted in query order
__global__ __launch_bounds__(256) void ratio_compact_kernel(const ochip_match *__restrict__ raw, const ochip_pair *__restrict__ pairs,
                                                            const uint32_t *__restrict__ img_n, unsigned int n_pairs,
                                                            const unsigned int *__restrict__ seg_begin, unsigned int *__restrict__ seg_end,
                                                            unsigned long long *__restrict__ recs)
{
    const unsigned int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_pairs)
        return;
    const int lane = threadIdx.x & 63;
    const unsigned int n1 = img_n[pairs[p].image_1], n2 = img_n[pairs[p].image_2];
    const unsigned int off = seg_begin[p];
    unsigned int kept = 0;
    if (n2 > 0)
        for (unsigned int a0 = 0; a0 < n1; a0 += 64)
        {
            const unsigned int a = a0 + lane;
            bool pass = false;
            ochip_match m{};
            if (a < n1)
            {
                m = raw[off + a];
                const double unit = 1.0 / 486; // 1.0 / feature_2d::DESCRIPTOR_BITS
                const double best = (double)m.best_count * unit;
                const double second = m.second_count == OCHIP_NO_SECOND ? __longlong_as_double(0x7FF0000000000000ll) : (double)m.second_count * unit;
                pass = best < 0.8 * second;
            }
            const unsigned long long mask = __ballot(pass);
            if (pass)
                recs[off + kept + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull))] = ((unsigned long long)m.best_count << 32) | a;
            kept += (unsigned int)__popcll(mask);
        }
    if (lane == 0)
        seg_end[p] = off + kept;
}

} // namespace

extern "C" int ochip_match_sort(ochip_ctx *ctx, const ochip_pair *pairs, uint32_t n_pairs, const uint64_t *out_offset, uint64_t out_total,
                                uint32_t *counts_out, uint8_t *fallback_out)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (n_pairs == 0)
        return OCHIP_OK;
    if (!pairs || !out_offset || !counts_out || !fallback_out)
        return ochip_fail(ctx, OCHIP_EINVAL, "NULL argument");
    if (out_total != ctx->match_out_total || !ctx->match_out_dev)
        return ochip_fail(ctx, OCHIP_ESTATE, "ochip_match_sort must follow ochip_match_launch with the same offsets");
    if (out_total >= 0xFFFFFFFFull)
        return ochip_fail(ctx, OCHIP_EINVAL, "ochip_match_sort: %llu query records in one batch (limit 2^32)", (unsigned long long)out_total);
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc = ochip_ensure(ctx, &ctx->ms_recs_dev, &ctx->ms_recs_cap, std::max<size_t>(out_total, 1) * 8);
    if (rc == OCHIP_OK)
        rc = ochip_ensure(ctx, &ctx->ms_seg_dev, &ctx->ms_seg_cap, (size_t)n_pairs * 16); // begin | end | pairs
    if (rc == OCHIP_OK)
        rc = ochip_ensure(ctx, &ctx->ms_flag_dev, &ctx->ms_flag_cap, (size_t)n_pairs);
    if (rc != OCHIP_OK)
        return rc;
    ctx->ms_pairs = n_pairs;
    unsigned int *seg_begin = (unsigned int *)ctx->ms_seg_dev, *seg_end = seg_begin + n_pairs;
    std::vector<unsigned int> begin(n_pairs);
    uint32_t max_n1 = 0;
    for (uint32_t p = 0; p < n_pairs; p++)
    {
        begin[p] = (unsigned int)out_offset[p];
        max_n1 = std::max(max_n1, ctx->img_n[pairs[p].image_1]);
    }
    ochip_pair *pairs_dev = (ochip_pair *)(seg_end + n_pairs);
    OCHIP_HIP(ctx, hipMemcpyAsync(seg_begin, begin.data(), (size_t)n_pairs * 4, hipMemcpyHostToDevice, st));
    OCHIP_HIP(ctx, hipMemcpyAsync(pairs_dev, pairs, (size_t)n_pairs * sizeof(ochip_pair), hipMemcpyHostToDevice, st));
    // (ochip_match_launch left the images' counts on the device)
    hipLaunchKernelGGL(ratio_compact_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, st, (const ochip_match *)ctx->match_out_dev,
                       (const ochip_pair *)pairs_dev, (const uint32_t *)ctx->img_n_dev, n_pairs, (const unsigned int *)seg_begin, seg_end,
                       (unsigned long long *)ctx->ms_recs_dev);
    std::vector<std::pair<void *, size_t>> allocs;
    rc = std_sort_enqueue(ctx, &allocs, (unsigned long long *)ctx->ms_recs_dev, out_total, seg_begin, seg_end, n_pairs, max_n1,
                          (unsigned char *)ctx->ms_flag_dev);
    std::vector<unsigned int> end(n_pairs);
    if (rc == OCHIP_OK && (hipMemcpyAsync(end.data(), seg_end, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                           hipMemcpyAsync(fallback_out, ctx->ms_flag_dev, n_pairs, hipMemcpyDeviceToHost, st) != hipSuccess))
        rc = ochip_fail(ctx, OCHIP_EHIP, "ochip_match_sort: download failed");
    const hipError_t werr = ochip_stream_wait(ctx, st);
    for (auto &a : allocs)
        ochip_pool_put(ctx, a.first, a.second);
    if (rc == OCHIP_OK && werr != hipSuccess)
        rc = ochip_fail(ctx, OCHIP_EHIP, "ochip_match_sort: %s", hipGetErrorString(werr));
    if (rc == OCHIP_OK)
        for (uint32_t p = 0; p < n_pairs; p++)
            counts_out[p] = end[p] - begin[p];
    return rc;
}
