// libochip.so — brute-force 486-bit Hamming 2-NN over image pairs (gfx950).
//
// Replaces the double loop of src/match/match_features.cpp:71-93 for every directed image pair of a
// link batch in one launch.
//
// Mapping (DESIGN.md "match kernel"): one lane owns QPT query descriptors (16 dwords each, resident in
// VGPRs for the whole kernel); the reference descriptor index k is wave-uniform, so the 64-byte
// reference rows are fetched by the scalar unit (s_load_dwordx16 -> SGPRs) and enter the VALU as
// scalar operands: per (query, reference) the vector pipe executes exactly 16 v_xor_b32 +
// 16 v_bcnt_u32_b32 (accumulating form) + 3 ops of top-2 bookkeeping, with no LDS traffic and no
// cross-lane step.  HBM traffic is one read of each image's descriptors per workgroup (L2/K$ absorb
// the re-reads), so the kernel is integer-VALU bound, not HBM bound.
//
// Top-2 semantics of the reference (strict '<' in a sequential scan over k): best = lexicographic
// minimum of (count, k); second = minimum count over the other k (equals best on a tie).  With
// key = count << 20 | k both are order independent: best' = min(best, key),
// second' = median(best, second, key).
#include "ctx.hpp"

namespace
{

constexpr int BLOCK = 256;
constexpr uint32_t KEY_SHIFT = 20;
constexpr uint32_t KEY_MASK = (1u << KEY_SHIFT) - 1;

__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int QPT>
__global__ __launch_bounds__(BLOCK) void hamming_2nn_kernel(const uint32_t *__restrict__ desc,
                                                            const uint64_t *__restrict__ img_off,
                                                            const uint32_t *__restrict__ img_n,
                                                            const ochip_pair *__restrict__ pairs,
                                                            const uint64_t *__restrict__ out_off,
                                                            ochip_match *__restrict__ out, uint32_t chunks_per_pair)
{
    const uint32_t pair = blockIdx.x / chunks_per_pair;
    const uint32_t chunk = blockIdx.x - pair * chunks_per_pair;
    const ochip_pair pr = pairs[pair];
    const uint32_t n1 = img_n[pr.image_1], n2 = img_n[pr.image_2];
    const uint32_t q0 = chunk * (BLOCK * QPT);
    if (q0 >= n1)
        return;
    const uint32_t *__restrict__ Q = desc + img_off[pr.image_1] * 16;
    const uint32_t *__restrict__ R = desc + img_off[pr.image_2] * 16;

    uint32_t q[QPT][16];
    uint32_t best[QPT], second[QPT];
#pragma unroll
    for (int j = 0; j < QPT; j++)
    {
        uint32_t qi = q0 + j * BLOCK + threadIdx.x;
        qi = qi < n1 ? qi : n1 - 1; // tail lanes recompute the last query and do not store
        const uint4 *src = reinterpret_cast<const uint4 *>(Q + (size_t)qi * 16);
#pragma unroll
        for (int v = 0; v < 4; v++)
        {
            const uint4 t = src[v];
            q[j][4 * v + 0] = t.x;
            q[j][4 * v + 1] = t.y;
            q[j][4 * v + 2] = t.z;
            q[j][4 * v + 3] = t.w;
        }
        best[j] = 0xFFFFFFFFu;
        second[j] = 0xFFFFFFFFu;
    }

    // Software-pipelined over k with two SGPR row buffers: the scalar load of row k+1 is in flight
    // while the VALU works on row k (wave-uniform addresses => s_load_dwordx16).
    auto load_row = [&](uint32_t (&rw)[16], uint32_t k) {
        const uint32_t kk = k < n2 ? k : n2 - 1;
        const uint32_t *__restrict__ r = R + (size_t)kk * 16;
#pragma unroll
        for (int w = 0; w < 16; w++)
            rw[w] = r[w];
    };
    auto score_row = [&](const uint32_t (&rw)[16], uint32_t k) {
#pragma unroll
        for (int j = 0; j < QPT; j++)
        {
            uint32_t cnt = 0;
#pragma unroll
            for (int w = 0; w < 16; w++)
                cnt += __builtin_popcount(q[j][w] ^ rw[w]);
            const uint32_t key = (cnt << KEY_SHIFT) | k;
            second[j] = med3_u32(best[j], second[j], key);
            best[j] = best[j] < key ? best[j] : key;
        }
    };
    if (n2 > 0)
    {
        uint32_t ra[16], rb[16];
        load_row(ra, 0);
        uint32_t k = 0;
        for (; k + 1 < n2; k += 2)
        {
            load_row(rb, k + 1);
            score_row(ra, k);
            load_row(ra, k + 2);
            score_row(rb, k + 1);
        }
        if (k < n2)
            score_row(ra, k);
    }

    ochip_match *__restrict__ o = out + out_off[pair];
#pragma unroll
    for (int j = 0; j < QPT; j++)
    {
        const uint32_t qi = q0 + j * BLOCK + threadIdx.x;
        if (qi < n1)
        {
            ochip_match m;
            m.best_k = best[j] & KEY_MASK;
            m.best_count = (uint16_t)(best[j] >> KEY_SHIFT);
            m.second_count = second[j] == 0xFFFFFFFFu ? (uint16_t)OCHIP_NO_SECOND : (uint16_t)(second[j] >> KEY_SHIFT);
            o[qi] = m;
        }
    }
}

} // namespace

extern "C"
{

int ochip_match_launch(ochip_ctx *ctx, const ochip_pair *pairs, uint32_t n_pairs, const uint64_t *out_offset,
                       uint64_t out_total)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (n_pairs && (!pairs || !out_offset))
        return ochip_fail(ctx, OCHIP_EINVAL, "pairs/out_offset is NULL");
    if (!ctx->desc_dev)
        return ochip_fail(ctx, OCHIP_ESTATE, "ochip_descriptors_reserve has not been called");
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    ctx->match_out_total = out_total;
    if (n_pairs == 0)
        return OCHIP_OK;

    uint32_t max_n1 = 0;
    for (uint32_t p = 0; p < n_pairs; p++)
    {
        const ochip_pair &pr = pairs[p];
        if (pr.image_1 >= ctx->n_images || pr.image_2 >= ctx->n_images || !ctx->img_set[pr.image_1] ||
            !ctx->img_set[pr.image_2])
            return ochip_fail(ctx, OCHIP_ESTATE, "pair %u references an image that was not uploaded", p);
        if (ctx->img_n[pr.image_2] > KEY_MASK)
            return ochip_fail(ctx, OCHIP_EINVAL, "image %u has %u descriptors; the kernel supports <= %u", pr.image_2,
                              ctx->img_n[pr.image_2], KEY_MASK);
        if (out_offset[p] + ctx->img_n[pr.image_1] > out_total)
            return ochip_fail(ctx, OCHIP_EINVAL, "out_offset[%u] + n1 exceeds out_total", p);
        max_n1 = ctx->img_n[pr.image_1] > max_n1 ? ctx->img_n[pr.image_1] : max_n1;
    }

    if (ctx->img_tables_dirty)
    {
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->img_off_dev, ctx->img_off.data(), (size_t)ctx->n_images * 8,
                                      hipMemcpyHostToDevice, ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->img_n_dev, ctx->img_n.data(), (size_t)ctx->n_images * 4,
                                      hipMemcpyHostToDevice, ctx->stream));
        ctx->img_tables_dirty = false;
    }
    if (n_pairs > ctx->pairs_cap)
    {
        if (ctx->pairs_dev)
            OCHIP_HIP(ctx, hipFree(ctx->pairs_dev));
        if (ctx->out_off_dev)
            OCHIP_HIP(ctx, hipFree(ctx->out_off_dev));
        ctx->pairs_dev = nullptr;
        ctx->out_off_dev = nullptr;
        ctx->pairs_cap = 0;
        if (hipMalloc((void **)&ctx->pairs_dev, (size_t)n_pairs * sizeof(ochip_pair)) != hipSuccess ||
            hipMalloc((void **)&ctx->out_off_dev, (size_t)n_pairs * 8) != hipSuccess)
            return ochip_fail(ctx, OCHIP_ENOMEM, "hipMalloc for the pair table failed");
        ctx->pairs_cap = n_pairs;
    }
    {
        void *p = ctx->match_out_dev;
        size_t cap = ctx->match_out_cap * sizeof(ochip_match);
        int rc = ochip_ensure(ctx, &p, &cap, (size_t)(out_total ? out_total : 1) * sizeof(ochip_match));
        ctx->match_out_dev = (ochip_match *)p;
        ctx->match_out_cap = cap / sizeof(ochip_match);
        if (rc)
            return rc;
    }
    OCHIP_HIP(ctx, hipMemcpyAsync(ctx->pairs_dev, pairs, (size_t)n_pairs * sizeof(ochip_pair), hipMemcpyHostToDevice,
                                  ctx->stream));
    OCHIP_HIP(ctx, hipMemcpyAsync(ctx->out_off_dev, out_offset, (size_t)n_pairs * 8, hipMemcpyHostToDevice,
                                  ctx->stream));
    // the pageable sources above must be consumed before we return to the caller
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    if (max_n1 == 0)
        return OCHIP_OK;

    static const int qpt = []() {
        const char *e = getenv("OCHIP_MATCH_QPT"); // tuning knob; default chosen from measurements
        const int v = e ? atoi(e) : 1;
        return (v == 1 || v == 2 || v == 4) ? v : 1;
    }();
    const uint32_t chunks = (max_n1 + BLOCK * qpt - 1) / (BLOCK * qpt);
    const uint64_t blocks = (uint64_t)chunks * n_pairs;
    if (blocks > 0x7FFFFFFFull)
        return ochip_fail(ctx, OCHIP_EINVAL, "batch too large: %llu workgroups", (unsigned long long)blocks);
    hipEvent_t e0, e1;
    ochip_prof_begin(ctx, OCHIP_K_MATCH, &e0, &e1);
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(BLOCK), 0, ctx->stream, ctx->desc_dev,
                           ctx->img_off_dev, ctx->img_n_dev, ctx->pairs_dev, ctx->out_off_dev, ctx->match_out_dev,
                           chunks);
    };
    if (qpt == 1)
        launch(hamming_2nn_kernel<1>);
    else if (qpt == 4)
        launch(hamming_2nn_kernel<4>);
    else
        launch(hamming_2nn_kernel<2>);
    ochip_prof_end(ctx, OCHIP_K_MATCH, e0, e1);
    OCHIP_HIP(ctx, hipGetLastError());
    return OCHIP_OK;
}

int ochip_match_fetch(ochip_ctx *ctx, ochip_match *out, uint64_t out_total)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (out_total != ctx->match_out_total)
        return ochip_fail(ctx, OCHIP_EINVAL, "out_total %llu differs from the launch (%llu)",
                          (unsigned long long)out_total, (unsigned long long)ctx->match_out_total);
    if (out_total == 0)
        return OCHIP_OK;
    if (!out)
        return ochip_fail(ctx, OCHIP_EINVAL, "out is NULL");
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    OCHIP_HIP(ctx, hipMemcpyAsync(out, ctx->match_out_dev, (size_t)out_total * sizeof(ochip_match),
                                  hipMemcpyDeviceToHost, ctx->stream));
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    return OCHIP_OK;
}

int ochip_match_batch(ochip_ctx *ctx, const ochip_pair *pairs, uint32_t n_pairs, const uint64_t *out_offset,
                      ochip_match *out)
{
    if (!ctx)
        return OCHIP_EINVAL;
    uint64_t total = 0;
    for (uint32_t p = 0; p < n_pairs; p++)
    {
        if (pairs[p].image_1 >= ctx->n_images)
            return ochip_fail(ctx, OCHIP_EINVAL, "pair %u: image out of range", p);
        const uint64_t end = out_offset[p] + ctx->img_n[pairs[p].image_1];
        total = end > total ? end : total;
    }
    int rc = ochip_match_launch(ctx, pairs, n_pairs, out_offset, total);
    if (rc)
        return rc;
    return ochip_match_fetch(ctx, out, total);
}

} // extern "C"
