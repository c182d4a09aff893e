// Internal context of libochip.so (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ochip.h"
#include "env.hpp"

struct ochip_profile_slot
{
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_list;
    uint64_t launches = 0;
    double total_ms = 0;
};

struct ochip_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;     // compute stream: every kernel of the hot path is launched here
    hipStream_t copy_stream = nullptr;
    std::vector<hipStream_t> retired_streams; // replaced by ochip_ctx_set_priority
    int stream_priority = -1; // what ochip_ctx_set_priority last set (-1: the default stream)
    hipEvent_t sync_event = nullptr;  // ochip_stream_wait: a blocking-sync event (created on first use)
    bool blocking_wait = true;        // OCHIP_BLOCKING_SYNC=0: let the runtime poll instead
    std::string error;
    hipDeviceProp_t prop{};

    // descriptor arena: [total][16] u32, image i at img_off[i] with img_n[i] descriptors
    uint32_t *desc_dev = nullptr;
    uint64_t desc_capacity = 0, desc_used = 0;
    uint32_t n_images = 0;
    std::vector<uint64_t> img_off;
    std::vector<uint32_t> img_n;
    std::vector<uint8_t> img_set;
    uint64_t *img_off_dev = nullptr;
    uint32_t *img_n_dev = nullptr;
    bool img_tables_dirty = true;

    // keypoint store, same indexing as the descriptor arena: pixel xy, owning image, unit rays
    double *kp_xy_dev = nullptr;
    double *rays_dev = nullptr;
    uint32_t *kp_image_dev = nullptr;
    double *models_dev = nullptr; // [n_images][8]: f, ppx, ppy, k1, k2, k3, p1, p2
    std::vector<uint8_t> kp_set;
    bool rays_dirty = false;
    bool kp_store_ready = false; // keypoint buffers sized for the current reservation
    // capacities (bytes) of the grow-only buffers above: a reservation reuses them instead of hipFree / hipMalloc,
    // which synchronise the whole device and would stall the other contexts' streams
    size_t desc_bytes = 0, img_off_bytes = 0, img_n_bytes = 0, kp_xy_bytes = 0, rays_bytes = 0, kp_image_bytes = 0,
           models_bytes = 0;

    // match scratch
    ochip_pair *pairs_dev = nullptr;
    uint64_t *out_off_dev = nullptr;
    size_t pairs_cap = 0;
    ochip_match *match_out_dev = nullptr;
    size_t match_out_cap = 0;
    uint64_t match_out_total = 0;
    // ochip_match_sort (match_sort.hip): per pair the matches that pass the ratio test as (count << 32 | query) records at the
    // pair's offset, in match_features_subset's output order; the pairs' offsets and match counts
    void *ms_recs_dev = nullptr, *ms_seg_dev = nullptr, *ms_flag_dev = nullptr;
    size_t ms_recs_cap = 0, ms_seg_cap = 0, ms_flag_cap = 0;
    uint32_t ms_pairs = 0;
    void *sym_jobs_dev = nullptr, *sym_part_dev = nullptr; // symmetric pairs of a match launch: job table, column partials
    size_t sym_jobs_cap = 0, sym_part_cap = 0;
    // operands of the matrix-core matcher (match.hip, hamming_2nn_mfma_kernel), same indexing as the descriptor arena: the
    // descriptor's 512 bits as FP4 values 0 / 1 (256 bytes), (512 - popcount) * 8192 as a float, the popcount; features
    // [0, fp4_valid) are expanded, the rest is done by the next match launch
    void *desc_fp4_dev = nullptr, *desc_negpop_dev = nullptr, *desc_pop_dev = nullptr;
    size_t desc_fp4_cap = 0, desc_negpop_cap = 0, desc_pop_cap = 0;
    uint64_t fp4_valid = 0;

    // generic scratch for the RANSAC / relax kernels (grown on demand)
    void *scratch_dev[8] = {nullptr};
    size_t scratch_cap[8] = {0};

    // page-locked host blocks handed out by ochip_host_alloc (live) and recycled ones (pool)
    std::vector<std::pair<void *, size_t>> pinned_live, pinned_pool;
    // recycled device blocks for per-call temporaries (hipMalloc / hipFree are slow and synchronising)
    std::vector<std::pair<void *, size_t>> dev_pool;

    ochip_profile_slot prof[OCHIP_K_COUNT];
    // descriptor distances the match launches computed / delivered since the last profile reset (a pair matched in both
    // directions from one pass computes its n1 x n2 distances once and delivers them twice)
    uint64_t match_computed = 0, match_delivered = 0;
    // fp64 multiply-adds x 2 the Cholesky factorisations of the relax solves issued on the matrix cores (panel and
    // trailing-update GEMMs over the rows inside the block envelope) since the last profile reset
    double relax_mfma_flops = 0;
    // roofline bookkeeping since the last profile reset: RANSAC loop trips x correspondences of their job (an upper bound of the
    // (hypothesis, correspondence) errors evaluated: the SPRT exit of ransac.cpp:197-200 leaves a hypothesis early), residual
    // blocks the relax evaluation kernels processed with / without Jacobians
    uint64_t ransac_hyp_corr = 0, relax_blocks_jac = 0, relax_blocks_cost = 0;
    uint64_t relax_system_bytes = 0, relax_system_dense_bytes = 0, relax_system_unknowns = 0; // largest reduced system held (relax_lm.hip)

    // sibling contexts on the same device (own streams, scratch and pools) handed out by ochip_ctx_sibling so
    // that independent batches can be in flight at once; owned by this context
    std::vector<ochip_ctx *> siblings;
    std::mutex siblings_mutex; // ochip_ctx_sibling may be called from concurrent runner threads (RelaxStage's group runners)
};


// Wait for a stream without spinning: an event created with hipEventBlockingSync puts the waiting host thread to sleep
// whatever scheduling flags the device's primary context was created with (hipSetDeviceFlags is refused once another
// library - PyTorch, RCCL - has initialised the device, which is exactly the multi-GPU case).  The threads that wait
// here would otherwise eat the CPU quota the OpenMP teams of the host phases need.
inline hipError_t ochip_stream_wait(ochip_ctx *ctx, hipStream_t st)
{
    if (!ctx->blocking_wait)
        return hipStreamSynchronize(st);
    if (!ctx->sync_event)
    {
        const hipError_t e = hipEventCreateWithFlags(&ctx->sync_event, hipEventBlockingSync | hipEventDisableTiming);
        if (e != hipSuccess)
            return e;
    }
    const hipError_t e = hipEventRecord(ctx->sync_event, st);
    return e != hipSuccess ? e : hipEventSynchronize(ctx->sync_event);
}

int ochip_fail(ochip_ctx *ctx, int code, const char *fmt, ...);
int ochip_ensure(ochip_ctx *ctx, void **ptr, size_t *cap, size_t bytes); // grow-only device buffer
int ochip_ensure_keypoint_store(ochip_ctx *ctx, size_t n_keypoints, size_t n_images); // kp_xy, rays, kp_image, models
void *ochip_pool_get(ochip_ctx *ctx, size_t bytes, size_t *got);          // device block from the pool (or hipMalloc); nullptr on failure
void ochip_pool_put(ochip_ctx *ctx, void *p, size_t bytes);              // hand it back (kept for reuse, freed with the context)
void ochip_prof_begin(ochip_ctx *ctx, int kid, hipEvent_t *start, hipEvent_t *stop);
void ochip_prof_end(ochip_ctx *ctx, int kid, hipEvent_t start, hipEvent_t stop);

#define OCHIP_HIP(ctx, call)                                                                                           \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e__ = (call);                                                                                       \
        if (e__ != hipSuccess)                                                                                         \
            return ochip_fail((ctx), OCHIP_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,    \
                              __LINE__);                                                                               \
    } while (0)

namespace ochip
{
// features.hip: the tail of extract_features prepared on the device for B images whose compacted keypoints lie in HBM
// (enqueued on the context's stream, results copied into `out`; the caller waits and returns `allocs` to the pool)
int feature_lists_enqueue(ochip_ctx *ctx, std::vector<std::pair<void *, size_t>> *allocs, uint32_t B, uint32_t max_kp,
                          const float *d_kp6, const unsigned long long *d_desc, const unsigned int *d_counts, uint32_t most,
                          int work_w, int work_h, double scale, double nms_radius, const ochip_feature_lists *out);
} // namespace ochip

namespace ochip
{
// std_sort.hip: libstdc++'s std::sort (comp(a, b) = high half of a > high half of b) on segments of 64-bit records in HBM
int std_sort_enqueue(ochip_ctx *ctx, std::vector<std::pair<void *, size_t>> *allocs, unsigned long long *recs, size_t total_len,
                     const unsigned int *seg_begin, const unsigned int *seg_end, uint32_t n_segs, uint32_t max_len,
                     unsigned char *fallback);
} // namespace ochip
