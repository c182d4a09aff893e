// libochip.so — context, device arenas, descriptor store, profiling.  gfx950 only.
#include "ctx.hpp"

#include <cstring>

static std::string g_create_error;

int ochip_fail(ochip_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->error = buf;
    else
        g_create_error = buf;
    return code;
}

int ochip_ensure(ochip_ctx *ctx, void **ptr, size_t *cap, size_t bytes)
{
    if (bytes <= *cap)
        return OCHIP_OK;
    if (*ptr)
        OCHIP_HIP(ctx, hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    if (hipMalloc(ptr, want) != hipSuccess)
        return ochip_fail(ctx, OCHIP_ENOMEM, "hipMalloc(%zu) failed", want);
    *cap = want;
    return OCHIP_OK;
}

int ochip_ensure_keypoint_store(ochip_ctx *ctx, size_t n_keypoints, size_t n_images)
{
    const size_t n = n_keypoints ? n_keypoints : 1, m = n_images ? n_images : 1;
    int rc = ochip_ensure(ctx, (void **)&ctx->kp_xy_dev, &ctx->kp_xy_bytes, n * 16);
    if (rc == OCHIP_OK)
        rc = ochip_ensure(ctx, (void **)&ctx->rays_dev, &ctx->rays_bytes, n * 24);
    if (rc == OCHIP_OK)
        rc = ochip_ensure(ctx, (void **)&ctx->kp_image_dev, &ctx->kp_image_bytes, n * 4);
    if (rc == OCHIP_OK)
        rc = ochip_ensure(ctx, (void **)&ctx->models_dev, &ctx->models_bytes, m * 64);
    if (rc == OCHIP_OK)
        ctx->kp_store_ready = true;
    return rc;
}

void *ochip_pool_get(ochip_ctx *ctx, size_t bytes, size_t *got)
{
    if (bytes == 0)
        bytes = 1;
    int best = -1;
    for (size_t i = 0; i < ctx->dev_pool.size(); i++)
        if (ctx->dev_pool[i].second >= bytes && (best < 0 || ctx->dev_pool[i].second < ctx->dev_pool[best].second))
            best = (int)i;
    if (best >= 0 && ctx->dev_pool[best].second <= 2 * bytes + (1 << 20))
    {
        void *p = ctx->dev_pool[best].first;
        *got = ctx->dev_pool[best].second;
        ctx->dev_pool.erase(ctx->dev_pool.begin() + best);
        return p;
    }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess)
    {
        // the pool may be hoarding memory: drop it and retry once
        for (auto &b : ctx->dev_pool)
            (void)hipFree(b.first);
        ctx->dev_pool.clear();
        if (hipMalloc(&p, bytes) != hipSuccess)
            return nullptr;
    }
    *got = bytes;
    return p;
}

void ochip_pool_put(ochip_ctx *ctx, void *p, size_t bytes)
{
    if (!p)
        return;
    ctx->dev_pool.emplace_back(p, bytes);
    while (ctx->dev_pool.size() > 256)
    {
        size_t s = 0;
        for (size_t k = 1; k < ctx->dev_pool.size(); k++)
            if (ctx->dev_pool[k].second < ctx->dev_pool[s].second)
                s = k;
        (void)hipFree(ctx->dev_pool[s].first);
        ctx->dev_pool.erase(ctx->dev_pool.begin() + s);
    }
}

void ochip_prof_begin(ochip_ctx *ctx, int kid, hipEvent_t *start, hipEvent_t *stop)
{
    auto &s = ctx->prof[kid];
    if (!s.free_list.empty())
    {
        *start = s.free_list.back().first;
        *stop = s.free_list.back().second;
        s.free_list.pop_back();
    }
    else
    {
        (void)hipEventCreate(start);
        (void)hipEventCreate(stop);
    }
    (void)hipEventRecord(*start, ctx->stream);
}

static void prof_drain(ochip_ctx *ctx, int kid)
{
    auto &s = ctx->prof[kid];
    for (auto &p : s.pending)
    {
        float ms = 0;
        (void)hipEventSynchronize(p.second);
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess)
        {
            s.total_ms += ms;
            s.launches++;
        }
        s.free_list.push_back(p);
    }
    s.pending.clear();
}

void ochip_prof_end(ochip_ctx *ctx, int kid, hipEvent_t start, hipEvent_t stop)
{
    (void)hipEventRecord(stop, ctx->stream);
    auto &s = ctx->prof[kid];
    s.pending.emplace_back(start, stop);
    if (s.pending.size() >= 4096)
        prof_drain(ctx, kid);
}

extern "C"
{

int ochip_ctx_create(int device, ochip_ctx **out)
{
    if (!out)
        return ochip_fail(nullptr, OCHIP_EINVAL, "out is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return ochip_fail(nullptr, OCHIP_ENODEV, "no HIP device visible");
    if (device < 0 || device >= count)
        return ochip_fail(nullptr, OCHIP_EINVAL, "device %d out of range (0..%d)", device, count - 1);
    ochip_ctx *ctx = new (std::nothrow) ochip_ctx();
    if (!ctx)
        return ochip_fail(nullptr, OCHIP_ENOMEM, "host allocation failed");
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    const char *bsync = getenv("OCHIP_BLOCKING_SYNC");
    ctx->blocking_wait = !(bsync && bsync[0] == '0');
    if (e == hipSuccess && ctx->blocking_wait) // default on; OCHIP_BLOCKING_SYNC=0 keeps the runtime's polling
    {
        // waits sleep instead of spinning: host threads that wait for the device do not eat into a CPU quota the
        // OpenMP teams of the host phases need (refused harmlessly if the device is already active with other flags)
        (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
        (void)hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipGetDeviceProperties(&ctx->prop, device);
    if (e == hipSuccess && std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0)
    {
        int rc = ochip_fail(nullptr, OCHIP_ENODEV, "device %d is %s; libochip is built for gfx950 only", device,
                            ctx->prop.gcnArchName);
        delete ctx;
        return rc;
    }
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess)
    {
        int rc = ochip_fail(nullptr, OCHIP_EHIP, "context creation failed: %s", hipGetErrorString(e));
        delete ctx;
        return rc;
    }
    *out = ctx;
    return OCHIP_OK;
}

void ochip_ctx_destroy(ochip_ctx *ctx)
{
    if (!ctx)
        return;
    for (ochip_ctx *sib : ctx->siblings)
        ochip_ctx_destroy(sib);
    ctx->siblings.clear();
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto &s : ctx->prof)
    {
        for (auto &p : s.pending)
            s.free_list.push_back(p);
        for (auto &p : s.free_list)
        {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    }
    void *bufs[] = {ctx->desc_dev,      ctx->img_off_dev,   ctx->img_n_dev, ctx->pairs_dev,    ctx->out_off_dev,
                    ctx->match_out_dev, ctx->kp_xy_dev,     ctx->rays_dev,  ctx->kp_image_dev, ctx->models_dev,
                    ctx->ms_recs_dev,   ctx->ms_seg_dev,    ctx->ms_flag_dev};
    for (void *b : bufs)
        if (b)
            (void)hipFree(b);
    for (void *b : ctx->scratch_dev)
        if (b)
            (void)hipFree(b);
    if (ctx->sym_jobs_dev)
        (void)hipFree(ctx->sym_jobs_dev);
    if (ctx->sym_part_dev)
        (void)hipFree(ctx->sym_part_dev);
    for (void *b : {ctx->desc_fp4_dev, ctx->desc_negpop_dev, ctx->desc_pop_dev})
        if (b)
            (void)hipFree(b);
    for (auto &b : ctx->dev_pool)
        (void)hipFree(b.first);
    for (auto &b : ctx->pinned_pool)
        (void)hipHostFree(b.first);
    for (auto &b : ctx->pinned_live)
        (void)hipHostFree(b.first);
    if (ctx->sync_event)
        (void)hipEventDestroy(ctx->sync_event);
    for (hipStream_t r : ctx->retired_streams)
        (void)hipStreamDestroy(r);
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream)
        (void)hipStreamDestroy(ctx->copy_stream);
    delete ctx;
}

int ochip_ctx_set_priority(ochip_ctx *ctx, int high)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (ctx->stream_priority == (high ? 1 : 0))
        return OCHIP_OK; // (asked again by the next survey's runner: the stream stays)
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    int least = 0, greatest = 0;
    OCHIP_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    OCHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipStream_t s = nullptr;
    OCHIP_HIP(ctx, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high ? greatest : least));
    ctx->retired_streams.push_back(ctx->stream); // destroyed with the context, not here: tools that trace the process
    ctx->stream = s;                             // (rocprofv3) keep per-stream state that other threads may still touch
    ctx->stream_priority = high ? 1 : 0;
    return OCHIP_OK;
}

int ochip_ctx_sibling(ochip_ctx *ctx, uint32_t index, ochip_ctx **out)
{
    if (!ctx || !out)
        return OCHIP_EINVAL;
    *out = nullptr;
    // the list grows under a lock: stage runners on several host threads ask for their contexts at the same time (the
    // calls ON a context stay the caller's to serialise; handing contexts out is not)
    std::lock_guard<std::mutex> lock(ctx->siblings_mutex);
    if (index > 64)
        return ochip_fail(ctx, OCHIP_EINVAL, "sibling index %u out of range", index);
    while (ctx->siblings.size() <= index)
    {
        ochip_ctx *s = nullptr;
        const int rc = ochip_ctx_create(ctx->device, &s);
        if (rc != OCHIP_OK)
            return ochip_fail(ctx, rc, "sibling context: %s", ochip_last_error(nullptr));
        ctx->siblings.push_back(s);
    }
    *out = ctx->siblings[index];
    return OCHIP_OK;
}

const char *ochip_last_error(const ochip_ctx *ctx)
{
    return ctx ? ctx->error.c_str() : g_create_error.c_str();
}

int ochip_device_info(const ochip_ctx *ctx, char *name, size_t name_len, int *compute_units, size_t *hbm_bytes)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (name && name_len)
        snprintf(name, name_len, "%s (%s)", ctx->prop.name[0] ? ctx->prop.name : "AMD Instinct (name table not installed)",
                 ctx->prop.gcnArchName); // the marketing name comes from libdrm's amdgpu.ids, absent on some boxes
    if (compute_units)
        *compute_units = ctx->prop.multiProcessorCount;
    if (hbm_bytes)
        *hbm_bytes = ctx->prop.totalGlobalMem;
    return OCHIP_OK;
}

int ochip_synchronize(ochip_ctx *ctx)
{
    if (!ctx)
        return OCHIP_EINVAL;
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->copy_stream));
    return OCHIP_OK;
}

int ochip_descriptors_reserve(ochip_ctx *ctx, uint32_t n_images, uint64_t total_descriptors)
{
    if (!ctx)
        return OCHIP_EINVAL;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    ctx->kp_set.clear();
    ctx->rays_dirty = false;
    ctx->kp_store_ready = false;
    ctx->desc_capacity = ctx->desc_used = 0;
    ctx->fp4_valid = 0;
    ctx->n_images = n_images;
    ctx->img_off.assign(n_images, 0);
    ctx->img_n.assign(n_images, 0);
    ctx->img_set.assign(n_images, 0);
    ctx->img_tables_dirty = true;
    const size_t n_img = n_images ? n_images : 1;
    int rc = ochip_ensure(ctx, (void **)&ctx->desc_dev, &ctx->desc_bytes, (size_t)(total_descriptors ? total_descriptors : 1) * 64);
    if (rc == OCHIP_OK)
        rc = ochip_ensure(ctx, (void **)&ctx->img_off_dev, &ctx->img_off_bytes, n_img * 8);
    if (rc == OCHIP_OK)
        rc = ochip_ensure(ctx, (void **)&ctx->img_n_dev, &ctx->img_n_bytes, n_img * 4);
    if (rc != OCHIP_OK)
        return rc;
    ctx->desc_capacity = total_descriptors;
    return OCHIP_OK;
}

int ochip_upload_descriptors(ochip_ctx *ctx, uint32_t image_id, const uint64_t *desc, uint32_t n)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (image_id >= ctx->n_images)
        return ochip_fail(ctx, OCHIP_EINVAL, "image_id %u >= reserved %u", image_id, ctx->n_images);
    if (ctx->img_set[image_id])
        return ochip_fail(ctx, OCHIP_ESTATE, "image %u already uploaded", image_id);
    if (n && !desc)
        return ochip_fail(ctx, OCHIP_EINVAL, "desc is NULL");
    if (ctx->desc_used + n > ctx->desc_capacity)
        return ochip_fail(ctx, OCHIP_ENOMEM, "descriptor arena full (%llu + %u > %llu)",
                          (unsigned long long)ctx->desc_used, n, (unsigned long long)ctx->desc_capacity);
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    if (n)
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->desc_dev + ctx->desc_used * 16, desc, (size_t)n * 64, hipMemcpyHostToDevice,
                                      ctx->stream));
    ctx->img_off[image_id] = ctx->desc_used;
    ctx->img_n[image_id] = n;
    ctx->img_set[image_id] = 1;
    ctx->desc_used += n;
    ctx->img_tables_dirty = true;
    // pageable source: hipMemcpyAsync has consumed it when it returns only after a sync
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    return OCHIP_OK;
}

int ochip_upload_batch(ochip_ctx *ctx, uint32_t n_images, const uint32_t *counts, const uint64_t *desc_all,
                       const double *xy_all, const double *models8)
{
    if (!ctx || (n_images && (!counts || !models8)))
        return OCHIP_EINVAL;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_images; i++)
        total += counts[i];
    if (total && (!desc_all || !xy_all))
        return ochip_fail(ctx, OCHIP_EINVAL, "NULL descriptor / keypoint array");
    int rc = ochip_descriptors_reserve(ctx, n_images, total);
    if (rc)
        return rc;
    const size_t cap = total ? total : 1;
    rc = ochip_ensure_keypoint_store(ctx, cap, n_images);
    if (rc)
        return rc;
    std::vector<uint32_t> ids(cap);
    uint64_t off = 0;
    for (uint32_t i = 0; i < n_images; i++)
    {
        ctx->img_off[i] = off;
        ctx->img_n[i] = counts[i];
        ctx->img_set[i] = 1;
        for (uint32_t k = 0; k < counts[i]; k++)
            ids[off + k] = i;
        off += counts[i];
    }
    ctx->desc_used = total;
    ctx->kp_set.assign(n_images, 1);
    if (total)
    {
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->desc_dev, desc_all, (size_t)total * 64, hipMemcpyHostToDevice, ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->kp_xy_dev, xy_all, (size_t)total * 16, hipMemcpyHostToDevice, ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->kp_image_dev, ids.data(), (size_t)total * 4, hipMemcpyHostToDevice,
                                      ctx->stream));
    }
    if (n_images)
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->models_dev, models8, (size_t)n_images * 64, hipMemcpyHostToDevice,
                                      ctx->stream));
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    ctx->img_tables_dirty = true;
    ctx->rays_dirty = true;
    return OCHIP_OK;
}

int ochip_host_alloc(ochip_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out)
        return OCHIP_EINVAL;
    *out = nullptr;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    // page-locking is expensive (tens of ms for hundreds of MB): recycle blocks between batches
    int best = -1;
    for (size_t i = 0; i < ctx->pinned_pool.size(); i++)
        if (ctx->pinned_pool[i].second >= bytes && (best < 0 || ctx->pinned_pool[i].second < ctx->pinned_pool[best].second))
            best = (int)i;
    if (best >= 0 && ctx->pinned_pool[best].second <= 2 * bytes + (1 << 20))
    {
        *out = ctx->pinned_pool[best].first;
        ctx->pinned_live.emplace_back(ctx->pinned_pool[best]);
        ctx->pinned_pool.erase(ctx->pinned_pool.begin() + best);
        return OCHIP_OK;
    }
    const size_t want = bytes ? bytes + bytes / 8 : 1;
    if (hipHostMalloc(out, want, hipHostMallocDefault) != hipSuccess)
        return ochip_fail(ctx, OCHIP_ENOMEM, "hipHostMalloc(%zu) failed", want);
    ctx->pinned_live.emplace_back(*out, want);
    return OCHIP_OK;
}

void ochip_host_free(ochip_ctx *ctx, void *p)
{
    if (!ctx || !p)
        return;
    for (size_t i = 0; i < ctx->pinned_live.size(); i++)
        if (ctx->pinned_live[i].first == p)
        {
            ctx->pinned_pool.push_back(ctx->pinned_live[i]);
            ctx->pinned_live.erase(ctx->pinned_live.begin() + i);
            // keep the pool bounded: drop the smallest blocks beyond 16 entries
            while (ctx->pinned_pool.size() > 16)
            {
                size_t s = 0;
                for (size_t k = 1; k < ctx->pinned_pool.size(); k++)
                    if (ctx->pinned_pool[k].second < ctx->pinned_pool[s].second)
                        s = k;
                (void)hipHostFree(ctx->pinned_pool[s].first);
                ctx->pinned_pool.erase(ctx->pinned_pool.begin() + s);
            }
            return;
        }
}

int ochip_descriptor_count(const ochip_ctx *ctx, uint32_t image_id, uint32_t *n)
{
    if (!ctx || !n || image_id >= ctx->n_images || !ctx->img_set[image_id])
        return OCHIP_EINVAL;
    *n = ctx->img_n[image_id];
    return OCHIP_OK;
}

int ochip_profile_reset(ochip_ctx *ctx)
{
    if (!ctx)
        return OCHIP_EINVAL;
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    for (int k = 0; k < OCHIP_K_COUNT; k++)
    {
        prof_drain(ctx, k);
        ctx->prof[k].launches = 0;
        ctx->prof[k].total_ms = 0;
    }
    ctx->match_computed = ctx->match_delivered = 0;
    ctx->relax_mfma_flops = 0;
    ctx->ransac_hyp_corr = ctx->relax_blocks_jac = ctx->relax_blocks_cost = 0;
    for (ochip_ctx *sib : ctx->siblings)
    {
        const int rc = ochip_profile_reset(sib);
        if (rc != OCHIP_OK)
            return rc;
    }
    return OCHIP_OK;
}

int ochip_relax_work(ochip_ctx *ctx, double *mfma_flops)
{
    if (!ctx || !mfma_flops)
        return OCHIP_EINVAL;
    *mfma_flops = ctx->relax_mfma_flops;
    for (ochip_ctx *sib : ctx->siblings)
        *mfma_flops += sib->relax_mfma_flops;
    return OCHIP_OK;
}

int ochip_work_counters(ochip_ctx *ctx, uint64_t *counters3)
{
    if (!ctx || !counters3)
        return OCHIP_EINVAL;
    counters3[0] = ctx->ransac_hyp_corr;
    counters3[1] = ctx->relax_blocks_jac;
    counters3[2] = ctx->relax_blocks_cost;
    for (ochip_ctx *sib : ctx->siblings)
    {
        counters3[0] += sib->ransac_hyp_corr;
        counters3[1] += sib->relax_blocks_jac;
        counters3[2] += sib->relax_blocks_cost;
    }
    return OCHIP_OK;
}

int ochip_relax_memory(ochip_ctx *ctx, uint64_t *unknowns, uint64_t *stored_bytes, uint64_t *dense_bytes)
{
    if (!ctx || !unknowns || !stored_bytes || !dense_bytes)
        return OCHIP_EINVAL;
    *unknowns = *stored_bytes = *dense_bytes = 0;
    auto take = [&](const ochip_ctx *c) {
        if (c->relax_system_unknowns > *unknowns)
        {
            *unknowns = c->relax_system_unknowns;
            *stored_bytes = c->relax_system_bytes;
            *dense_bytes = c->relax_system_dense_bytes;
        }
    };
    take(ctx);
    for (ochip_ctx *sib : ctx->siblings)
        take(sib);
    return OCHIP_OK;
}

int ochip_match_work(ochip_ctx *ctx, uint64_t *computed, uint64_t *delivered)
{
    if (!ctx)
        return OCHIP_EINVAL;
    uint64_t c = ctx->match_computed, d = ctx->match_delivered;
    for (ochip_ctx *sib : ctx->siblings)
    {
        c += sib->match_computed;
        d += sib->match_delivered;
    }
    if (computed)
        *computed = c;
    if (delivered)
        *delivered = d;
    return OCHIP_OK;
}

int ochip_profile_get(ochip_ctx *ctx, int kernel_id, uint64_t *launches, double *total_ms)
{
    if (!ctx || kernel_id < 0 || kernel_id >= OCHIP_K_COUNT)
        return OCHIP_EINVAL;
    prof_drain(ctx, kernel_id);
    uint64_t n = ctx->prof[kernel_id].launches;
    double ms = ctx->prof[kernel_id].total_ms;
    for (ochip_ctx *sib : ctx->siblings) // launches made through sibling contexts count for their owner
    {
        uint64_t sn = 0;
        double sms = 0;
        const int rc = ochip_profile_get(sib, kernel_id, &sn, &sms);
        if (rc != OCHIP_OK)
            return rc;
        n += sn;
        ms += sms;
    }
    if (launches)
        *launches = n;
    if (total_ms)
        *total_ms = ms;
    return OCHIP_OK;
}

} // extern "C"
