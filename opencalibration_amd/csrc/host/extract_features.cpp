#include "extract_features.hpp"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <mutex>
#include <thread>

#include <omp.h>

namespace opencalibration_amd
{

namespace
{
// kept features in a bucket grid (cell = NMS radius in full-resolution pixels): the exact nearest-neighbour
// squared distance is what enters the decision, as with the reference's KD-tree (extract_features.cpp:66-73)
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
    double nearest2(double x, double y) const
    {
        const long cx = (long)((x - minx) / cell), cy = (long)((y - miny) / cell);
        double best = std::numeric_limits<double>::infinity();
        for (long yy = std::max(cy - 1, 0L); yy <= std::min(cy + 1, (long)gh - 1); yy++)
            for (long xx = std::max(cx - 1, 0L); xx <= std::min(cx + 1, (long)gw - 1); xx++)
                for (int32_t e = head[(size_t)yy * gw + xx]; e >= 0; e = next[e])
                {
                    const double dx = x - pts[e].first, dy = y - pts[e].second;
                    double d = 0;
                    d += dx * dx;
                    d += dy * dy;
                    best = std::min(best, d);
                }
        return best;
    }
};

// The host tail of src/extract/extract_features.cpp:38-87 for one image: n device keypoints (k6 = x, y, size,
// angle, response, level in working-image pixels; AKAZE's detection order) -> [sparse..., dense...] features.
static double g_tail_prof[5]; // CPU seconds: ordering, NMS, feature records, total (OCHIP_EXTRACT_VERBOSE)
static double thread_cpu_now()
{
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void extract_tail(const float *k6, const uint64_t *dd, uint32_t n, double scale, extracted_features &out)
{
    static const bool prof = std::getenv("OCHIP_EXTRACT_VERBOSE") != nullptr;
    const double tp0 = prof ? thread_cpu_now() : 0;
    double tp1 = 0, tp2 = 0;
    const double nms_pixel_radius = 8;
    // The device hands the keypoints over in AKAZE's detection order (level, row, column of the extremum), which is
    // what the reference's unstable std::sort by response starts from.  That starting order only matters when two
    // responses are equal, so the responses are first ordered with a radix sort of their bit patterns (positive floats
    // order like unsigned integers; three counting passes over 16 k keys); an image in which two keypoints share a
    // response exactly (most of the rendered benchmark views do) takes the reference's route: std::sort of the
    // indices 0..n-1, whose compare/move sequence depends only on the comparator's answers and therefore yields the
    // permutation that sorting the feature structs themselves would.
    std::vector<uint32_t> order(n);
    bool unique_responses = n > 0;
    {
        static thread_local std::vector<uint64_t> a, c; // ~response bits << 32 | index: ascending = strongest first
        a.resize(n);
        c.resize(n);
        for (uint32_t i = 0; i < n; i++)
        {
            uint32_t rb;
            std::memcpy(&rb, k6 + 6 * (size_t)i + 4, 4);
            if (rb & 0x80000000u) // a negative (or -0) response does not order like its bits: comparison route
                unique_responses = false;
            a[i] = ((uint64_t)(~rb) << 32) | i;
        }
        if (unique_responses)
        {
            for (int pass = 0; pass < 3; pass++)
            {
                const int shift = 32 + 11 * pass;
                uint32_t count[2048] = {0};
                for (uint32_t i = 0; i < n; i++)
                    count[(a[i] >> shift) & 2047]++;
                uint32_t sum = 0;
                for (uint32_t d = 0; d < 2048; d++)
                {
                    const uint32_t t = count[d];
                    count[d] = sum;
                    sum += t;
                }
                for (uint32_t i = 0; i < n; i++)
                    c[count[(a[i] >> shift) & 2047]++] = a[i];
                a.swap(c);
            }
            for (uint32_t i = 0; i < n; i++)
            {
                order[i] = (uint32_t)a[i];
                if (i && (a[i] >> 32) == (a[i - 1] >> 32))
                    unique_responses = false;
            }
        }
    }
    if (!unique_responses)
    {
        if (prof)
        {
#pragma omp atomic
            g_tail_prof[4] += 1.0;
        }
        // the records carry the response next to the index: the comparator then touches one cache line per element
        // (the permutation only depends on the comparator's answers, which are those of sorting the feature structs)
        struct by_response
        {
            float response;
            uint32_t index;
        };
        static thread_local std::vector<by_response> recs;
        recs.resize(n);
        for (uint32_t i = 0; i < n; i++)
            recs[i] = by_response{k6[6 * (size_t)i + 4], i};
        std::sort(recs.begin(), recs.end(), [](const by_response &a, const by_response &c) -> bool { return a.response > c.response; });
        for (uint32_t i = 0; i < n; i++)
            order[i] = recs[i].index;
    }
    if (prof)
        tp1 = thread_cpu_now();
    out.features.clear();
    out.num_sparse_features = 0;
    if (n == 0)
        return;
    // non-maximal suppression, extract_features.cpp:58-83 (the seeded first keypoint is visited again by the
    // loop and therefore also heads the dense list)
    static thread_local std::vector<double> lx, ly;
    lx.resize(n);
    ly.resize(n);
    double minx = std::numeric_limits<double>::infinity(), miny = minx, maxx = -minx, maxy = -minx;
    for (uint32_t i = 0; i < n; i++)
    {
        if (i + 16 < n)
            __builtin_prefetch(k6 + 6 * (size_t)order[i + 16]);
        lx[i] = k6[6 * (size_t)order[i]] / scale;
        ly[i] = k6[6 * (size_t)order[i] + 1] / scale;
        minx = std::min(minx, lx[i]);
        maxx = std::max(maxx, lx[i]);
        miny = std::min(miny, ly[i]);
        maxy = std::max(maxy, ly[i]);
    }
    nn_grid grid(nms_pixel_radius / scale, minx, miny, maxx, maxy);
    std::vector<uint32_t> sparse, dense;
    sparse.reserve(n);
    dense.reserve(n);
    grid.add(lx[0], ly[0]);
    sparse.push_back(0);
    for (uint32_t i = 0; i < n; i++)
    {
        // nn[0].distance * sqr(scale) > sqr(nms_pixel_radius)
        if (grid.nearest2(lx[i], ly[i]) * (scale * scale) > nms_pixel_radius * nms_pixel_radius)
        {
            grid.add(lx[i], ly[i]);
            sparse.push_back(i);
        }
        else
            dense.push_back(i);
    }
    if (prof)
        tp2 = thread_cpu_now();
    out.num_sparse_features = sparse.size();
    out.features.reserve(sparse.size() + dense.size());
    auto emit = [&](uint32_t i) { // i = position in strength order
        const uint32_t s = order[i];
        out.features.emplace_back();
        feature_2d &p = out.features.back();
        p.location[0] = lx[i]; // keypoints[i].pt.x / scale, extract_features.cpp:44-45
        p.location[1] = ly[i];
        p.strength = k6[6 * (size_t)s + 4];
        std::memcpy(p.descriptor, dd + 8 * (size_t)s, 64);
    };
    // the records are gathered in strength order, i.e. from random places of the device's arrays: ask for the lines of
    // the entries a few steps ahead while this one is copied
    auto emit_all = [&](const std::vector<uint32_t> &list) {
        const size_t m = list.size();
        for (size_t j = 0; j < m; j++)
        {
            if (j + 8 < m)
            {
                const uint32_t s = order[list[j + 8]];
                __builtin_prefetch(dd + 8 * (size_t)s);
                __builtin_prefetch(k6 + 6 * (size_t)s + 4);
            }
            emit(list[j]);
        }
    };
    emit_all(sparse);
    emit_all(dense);
    if (prof)
    {
        const double tp3 = thread_cpu_now();
        const double d[4] = {tp1 - tp0, tp2 - tp1, tp3 - tp2, tp3 - tp0};
        for (int i = 0; i < 4; i++)
        {
#pragma omp atomic
            g_tail_prof[i] += d[i];
        }
    }
}

struct chunk_buffers
{
    float *kp = nullptr;      // page-locked: chunk x max_keypoints x 6
    uint64_t *desc = nullptr; // page-locked: chunk x max_keypoints x 8
    std::vector<uint32_t> counts;
    uint32_t first = 0, n = 0;
};
} // namespace

uint32_t extract_chunk_size()
{
    if (const char *e = std::getenv("OCHIP_EXTRACT_CHUNK"))
    {
        const long v = std::atol(e);
        if (v >= 1 && v <= 1024)
            return (uint32_t)v;
    }
    return 100;
}

std::vector<extracted_features> extract_features_batch(ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images,
                                                       int width, int height, uint32_t max_keypoints, std::string *error,
                                                       bool images_on_device)
{
    std::vector<extracted_features> out(n_images);
    const bool ok = extract_features_stream(ctx, images_bgr, n_images, width, height, max_keypoints, images_on_device, 0,
                                            [&](uint32_t first, uint32_t count, extracted_features *f) {
                                                for (uint32_t i = 0; i < count; i++)
                                                    out[first + i] = std::move(f[i]);
                                            },
                                            error);
    if (!ok)
        return {};
    return out;
}

bool extract_features_stream(ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images, int width, int height,
                             uint32_t max_keypoints, bool images_on_device, int host_threads,
                             const std::function<void(uint32_t, uint32_t, extracted_features *)> &on_chunk,
                             std::string *error)
{
    if (n_images == 0)
        return true;
    if (width <= 0 || height <= 0) // image.empty(): {results, 0}, extract_features.cpp:20-23
    {
        std::vector<extracted_features> empty(n_images);
        on_chunk(0, n_images, empty.data());
        return true;
    }
    const int tail_threads = host_threads > 0 ? host_threads : omp_get_max_threads();
    const int max_length_pixels = 1600;
    const double scale = std::min(1.f, float(max_length_pixels) / (float)std::max(width, height));
    const uint32_t chunk = std::min(extract_chunk_size(), n_images);
    const size_t image_bytes = (size_t)width * height * 3;

    // Driver threads (one per device context: the caller's and its siblings) keep the device busy - a chunk's
    // result copies over PCIe, table uploads and launch gaps overlap the kernels of the other context's chunk -
    // while this thread's OpenMP team runs the host tail of the finished chunks.  ochip_akaze_batch is a blocking
    // call; every driver owns two result buffers.
    uint32_t n_drivers = 3;
    if (const char *e = std::getenv("OCHIP_EXTRACT_STREAMS"))
        n_drivers = (uint32_t)std::max(1L, std::min(4L, std::atol(e)));
    const uint32_t n_chunks = (n_images + chunk - 1) / chunk;
    n_drivers = std::min(n_drivers, n_chunks);
    std::vector<ochip_ctx *> ctxs(n_drivers, ctx);
    for (uint32_t d = 1; d < n_drivers; d++)
        if (ochip_ctx_sibling(ctx, d - 1, &ctxs[d]) != OCHIP_OK)
        {
            if (error)
                *error = std::string("ochip_ctx_sibling: ") + ochip_last_error(ctx);
            return false;
        }
    const uint32_t n_bufs = 2 * n_drivers;
    std::vector<chunk_buffers> bufs(n_bufs);
    std::vector<ochip_ctx *> buf_ctx(n_bufs);
    std::string fail;
    auto release = [&]() {
        for (uint32_t i = 0; i < n_bufs; i++)
        {
            chunk_buffers &b = bufs[i];
            if (b.kp)
                ochip_host_free(buf_ctx[i], b.kp);
            if (b.desc)
                ochip_host_free(buf_ctx[i], b.desc);
            b.kp = nullptr;
            b.desc = nullptr;
        }
    };
    for (uint32_t i = 0; i < n_bufs; i++)
    {
        chunk_buffers &b = bufs[i];
        buf_ctx[i] = ctxs[i / 2];
        void *p = nullptr, *q = nullptr;
        if (ochip_host_alloc(buf_ctx[i], (size_t)chunk * max_keypoints * 6 * sizeof(float), &p) != OCHIP_OK ||
            ochip_host_alloc(buf_ctx[i], (size_t)chunk * max_keypoints * 8 * sizeof(uint64_t), &q) != OCHIP_OK)
        {
            if (p)
                ochip_host_free(buf_ctx[i], p);
            release();
            if (error)
                *error = std::string("ochip_host_alloc: ") + ochip_last_error(buf_ctx[i]);
            return false;
        }
        b.kp = (float *)p;
        b.desc = (uint64_t *)q;
        b.counts.resize(chunk);
    }
    double tail_cpu_seconds = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> ready; // filled buffers
    std::vector<char> buffer_free(n_bufs, 1);
    uint32_t next_chunk = 0, drivers_done = 0;

    auto driver = [&](uint32_t d) {
        ochip_ctx *dctx = ctxs[d];
        int which = 2 * (int)d;
        for (;;)
        {
            uint32_t c;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return buffer_free[which] != 0; });
                if (!fail.empty() || next_chunk >= n_chunks)
                    break;
                c = next_chunk++;
                buffer_free[which] = 0;
            }
            chunk_buffers &b = bufs[which];
            b.first = c * chunk;
            b.n = std::min(chunk, n_images - b.first);
            int wh[2];
            const uint8_t *src = images_bgr + (size_t)b.first * image_bytes;
            const int rc = images_on_device ? ochip_akaze_batch_dev(dctx, src, b.n, width, height, max_keypoints, b.kp, b.desc,
                                                                    b.counts.data(), wh)
                                            : ochip_akaze_batch(dctx, src, b.n, width, height, max_keypoints, b.kp, b.desc,
                                                                b.counts.data(), wh);
            std::unique_lock<std::mutex> lk(mu);
            if (rc != OCHIP_OK)
            {
                if (fail.empty())
                    fail = std::string("ochip_akaze_batch: ") + ochip_last_error(dctx);
                buffer_free[which] = 1;
                break;
            }
            ready.push_back(which);
            cv.notify_all();
            which = 2 * (int)d + ((which + 1) & 1);
        }
        std::unique_lock<std::mutex> lk(mu);
        drivers_done++;
        cv.notify_all();
    };
    std::vector<std::thread> drivers;
    for (uint32_t d = 0; d < n_drivers; d++)
        drivers.emplace_back(driver, d);

    for (;;)
    {
        int which;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !ready.empty() || drivers_done == n_drivers; });
            if (ready.empty())
                break;
            which = ready.front();
            ready.pop_front();
        }
        chunk_buffers &b = bufs[which];
        double cpu = 0;
        std::vector<extracted_features> done(b.n);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : cpu) num_threads(tail_threads)
        for (uint32_t i = 0; i < b.n; i++)
        {
            const double t0 = omp_get_wtime();
            extract_tail(b.kp + (size_t)i * max_keypoints * 6, b.desc + (size_t)i * max_keypoints * 8, b.counts[i], scale,
                         done[i]);
            cpu += omp_get_wtime() - t0;
        }
        tail_cpu_seconds += cpu;
        const uint32_t chunk_first = b.first, chunk_n = b.n;
        {
            std::unique_lock<std::mutex> lk(mu);
            buffer_free[which] = 1;
            cv.notify_all();
        }
        on_chunk(chunk_first, chunk_n, done.data()); // the device is already busy with the next chunks
    }
    for (auto &t : drivers)
        t.join();
    if (std::getenv("OCHIP_EXTRACT_VERBOSE"))
    {
        fprintf(stderr, "[extract] %u images: host tail %.3f thread-seconds of wall time (%.2f ms per image)\n", n_images,
                tail_cpu_seconds, 1e3 * tail_cpu_seconds / n_images);
        fprintf(stderr, "[extract] cumulative CPU seconds of the tail: ordering %.3f, NMS %.3f, feature records %.3f, total %.3f; images with tied responses %.0f\n",
                g_tail_prof[0], g_tail_prof[1], g_tail_prof[2], g_tail_prof[3], g_tail_prof[4]);
    }
    release();
    if (!fail.empty())
    {
        if (error)
            *error = fail;
        return false;
    }
    return true;
}

} // namespace opencalibration_amd

// The host tail alone (no device): what extract_features.cpp:38-87 does with the keypoints cv::AKAZE returned, given
// as kp6 rows {x, y, size, angle, response, level} in detection order.  For the CPU-side tests.
extern "C" size_t och_extract_tail(const float *kp6, const uint64_t *desc, uint32_t n, double scale, double *loc, float *strength,
                                   uint64_t *desc_out, uint64_t *num_sparse)
{
    opencalibration_amd::extracted_features out;
    opencalibration_amd::extract_tail(kp6, desc, n, scale, out);
    for (size_t i = 0; i < out.features.size(); i++)
    {
        loc[2 * i] = out.features[i].location[0];
        loc[2 * i + 1] = out.features[i].location[1];
        strength[i] = out.features[i].strength;
        std::memcpy(desc_out + 8 * i, out.features[i].descriptor, 64);
    }
    *num_sparse = out.num_sparse_features;
    return out.features.size();
}
