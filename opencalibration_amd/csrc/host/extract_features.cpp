#include "../env.hpp"
#include "extract_features.hpp"
#include "sort_like_std.hpp"

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

#include <emmintrin.h>
#include <omp.h>

namespace opencalibration_amd
{

namespace
{
// kept features in a bucket grid (cell = NMS radius in full-resolution pixels): the decision of the reference's KD-tree
// query (extract_features.cpp:66-73) is "nearest kept feature farther than the radius", i.e. NO kept feature within it.
// Kept features are farther than one cell side from each other, so a cell holds at most four of them (a fifth would
// have to be within a side of one of the four): a cell is four indices into the coordinate arrays (16 bytes, the 3 x 3
// neighbourhood is three runs of 48 bytes in a table of half a megabyte), no lists to chase, and since the visiting order
// is known the lines of the neighbourhoods a few keypoints ahead are requested early (the visits are in strength order,
// i.e. spatially random: this loop is cache-miss latency, not arithmetic).  The scaled squared distance is monotone in
// the squared distance, so the test on every candidate separately gives the answer of the test on the minimum.
struct nn_grid
{
    static constexpr int CAP = 4;
    static constexpr uint32_t EMPTY = 0xffffffffu;
    long gw, gh;
    uint32_t *cells; // (gh + 2) x (gw + 2) x CAP indices: a border of empty cells saves the clamping
    const double *px, *py;
    const uint32_t *home;           // cell of every point (computed once: two divisions per point, not six)
    std::vector<uint32_t> overflow; // never used by the argument above; kept for safety
    nn_grid(double cell, double minx, double miny, double maxx, double maxy, const double *px_, const double *py_, uint32_t n)
        : px(px_), py(py_)
    {
        gw = (long)std::floor((maxx - minx) / cell) + 3;
        gh = (long)std::floor((maxy - miny) / cell) + 3;
        static thread_local std::vector<uint32_t> buf, homes;
        if (buf.size() < (size_t)(gw * gh) * CAP)
            buf.resize((size_t)(gw * gh) * CAP);
        cells = buf.data();
        std::memset(cells, 0xff, (size_t)(gw * gh) * CAP * sizeof(uint32_t));
        homes.resize(n);
        for (uint32_t i = 0; i < n; i++)
            homes[i] = (uint32_t)(((long)((py[i] - miny) / cell) + 1) * gw + (long)((px[i] - minx) / cell) + 1);
        home = homes.data();
    }
    void prefetch(uint32_t i) const
    {
        const uint32_t *c = cells + (size_t)home[i] * CAP;
        __builtin_prefetch(c - (gw + 1) * CAP);
        __builtin_prefetch(c - (gw - 1) * CAP + CAP - 1);
        __builtin_prefetch(c - CAP);
        __builtin_prefetch(c + 2 * CAP - 1);
        __builtin_prefetch(c + (gw - 1) * CAP);
        __builtin_prefetch(c + (gw + 1) * CAP + CAP - 1);
    }
    void add(uint32_t i)
    {
        uint32_t *c = cells + (size_t)home[i] * CAP;
        for (int e = 0; e < CAP; e++)
            if (c[e] == EMPTY)
            {
                c[e] = i;
                return;
            }
        overflow.push_back(i);
    }
    // true if a kept feature lies within the radius of point i: d2 * scale2 <= radius2, d2 accumulated as the KD-tree
    // does.  Almost all of the 36 slots are empty: the occupied ones of a row are found with three vector compares.
    bool any_within(uint32_t i, double scale2, double radius2) const
    {
        const double x = px[i], y = py[i];
        auto close = [&](uint32_t k) {
            const double dx = x - px[k], dy = y - py[k];
            double d = 0;
            d += dx * dx;
            d += dy * dy;
            return !(d * scale2 > radius2);
        };
        const uint32_t *c0 = cells + (size_t)home[i] * CAP;
        const __m128i empty = _mm_set1_epi32(-1);
        for (long yy = -1; yy <= 1; yy++)
        {
            const uint32_t *row = c0 + (yy * gw - 1) * CAP; // three neighbouring cells = twelve consecutive indices
            unsigned used = 0;
            for (int q = 0; q < 3; q++)
                used |= (unsigned)(0xf ^ _mm_movemask_ps(_mm_castsi128_ps(
                                             _mm_cmpeq_epi32(_mm_loadu_si128((const __m128i *)(row + 4 * q)), empty))))
                        << (4 * q);
            while (used)
            {
                const int q = __builtin_ctz(used);
                used &= used - 1;
                if (close(row[q]))
                    return true;
            }
        }
        for (uint32_t k : overflow)
            if (close(k))
                return true;
        return false;
    }
};

// The host tail of src/extract/extract_features.cpp:38-87 for one image: n device keypoints (k6 = x, y, size,
// angle, response, level in working-image pixels; AKAZE's detection order) -> [sparse..., dense...] features.
double g_tail_prof[5]; // CPU seconds: ordering, NMS, feature records, total (OCHIP_VERBOSE=extract)
static double thread_cpu_now()
{
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct by_response
{
    float response;
    uint32_t index;
};

// non-maximal suppression, extract_features.cpp:58-83, over locations given in strength order (the seeded first keypoint
// is visited again by the loop and therefore also heads the dense list)
void suppress_in_order(const double *lx, const double *ly, uint32_t n, double minx, double miny, double maxx, double maxy, double scale,
                       double nms_pixel_radius, std::vector<uint32_t> &sparse, std::vector<uint32_t> &dense)
{
    nn_grid grid(nms_pixel_radius / scale, minx, miny, maxx, maxy, lx, ly, n);
    sparse.reserve(n);
    dense.reserve(n);
    grid.add(0);
    sparse.push_back(0);
    for (uint32_t i = 0; i < n; i++)
    {
        if (i + 12 < n)
            grid.prefetch(i + 12);
        // nn[0].distance * sqr(scale) > sqr(nms_pixel_radius)
        if (!grid.any_within(i, scale * scale, nms_pixel_radius * nms_pixel_radius))
        {
            grid.add(i);
            sparse.push_back(i);
        }
        else
            dense.push_back(i);
    }
}

void extract_tail(const float *k6, const uint64_t *dd, uint32_t n, double scale, extracted_features &out)
{
    static const bool prof = ochip_verbose("extract");
    const double tp0 = prof ? thread_cpu_now() : 0;
    double tp1 = 0, tp2 = 0;
    const double nms_pixel_radius = 8;
    // The device hands the keypoints over in AKAZE's detection order (level, row, column of the extremum), which is
    // what the reference's unstable std::sort by response starts from - and that starting order decides the result
    // wherever two responses are equal, which happens in almost every image (a birthday effect among 20 k floats).  The
    // order is therefore libstdc++'s introsort's, move for move, computed by sort_like_std (the same algorithm with a
    // branch-free partition scan) on (response, index) records: the permutation only depends on the comparator's
    // answers, which are those of sorting the feature structs themselves.
    std::vector<uint32_t> order(n);
    {
        static thread_local std::vector<by_response> recs;
        recs.resize(n);
        for (uint32_t i = 0; i < n; i++)
            recs[i] = by_response{k6[6 * (size_t)i + 4], i};
        sort_like_std(recs.data(), recs.data() + n, [](const by_response &a, const by_response &c) -> bool { return a.response > c.response; });
        bool tied = false;
        for (uint32_t i = 0; i < n; i++)
        {
            order[i] = recs[i].index;
            tied = tied || (i && recs[i].response == recs[i - 1].response);
        }
        if (prof && tied)
        {
#pragma omp atomic
            g_tail_prof[4] += 1.0;
        }
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
        if (i + 32 < n)
            __builtin_prefetch(k6 + 6 * (size_t)order[i + 32]);
        lx[i] = k6[6 * (size_t)order[i]] / scale;
        ly[i] = k6[6 * (size_t)order[i] + 1] / scale;
        minx = std::min(minx, lx[i]);
        maxx = std::max(maxx, lx[i]);
        miny = std::min(miny, ly[i]);
        maxy = std::max(maxy, ly[i]);
    }
    std::vector<uint32_t> sparse, dense;
    suppress_in_order(lx.data(), ly.data(), n, minx, miny, maxx, maxy, scale, nms_pixel_radius, sparse, dense);
    if (prof)
        tp2 = thread_cpu_now();
    out.num_sparse_features = sparse.size();
    out.features.reserve(sparse.size() + dense.size());
    auto emit = [&](uint32_t i) { // i = position in strength order
        const uint32_t s = order[i];
        feature_2d p; // built in registers and written once (emplace_back() + assignment writes the record twice)
        p.location[0] = lx[i]; // keypoints[i].pt.x / scale, extract_features.cpp:44-45
        p.location[1] = ly[i];
        p.strength = k6[6 * (size_t)s + 4];
        std::memcpy(p.descriptor, dd + 8 * (size_t)s, 64);
        out.features.push_back(p);
    };
    // the records are gathered in strength order, i.e. from random places of the device's arrays: ask for the lines of
    // the entries a few steps ahead while this one is copied
    constexpr size_t PREFETCH_AHEAD = 24;
    auto emit_all = [&](const std::vector<uint32_t> &list) {
        const size_t m = list.size();
        for (size_t j = 0; j < m; j++)
        {
            if (j + PREFETCH_AHEAD < m) // far enough for a DRAM miss (the device's copy does not land in a cache)
            {
                const uint32_t s = order[list[j + PREFETCH_AHEAD]];
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

// The same tail when the device has run it (csrc/features.hip, ochip_akaze_features): `records` is the image's whole output
// list [sparse..., dense...] - the strength order included, libstdc++'s std::sort restated on the device
// (csrc/std_sort.hip) - and the host copies it.  conflict: the image's responses drove introsort to its depth limit, where
// libstdc++ heap-sorts and the device does not follow: the order and the suppression are then computed here, from the
// records (slot[s]: the record of detection index s) and the responses in detection order.
void extract_tail_prepared(const uint8_t *records, const float *response, const uint32_t *slot, uint32_t num_sparse, bool conflict,
                           uint32_t n, double scale, extracted_features &out)
{
    static const bool prof = ochip_verbose("extract");
    const double tp0 = prof ? thread_cpu_now() : 0;
    double tp1 = tp0, tp2 = tp0;
    const double nms_pixel_radius = 8;
    out.features.clear();
    out.num_sparse_features = 0;
    if (n == 0)
        return;
    const feature_2d *R = reinterpret_cast<const feature_2d *>(records);
    if (conflict)
    {
        static thread_local std::vector<by_response> recs;
        recs.resize(n);
        for (uint32_t i = 0; i < n; i++)
            recs[i] = by_response{response[i], i};
        sort_like_std(recs.data(), recs.data() + n, [](const by_response &a, const by_response &c) -> bool { return a.response > c.response; });
        if (prof)
            tp1 = thread_cpu_now();
        static thread_local std::vector<double> lx, ly;
        lx.resize(n);
        ly.resize(n);
        double minx = std::numeric_limits<double>::infinity(), miny = minx, maxx = -minx, maxy = -minx;
        for (uint32_t i = 0; i < n; i++)
        {
            const feature_2d &f = R[slot[recs[i].index]];
            lx[i] = f.location[0];
            ly[i] = f.location[1];
            minx = std::min(minx, lx[i]);
            maxx = std::max(maxx, lx[i]);
            miny = std::min(miny, ly[i]);
            maxy = std::max(maxy, ly[i]);
        }
        std::vector<uint32_t> sparse, dense;
        suppress_in_order(lx.data(), ly.data(), n, minx, miny, maxx, maxy, scale, nms_pixel_radius, sparse, dense);
        if (prof)
            tp2 = thread_cpu_now();
        out.features.reserve((size_t)n + 1);
        for (uint32_t i : sparse)
            out.features.push_back(R[slot[recs[i].index]]);
        out.num_sparse_features = out.features.size();
        for (uint32_t i : dense)
            out.features.push_back(R[slot[recs[i].index]]);
        if (prof)
        {
#pragma omp atomic
            g_tail_prof[4] += 1.0;
        }
    }
    else
    {
        out.features.assign(R, R + (size_t)n + 1);
        out.num_sparse_features = num_sparse;
    }
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
    // or, with the tail prepared on the device (ochip_feature_lists): one page-locked block
    uint8_t *prepared = nullptr;
    ochip_feature_lists lists{};
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
    // Images that start in host memory (the reference's boundary: a cv::Mat per image) are uploaded by the launch sequence
    // that extracts them, one copy per chunk on the sequence's stream.  With chunks of 100 a sequence's 3.6 GB copy takes
    // three times as long as its kernels and the four sequences in flight mostly wait for the link together; chunks of 25
    // keep the PCIe link busy under the other sequences' kernels.
    const uint32_t host_chunk = 25;
    const uint32_t chunk = std::min(images_on_device ? extract_chunk_size() : host_chunk, n_images);
    const size_t image_bytes = (size_t)width * height * 3;

    // Driver threads (one per device context: the caller's and its siblings) keep the device busy - a chunk's
    // result copies over PCIe, table uploads and launch gaps overlap the kernels of the other context's chunk -
    // while this thread's OpenMP team runs the host tail of the finished chunks.  ochip_akaze_batch is a blocking
    // call; every driver owns two result buffers.  Four sequences in flight: staged extraction of the 1 000-image grid
    // 0.270 s with three, 0.245 s with four, no gain from five or six (each sequence holds a 6 GB arena of level planes).
    uint32_t n_drivers = images_on_device ? 4 : 5; // (from host memory a fifth sequence keeps the PCIe link busier: 1 416 -> 1 461 images/s)
    if (const char *e = std::getenv("OCHIP_EXTRACT_STREAMS"))
        n_drivers = (uint32_t)std::max(1L, std::min(5L, std::atol(e))); // siblings 4.. belong to the link runners (load_link.cpp)
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
    // the tail's data-parallel part on the device (csrc/features.hip); OCHIP_TEST_HOOKS=host_tail: all of it here, from the raw
    // keypoint arrays (the two must give the same lists: tests/test_gpu_extract.py)
    const bool device_tail = !ochip_test_hook("host_tail");
    const double nms_pixel_radius = 8;
    const bool force_host_nms = ochip_test_hook("host_nms"); // (test knob: the conflict path for every image)
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
            if (b.prepared)
                ochip_host_free(buf_ctx[i], b.prepared);
            b.kp = nullptr;
            b.desc = nullptr;
            b.prepared = nullptr;
        }
    };
    for (uint32_t i = 0; i < n_bufs; i++)
    {
        chunk_buffers &b = bufs[i];
        buf_ctx[i] = ctxs[i / 2];
        void *p = nullptr, *q = nullptr;
        if (device_tail)
        {
            const size_t rows = (size_t)chunk * max_keypoints;
            const size_t o_resp = (rows + chunk) * 88, o_slot = o_resp + rows * 4, o_ns = o_slot + rows * 4, o_conf = o_ns + (size_t)chunk * 4;
            const size_t pad = ((size_t)chunk + 15) / 16 * 16;
            const size_t o_sub = o_conf + pad, o_nsub = o_sub + (size_t)chunk * OCHIP_SUBSET_CAP * 4, o_sconf = o_nsub + (size_t)chunk * 4;
            if (ochip_host_alloc(buf_ctx[i], o_sconf + pad, &p) != OCHIP_OK)
            {
                release();
                if (error)
                    *error = std::string("ochip_host_alloc: ") + ochip_last_error(buf_ctx[i]);
                return false;
            }
            b.prepared = (uint8_t *)p;
            b.lists.records = b.prepared;
            b.lists.response = (float *)(b.prepared + o_resp);
            b.lists.slot = (uint32_t *)(b.prepared + o_slot);
            b.lists.num_sparse = (uint32_t *)(b.prepared + o_ns);
            b.lists.conflict = b.prepared + o_conf;
            // the 40 px subset LinkStage wants of every image (link_stage.cpp:63-65) comes with the list
            b.lists.subset = (uint32_t *)(b.prepared + o_sub);
            b.lists.num_subset = (uint32_t *)(b.prepared + o_nsub);
            b.lists.subset_conflict = b.prepared + o_sconf;
            b.lists.subset_spacing = 40.0;
            b.counts.resize(chunk);
            continue;
        }
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

    std::vector<double> driver_cpu(n_drivers, 0.0);
    auto driver = [&](uint32_t d) {
        ochip_ctx *dctx = ctxs[d];
        int which = 2 * (int)d;
        const double cpu0 = thread_cpu_now();
        struct at_exit
        {
            double &out, t0;
            ~at_exit() { out = thread_cpu_now() - t0; }
        } record{driver_cpu[d], cpu0};
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
            int rc;
            if (device_tail)
                rc = images_on_device ? ochip_akaze_features_dev(dctx, src, b.n, width, height, max_keypoints, nms_pixel_radius,
                                                                 b.counts.data(), &b.lists, wh)
                                      : ochip_akaze_features(dctx, src, b.n, width, height, max_keypoints, nms_pixel_radius,
                                                             b.counts.data(), &b.lists, wh);
            else
                rc = images_on_device ? ochip_akaze_batch_dev(dctx, src, b.n, width, height, max_keypoints, b.kp, b.desc,
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
            if (device_tail)
            {
                const bool conflict = b.lists.conflict[i] != 0 || force_host_nms;
                extract_tail_prepared(b.lists.records + (size_t)i * ((size_t)max_keypoints + 1) * 88,
                                      b.lists.response + (size_t)i * max_keypoints, b.lists.slot + (size_t)i * max_keypoints,
                                      b.lists.num_sparse[i], conflict, b.counts[i], scale, done[i]);
                if (!conflict && !b.lists.subset_conflict[i] && b.lists.num_subset[i] <= OCHIP_SUBSET_CAP)
                {
                    const uint32_t *sub = b.lists.subset + (size_t)i * OCHIP_SUBSET_CAP;
                    done[i].coarse_subset.assign(sub, sub + b.lists.num_subset[i]);
                    done[i].coarse_spacing = b.lists.subset_spacing;
                }
            }
            else
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
    if (ochip_verbose("extract"))
    {
        fprintf(stderr, "[extract] %u images: host tail %.3f thread-seconds of wall time (%.2f ms per image)\n", n_images,
                tail_cpu_seconds, 1e3 * tail_cpu_seconds / n_images);
        double dc = 0;
        for (double v : driver_cpu)
            dc += v;
        fprintf(stderr, "[extract] CPU seconds of the %u device driver threads (launching and waiting): %.3f\n", n_drivers, dc);
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

// cumulative CPU seconds of the tail's phases (ordering, NMS, feature records, total, images with tied responses) when
// OCHIP_VERBOSE=extract is set
extern "C" void och_extract_tail_profile(double *out5)
{
    for (int i = 0; i < 5; i++)
        out5[i] = opencalibration_amd::g_tail_prof[i];
}

// The strength order alone: order_out = indices 0..n-1 as std::sort by descending response leaves them when it starts
// from 0..n-1 (use_std != 0: std::sort itself; 0: sort_like_std, which must give the same) - for the tests.
extern "C" void och_sort_by_response(const float *response, uint32_t n, uint32_t *order_out, int use_std)
{
    using opencalibration_amd::by_response;
    std::vector<by_response> recs(n);
    for (uint32_t i = 0; i < n; i++)
        recs[i] = by_response{response[i], i};
    auto comp = [](const by_response &a, const by_response &c) -> bool { return a.response > c.response; };
    if (use_std)
        std::sort(recs.begin(), recs.end(), comp);
    else
        opencalibration_amd::sort_like_std(recs.data(), recs.data() + n, comp);
    for (uint32_t i = 0; i < n; i++)
        order_out[i] = recs[i].index;
}

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

// The tail on device-prepared lists (ochip_feature_lists_from_keypoints / ochip_akaze_features) for one image; conflict
// != 0: the suppression is run here.  Output as och_extract_tail.
extern "C" size_t och_extract_tail_prepared(const uint8_t *records, const float *response, const uint32_t *slot, uint32_t num_sparse_in,
                                            int conflict, uint32_t n, double scale, double *loc, float *strength, uint64_t *desc_out,
                                            uint64_t *num_sparse)
{
    opencalibration_amd::extracted_features out;
    opencalibration_amd::extract_tail_prepared(records, response, slot, num_sparse_in, conflict != 0, n, scale, out);
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
