#include "../env.hpp"
#include "link_stage.hpp"
#include "sort_like_std.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <tuple>

#include <omp.h>

namespace opencalibration_amd
{

namespace
{
// page-locked host buffer from the device library (freed on scope exit)
template <typename T> struct pinned
{
    ochip_ctx *ctx;
    T *ptr = nullptr;
    pinned(ochip_ctx *c, size_t n) : ctx(c)
    {
        void *p = nullptr;
        if (ochip_host_alloc(c, n * sizeof(T), &p) == OCHIP_OK)
            ptr = (T *)p;
    }
    ~pinned()
    {
        ochip_host_free(ctx, ptr);
    }
    pinned(const pinned &) = delete;
    pinned &operator=(const pinned &) = delete;
};
using clk = std::chrono::steady_clock;
double since(clk::time_point t0)
{
    return std::chrono::duration<double>(clk::now() - t0).count();
}
} // namespace

LinkStage::LinkStage(ochip_ctx *ctx, int runners) : _ctx(ctx), _runners(runners)
{
    if (_runners <= 0)
    {
        const char *e = std::getenv("OCHIP_LINK_RUNNERS");
        _runners = e ? std::atoi(e) : 3;
    }
    _runners = std::max(1, std::min(_runners, 16));
}

void LinkStage::init(const MeasurementGraph &graph, const std::vector<size_t> &node_ids)
{
    const auto t0 = clk::now();
    _links.clear();
    _links.reserve(node_ids.size());
    const auto &nodes = graph.nodes();
    std::vector<std::pair<double, size_t>> dist(nodes.size());
    for (size_t node_id : node_ids)
    {
        const image &img = graph.getNode(node_id)->payload;
        for (size_t j = 0; j < nodes.size(); j++)
        {
            // SquaredL2 of the reference's KD-tree (external/jk-tree/include/jk/KDTree.h:681-691)
            const double dx = img.position[0] - nodes[j].payload.position[0];
            const double dy = img.position[1] - nodes[j].payload.position[1];
            double d = 0;
            d += dx * dx;
            d += dy * dy;
            dist[j] = {d, j};
        }
        const size_t k = std::min<size_t>(10, dist.size());
        // a total order: distance, then the place in the graph (equidistant neighbours are the rule on a regular survey
        // grid; the reference's KD-tree breaks such ties by its traversal - here, and in parallel.knn_pairs, the image
        // added first wins, so the C++ stage and the Python sharding always agree on the pair set)
        std::partial_sort(dist.begin(), dist.begin() + k, dist.end());
        NodeLinks link;
        link.node_id = node_id;
        link.link_ids.reserve(k);
        for (size_t j = 0; j < k; j++)
            if (nodes[dist[j].second].id != node_id)
                link.link_ids.push_back(nodes[dist[j].second].id);
        _links.emplace_back(std::move(link));
    }
    timers.link_init += since(t0);
}

void LinkStage::prepare_index(const MeasurementGraph &graph)
{
    _prepared_index.clear();
    size_t n = 0;
    auto add = [&](size_t id) {
        if (graph.getNode(id) != nullptr && _prepared_index.emplace(id, n).second)
            n++;
    };
    for (const auto &link : _links)
    {
        add(link.node_id);
        for (size_t m : link.link_ids)
            add(m);
    }
    _subsets.assign(n, {});
    _rays.assign(n, {});
    _remote.assign(n, {});
}

void LinkStage::prepare_images(const MeasurementGraph &graph, const std::vector<size_t> &node_ids, int threads)
{
    const auto t0 = clk::now();
    const double coarse_spacing_pixels = 40.0;
    const bool use_host_subset = ochip_test_hook("host_subset"); // (tests)
    const int nt = threads > 0 ? threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (size_t k = 0; k < node_ids.size(); k++)
    {
        const auto it = _prepared_index.find(node_ids[k]);
        if (it == _prepared_index.end())
            continue;
        const size_t s = it->second;
        const image &img = graph.getNode(node_ids[k])->payload;
        if (img.coarse_spacing == coarse_spacing_pixels && !use_host_subset) // computed on the device with the feature list
            _subsets[s].assign(img.coarse_subset.begin(), img.coarse_subset.end());
        else
            _subsets[s] = spatially_subsample_feature_indices(img.features, coarse_spacing_pixels, img.num_sparse_features);
        _rays[s].resize(_subsets[s].size() * 3);
        for (size_t q = 0; q < _subsets[s].size(); q++)
            image_to_3d(img.features[_subsets[s][q]].location, *img.model, &_rays[s][3 * q]);
    }
    const double dt = since(t0);
    std::lock_guard<std::mutex> lock(_measurement_mutex);
    timers.subsample += dt;
}

void LinkStage::prepare(const MeasurementGraph &graph)
{
    prepare_index(graph);
    std::vector<size_t> ids(_prepared_index.size());
    for (const auto &kv : _prepared_index)
        ids[kv.second] = kv.first;
    prepare_images(graph, ids, 0);
}

void LinkStage::run_range(const MeasurementGraph &graph, size_t link_begin, size_t link_end, ochip_ctx *ctx, int omp_threads)
{
    std::vector<link_pair> pairs;
    for (size_t i = link_begin; i < link_end; i++)
        for (size_t match_node_id : _links[i].link_ids)
            pairs.emplace_back(i, match_node_id);
    run_batch(graph, pairs, ctx, std::max(1, omp_threads));
}

void LinkStage::run_pairs(const MeasurementGraph &graph, const std::vector<link_pair> &pairs, ochip_ctx *ctx, int omp_threads)
{
    run_batch(graph, pairs, ctx, std::max(1, omp_threads));
}

std::vector<std::function<void()>> LinkStage::get_runners(const MeasurementGraph &graph)
{
    std::vector<std::function<void()>> funcs;
    prepare(graph);
    // a runner needs enough pairs to fill the device; the debug record is kept in runner order, so one runner then.
    // One batch per runner (walking a range in sub-batches so that the runners drift out of phase did not pay on C3:
    // smaller launches, more uploads), the OpenMP team divided among the runners.
    size_t n = keep_debug ? 1 : std::min<size_t>((size_t)_runners, std::max<size_t>(1, _links.size() / 64));
    const size_t sub = 1;
    const int omp_threads = std::max(1, omp_get_max_threads() / (int)n);
    for (size_t r = 0; r < n; r++)
    {
        const size_t begin = _links.size() * r / n, end = _links.size() * (r + 1) / n;
        ochip_ctx *ctx = _ctx;
        if (r > 0 && ochip_ctx_sibling(_ctx, (uint32_t)(r - 1), &ctx) != OCHIP_OK)
        {
            error = std::string("ochip_ctx_sibling: ") + ochip_last_error(_ctx);
            ctx = nullptr;
        }
        funcs.push_back([this, &graph, begin, end, ctx, omp_threads, sub, r]() {
            if (!ctx)
                return;
            // uneven first sub-batch (shorter for later runners) staggers the runners from the start
            const size_t len = end - begin;
            std::vector<size_t> cuts{begin};
            for (size_t k = 1; k < sub; k++)
                cuts.push_back(begin + len * k / sub - (len / (sub * 4)) * (r % 3));
            cuts.push_back(end);
            for (size_t k = 0; k + 1 < cuts.size(); k++)
                if (cuts[k + 1] > cuts[k])
                    run_range(graph, cuts[k], cuts[k + 1], ctx, omp_threads);
        });
    }
    return funcs;
}

static double g_link_cpu[6]; // CPU seconds (OCHIP_VERBOSE=link): pack+upload, ratio+sort, prosac, pack jobs, decompose, assemble
static double link_thread_cpu()
{
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void link_cpu_add(int i, double v)
{
#pragma omp atomic
    g_link_cpu[i] += v;
}
static std::atomic<long> g_link_batches{0}, g_link_batches_flagged{0}; // batches run, batches a device sort sent to the host route
void link_cpu_report()
{
    fprintf(stderr, "[link] batches %ld, of which a device sort flagged (depth limit) and the host sorted: %ld\n", g_link_batches.load(),
            g_link_batches_flagged.load());
    fprintf(stderr, "[link] CPU seconds: pack %.3f, ratio+sort %.3f, prosac order %.3f, pack jobs %.3f, decompose %.3f, assemble %.3f\n",
            g_link_cpu[0], g_link_cpu[1], g_link_cpu[2], g_link_cpu[3], g_link_cpu[4], g_link_cpu[5]);
}

static thread_local bool force_host_sort = false; // (run_batch's second attempt after a device sort flagged a pair)

void LinkStage::run_batch(const MeasurementGraph &graph, const std::vector<link_pair> &link_pairs, ochip_ctx *ctx, int omp_threads)
{
    static const bool prof = ochip_verbose("link");
    LinkTimers lt; // this runner's phases, added to `timers` at the end
    struct timers_guard
    {
        LinkStage *self;
        LinkTimers *lt;
        ~timers_guard()
        {
            std::lock_guard<std::mutex> lock(self->_measurement_mutex);
            self->timers.subsample += lt->subsample;
            self->timers.upload += lt->upload;
            self->timers.match_device += lt->match_device;
            self->timers.match_host += lt->match_host;
            self->timers.ransac_device += lt->ransac_device;
            self->timers.decompose_host += lt->decompose_host;
        }
    } guard{this, &lt};
    struct pair_job
    {
        size_t loop_index, node_id, match_node_id;
        uint32_t slot_1, slot_2;
    };
    // ---- images of this batch -> device slots
    std::vector<size_t> slot_node;
    std::unordered_map<size_t, uint32_t> slot_of;
    auto slot_for = [&](size_t id) {
        auto it = slot_of.find(id);
        if (it != slot_of.end())
            return it->second;
        const uint32_t s = (uint32_t)slot_node.size();
        slot_of.emplace(id, s);
        slot_node.push_back(id);
        return s;
    };
    std::vector<pair_job> jobs;
    for (const link_pair &lp : link_pairs)
    {
        const size_t i = lp.first, match_node_id = lp.second;
        if (graph.getNode(match_node_id) == nullptr) // link_stage.cpp:69-73
            continue;
        jobs.push_back(pair_job{i, _links[i].node_id, match_node_id, slot_for(_links[i].node_id), slot_for(match_node_id)});
    }
    const size_t n_slots = slot_node.size(), n_pairs = jobs.size();
    if (n_pairs == 0)
        return;
    auto fail = [this, ctx](const char *what) {
        std::lock_guard<std::mutex> lock(_measurement_mutex);
        error = std::string(what) + ": " + ochip_last_error(ctx);
    };

    // ---- 40 px subsets and unit rays: computed once per image by prepare() (link_stage.cpp:63-65,80-81; the
    //      per-pair recomputation of the reference yields the same vector every time)
    auto t0 = clk::now();
    std::vector<const std::vector<size_t> *> subset_p(n_slots);
    std::vector<const std::vector<double> *> rays_p(n_slots);
    std::vector<const remote_subset *> remote_p(n_slots, nullptr); // images whose features live on another rank
    for (size_t s = 0; s < n_slots; s++)
    {
        const size_t k = _prepared_index.at(slot_node[s]);
        subset_p[s] = &_subsets[k];
        rays_p[s] = &_rays[k];
        if (!_remote[k].xy.empty())
        {
            remote_p[s] = &_remote[k];
            if (_remote[k].desc.size() != 8 * _subsets[k].size())
            {
                std::lock_guard<std::mutex> lock(_measurement_mutex);
                error = "link stage: an image of this batch has no descriptors on this rank";
                return;
            }
        }
    }
    struct deref_subsets
    {
        const std::vector<const std::vector<size_t> *> &p;
        const std::vector<size_t> &operator[](size_t i) const
        {
            return *p[i];
        }
    } subset{subset_p};
    struct deref_rays
    {
        const std::vector<const std::vector<double> *> &p;
        const std::vector<double> &operator[](size_t i) const
        {
            return *p[i];
        }
    } rays{rays_p};

    // ---- upload descriptors + keypoints (one packed batch, packed in parallel)
    t0 = clk::now();
    std::vector<uint32_t> counts(n_slots);
    std::vector<uint64_t> slot_off(n_slots + 1, 0);
    for (size_t s = 0; s < n_slots; s++)
    {
        counts[s] = (uint32_t)subset[s].size();
        slot_off[s + 1] = slot_off[s] + counts[s];
    }
    const uint64_t total_desc = slot_off[n_slots];
    // (the subsets' pixel locations stay at hand for assembleInliers below: 16 bytes per subset feature, one after the other,
    // instead of a walk through the images' 88-byte feature records)
    pinned<double> xybuf(ctx, total_desc * 2);
    pinned<uint32_t> idxbuf(ctx, total_desc ? total_desc : 1); // the subset features' indices in their images' feature lists
    if (!idxbuf.ptr)
        return fail("ochip_host_alloc");
    {
        pinned<uint64_t> dbuf(ctx, total_desc * 8);
        std::vector<double> models((size_t)n_slots * 8);
        if (!dbuf.ptr || !xybuf.ptr)
            return fail("ochip_host_alloc");
#pragma omp parallel for schedule(dynamic, 1) num_threads(omp_threads)
        for (size_t s = 0; s < n_slots; s++)
        {
            const double tc0 = prof ? link_thread_cpu() : 0;
            const image &img = graph.getNode(slot_node[s])->payload;
            uint64_t *d = dbuf.ptr + slot_off[s] * 8;
            double *xy = xybuf.ptr + slot_off[s] * 2;
            for (size_t k = 0; k < subset[s].size(); k++)
                idxbuf.ptr[slot_off[s] + k] = (uint32_t)subset[s][k];
            if (remote_p[s])
            {
                std::memcpy(d, remote_p[s]->desc.data(), subset[s].size() * 64);
                std::memcpy(xy, remote_p[s]->xy.data(), subset[s].size() * 16);
            }
            else
                for (size_t k = 0; k < subset[s].size(); k++)
                {
                    const feature_2d &f = img.features[subset[s][k]];
                    std::memcpy(d + 8 * k, f.descriptor, 64);
                    xy[2 * k] = f.location[0];
                    xy[2 * k + 1] = f.location[1];
                }
            const CameraModel &m = *img.model;
            const double model8[8] = {m.focal_length_pixels,      m.principle_point[0],      m.principle_point[1],
                                      m.radial_distortion[0],     m.radial_distortion[1],    m.radial_distortion[2],
                                      m.tangential_distortion[0], m.tangential_distortion[1]};
            std::memcpy(&models[s * 8], model8, sizeof model8);
            if (prof)
                link_cpu_add(0, link_thread_cpu() - tc0);
        }
        if (ochip_upload_batch(ctx, (uint32_t)n_slots, counts.data(), dbuf.ptr, xybuf.ptr, models.data()) != OCHIP_OK)
            return fail("ochip_upload_batch");
    }
    lt.upload += since(t0);

    // ---- device: Hamming 2-NN for every pair
    t0 = clk::now();
    std::vector<ochip_pair> pairs(n_pairs);
    std::vector<uint64_t> out_off(n_pairs);
    uint64_t out_total = 0;
    for (size_t p = 0; p < n_pairs; p++)
    {
        pairs[p] = ochip_pair{jobs[p].slot_1, jobs[p].slot_2};
        out_off[p] = out_total;
        out_total += subset[jobs[p].slot_1].size();
    }
    if (ochip_match_launch(ctx, pairs.data(), (uint32_t)n_pairs, out_off.data(), out_total) != OCHIP_OK)
        return fail("ochip_match_launch");
    // The tail of match_features_subset (ratio test, std::sort by distance) and the PROSAC order stay on the device
    // (csrc/match_sort.hip, csrc/std_sort.hip: libstdc++'s permutation among the equal Hamming counts): the host learns
    // the pairs' match counts here and receives the sorted correspondences with the RANSAC results.  A pair whose sort
    // would need libstdc++'s heap sort (flagged; not seen on descriptor distances) sends the batch down the host route
    // below, which is also what OCHIP_TEST_HOOKS=host_sort selects.
    const bool host_sort_env = ochip_test_hook("host_sort");
    bool device_sorted = !host_sort_env && !force_host_sort;
    std::vector<uint32_t> match_counts(n_pairs, 0);
    std::vector<uint8_t> sort_flags(n_pairs ? n_pairs : 1, 0);
    if (device_sorted)
    {
        if (ochip_match_sort(ctx, pairs.data(), (uint32_t)n_pairs, out_off.data(), out_total, match_counts.data(), sort_flags.data()) !=
            OCHIP_OK)
            return fail("ochip_match_sort");
        for (size_t p = 0; p < n_pairs; p++)
            device_sorted = device_sorted && !sort_flags[p];
        g_link_batches++;
        if (!device_sorted)
            g_link_batches_flagged++;
    }
    pinned<ochip_match> raw(ctx, !device_sorted && out_total ? out_total : 1);
    if (!raw.ptr)
        return fail("ochip_host_alloc");
    if (!device_sorted && ochip_match_fetch(ctx, raw.ptr, out_total) != OCHIP_OK)
        return fail("ochip_match_fetch");
    lt.match_device += since(t0);

    // ---- host: ratio test, std::sort, PROSAC order; pack the RANSAC jobs
    t0 = clk::now();
    struct sorted_match // 12 bytes: what the comparator and the outputs need (distance = count / 486 is monotone in count)
    {
        uint32_t k1, k2;
        uint16_t count;
    };
    std::vector<std::vector<feature_match>> matches(n_pairs);
    std::vector<std::vector<ochip_ransac_match>> rmatches(n_pairs);
    std::vector<std::vector<uint32_t>> sorted_idx(n_pairs);
#pragma omp parallel for schedule(dynamic, 1) num_threads(omp_threads)
    for (size_t p = 0; p < (device_sorted ? 0 : n_pairs); p++)
    {
        const double tc0 = prof ? link_thread_cpu() : 0;
        const auto &idx1 = subset[jobs[p].slot_1], &idx2 = subset[jobs[p].slot_2];
        const ochip_match *r = raw.ptr + out_off[p];
        // match_features_subset tail (match_features.cpp:94-101) on compact records: the permutation std::sort
        // produces depends only on the comparison results, which are those of the reference's comparator
        // (f1.distance > f2.distance) on the same sequence
        std::vector<sorted_match> sm;
        sm.reserve(idx1.size());
        if (!idx2.empty())
            for (size_t a = 0; a < idx1.size(); a++)
            {
                const double best = (size_t)r[a].best_count * (1.0 / feature_2d::DESCRIPTOR_BITS);
                const double second = r[a].second_count == OCHIP_NO_SECOND
                                          ? std::numeric_limits<double>::infinity()
                                          : (size_t)r[a].second_count * (1.0 / feature_2d::DESCRIPTOR_BITS);
                if (best < 0.8 * second)
                    sm.push_back(sorted_match{(uint32_t)a, r[a].best_k, r[a].best_count});
            }
        // (sort_like_std = libstdc++'s std::sort move for move, host/sort_like_std.hpp: Hamming counts tie all the time)
        sort_like_std(sm.data(), sm.data() + sm.size(),
                      [](const sorted_match &f1, const sorted_match &f2) -> bool { return f1.count > f2.count; });
        const size_t M = sm.size();
        matches[p].resize(M);
        rmatches[p].resize(M);
        bool has_quality = false;
        for (size_t i = 0; i < M; i++)
        {
            matches[p][i] = feature_match{idx1[sm[i].k1], idx2[sm[i].k2], (size_t)sm[i].count * (1.0 / feature_2d::DESCRIPTOR_BITS)};
            rmatches[p][i] = ochip_ransac_match{sm[i].k1, sm[i].k2, sm[i].count, 0};
            has_quality = has_quality || sm[i].count != 0;
        }
        const double tc1 = prof ? link_thread_cpu() : 0;
        // PROSAC order (ransac.cpp:83-90): iota sorted by quality ascending; empty if no quality is non-zero
        if (has_quality)
        {
            // std::sort of iota(M) by quality: the records carry the key next to the index (the permutation depends on the
            // comparator's answers only, which are those of the reference's indirect comparison)
            struct by_quality
            {
                uint32_t index;
                uint16_t count;
            };
            std::vector<by_quality> order(M);
            for (size_t i = 0; i < M; i++)
                order[i] = by_quality{(uint32_t)i, sm[i].count};
            sort_like_std(order.data(), order.data() + M, [](const by_quality &a, const by_quality &b) { return a.count < b.count; });
            sorted_idx[p].resize(M);
            for (size_t i = 0; i < M; i++)
                sorted_idx[p][i] = order[i].index;
        }
        if (prof)
        {
            const double tc2 = link_thread_cpu();
            link_cpu_add(1, tc1 - tc0);
            link_cpu_add(2, tc2 - tc1);
        }
    }
    std::vector<ochip_ransac_job> rjobs(n_pairs);
    uint64_t total_matches = 0;
    std::map<size_t, uint64_t> eval_off_of; // distinct M -> offset into the eval_order table
    std::vector<uint32_t> eval_table;
    for (size_t p = 0; p < n_pairs; p++)
    {
        const size_t M = device_sorted ? match_counts[p] : matches[p].size();
        auto it = eval_off_of.find(M);
        uint32_t rng_state = 42;
        if (it == eval_off_of.end())
        {
            const eval_order_entry &e = _eval_cache.get(M);
            it = eval_off_of.emplace(M, eval_table.size()).first;
            eval_table.insert(eval_table.end(), e.order.begin(), e.order.end());
            rng_state = e.rng_state;
        }
        else
            rng_state = _eval_cache.get(M).rng_state;
        rjobs[p] = ochip_ransac_job{jobs[p].slot_1, jobs[p].slot_2, (uint32_t)M, rng_state, total_matches, it->second};
        total_matches += M;
    }
    pinned<ochip_ransac_match> rm_flat(ctx, total_matches ? total_matches : 1);
    pinned<uint32_t> si_flat(ctx, total_matches ? total_matches : 1);
    pinned<uint8_t> inl_flat(ctx, total_matches ? total_matches : 1);
    if (!rm_flat.ptr || !si_flat.ptr || !inl_flat.ptr)
        return fail("ochip_host_alloc");
#pragma omp parallel for schedule(dynamic, 8) num_threads(omp_threads)
    for (size_t p = 0; p < (device_sorted ? 0 : n_pairs); p++)
    {
        std::copy(rmatches[p].begin(), rmatches[p].end(), rm_flat.ptr + rjobs[p].match_offset);
        if (!sorted_idx[p].empty())
            std::copy(sorted_idx[p].begin(), sorted_idx[p].end(), si_flat.ptr + rjobs[p].match_offset);
        else
            std::fill(si_flat.ptr + rjobs[p].match_offset, si_flat.ptr + rjobs[p].match_offset + rjobs[p].n, 0u);
    }
    lt.match_host += since(t0);

    // ---- device: RANSAC
    t0 = clk::now();
    std::vector<ochip_ransac_result> results(n_pairs);
    const homography_model model_defaults;
    std::vector<ochip_decomposition> decomps(device_sorted ? n_pairs : 0);
    if (device_sorted)
    {
        // (the correspondences come back in rm_flat; the PROSAC order is built and used on the device, and
        // homography_model::decompose with its cheirality vote runs there too: csrc/decompose.hpp is the host's code)
        if (ochip_ransac_homography_batch_sorted(ctx, rjobs.data(), (uint32_t)n_pairs, total_matches, eval_table.data(), eval_table.size(),
                                                 model_defaults.inlier_threshold, results.data(), inl_flat.ptr, rm_flat.ptr,
                                                 sort_flags.data(), decomps.data()) != OCHIP_OK)
            return fail("ochip_ransac_homography_batch_sorted");
        for (size_t p = 0; p < n_pairs; p++)
            if (sort_flags[p])
            {
                // the PROSAC order of a pair needs libstdc++'s heap sort: the whole batch again, sorted on the host
                g_link_batches_flagged++;
                force_host_sort = true;
                run_batch(graph, link_pairs, ctx, omp_threads);
                force_host_sort = false;
                return;
            }
    }
    // the edges' lists (matches, inlier matches) are gathers of data the device holds: written there, copied here
    std::vector<uint64_t> inlier_off(device_sorted ? n_pairs + 1 : 1, 0);
    if (device_sorted)
        for (size_t p = 0; p < n_pairs; p++)
            inlier_off[p + 1] = inlier_off[p] + decomps[p].n_inliers;
    const uint64_t total_inliers = inlier_off.back();
    pinned<feature_match> fm_flat(ctx, device_sorted && total_matches ? total_matches : 1);
    pinned<feature_match_denormalized> fmd_flat(ctx, device_sorted && total_inliers ? total_inliers : 1);
    if (!fm_flat.ptr || !fmd_flat.ptr)
        return fail("ochip_host_alloc");
    if (device_sorted)
    {
        if (ochip_edge_lists(ctx, (uint32_t)n_pairs, total_matches, idxbuf.ptr, total_desc, inlier_off.data(), total_inliers, fm_flat.ptr,
                             fmd_flat.ptr) != OCHIP_OK)
            return fail("ochip_edge_lists");
    }
    else if (ochip_ransac_homography_batch(ctx, rjobs.data(), (uint32_t)n_pairs, rm_flat.ptr, si_flat.ptr, total_matches,
                                           eval_table.data(), eval_table.size(), model_defaults.inlier_threshold,
                                           results.data(), inl_flat.ptr) != OCHIP_OK)
        return fail("ochip_ransac_homography_batch");
    lt.ransac_device += since(t0);

    // ---- host: decompose, accept, assemble (link_stage.cpp:95-111)
    t0 = clk::now();
    std::vector<edge_payload> payloads(n_pairs);
    std::vector<pair_debug> dbg(keep_debug ? n_pairs : 0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(omp_threads)
    for (size_t p = 0; p < n_pairs; p++)
    {
        const double tc0 = prof ? link_thread_cpu() : 0;
        const size_t M = rjobs[p].n;
        const uint8_t *inl = inl_flat.ptr + rjobs[p].match_offset;
        const ochip_ransac_match *rm = rm_flat.ptr + rjobs[p].match_offset;
        if (device_sorted) // match_features_subset's output (match_features.cpp:86-101), as the device listed it
            matches[p].assign(fm_flat.ptr + rjobs[p].match_offset, fm_flat.ptr + rjobs[p].match_offset + M);
        camera_relations relations;
        homography_model h;
        std::memcpy(h.homography, results[p].H, sizeof h.homography);
        std::memcpy(relations.ransac_relation, results[p].H, sizeof relations.ransac_relation);
        relations.relationType = camera_relations::RelationType::HOMOGRAPHY;

        // the cheirality vote reads the inliers' rays where they lie (two gathers per inlier, all solutions in one pass)
        static thread_local std::vector<uint32_t> inlier_at;
        inlier_at.clear();
        if (!device_sorted)
            for (size_t i = 0; i < M; i++)
                if (inl[i])
                    inlier_at.push_back((uint32_t)i);
        const size_t num_coarse_inliers = device_sorted ? decomps[p].n_inliers : inlier_at.size();
        const auto &ray1 = rays[jobs[p].slot_1], &ray2 = rays[jobs[p].slot_2];
        const uint32_t *at = inlier_at.data();
        bool can_decompose;
        if (device_sorted)
        {
            const ochip_decomposition &d = decomps[p];
            for (int k = 0; k < 4; k++)
            {
                std::memcpy(relations.relative_poses[k].orientation, &d.pose[k][0], 4 * sizeof(double));
                std::memcpy(relations.relative_poses[k].position, &d.pose[k][4], 3 * sizeof(double));
                relations.relative_poses[k].score = (int)d.pose[k][7];
            }
            can_decompose = d.can_decompose != 0;
        }
        else
            can_decompose = h.decompose_with(
                num_coarse_inliers,
                [&](size_t j, const double *&m1, const double *&m2) {
                    m1 = &ray1[3 * (size_t)rm[at[j]].k1];
                    m2 = &ray2[3 * (size_t)rm[at[j]].k2];
                },
                relations.relative_poses);
        if (keep_debug)
        {
            dbg[p].node_id = jobs[p].node_id;
            dbg[p].match_node_id = jobs[p].match_node_id;
            dbg[p].matches = matches[p];
            dbg[p].inliers.assign(inl, inl + M);
            dbg[p].score = results[p].score;
            dbg[p].iterations = results[p].iterations;
            dbg[p].improvements = results[p].improvements;
            dbg[p].can_decompose = can_decompose;
        }
        const double tc1 = prof ? link_thread_cpu() : 0;
        if (can_decompose && num_coarse_inliers > h.MINIMUM_POINTS * 1.5)
        {
            relations.matches = std::move(matches[p]);
            if (device_sorted) // assembleInliers (ransac.cpp:263-282), as the device listed it
                relations.inlier_matches.assign(fmd_flat.ptr + inlier_off[p], fmd_flat.ptr + inlier_off[p + 1]);
            else
            {
                // assembleInliers (ransac.cpp:263-282) with the pixels read from the subsets' packed copy: the same numbers
                // (and an image whose feature list lives on another rank has nothing else)
                auto px_of = [&](uint32_t slot, uint32_t k) -> const double * { return xybuf.ptr + 2 * (slot_off[slot] + k); };
                relations.inlier_matches.reserve(num_coarse_inliers);
                for (size_t j = 0; j < num_coarse_inliers; j++)
                {
                    const size_t i = at[j];
                    const double *p1 = px_of(jobs[p].slot_1, rm[i].k1), *p2 = px_of(jobs[p].slot_2, rm[i].k2);
                    feature_match_denormalized fmd;
                    fmd.pixel_1[0] = p1[0], fmd.pixel_1[1] = p1[1];
                    fmd.pixel_2[0] = p2[0], fmd.pixel_2[1] = p2[1];
                    fmd.feature_index_1 = relations.matches[i].feature_index_1;
                    fmd.feature_index_2 = relations.matches[i].feature_index_2;
                    fmd.match_index = i;
                    relations.inlier_matches.push_back(fmd);
                }
            }
        }
        payloads[p] = edge_payload{jobs[p].loop_index, jobs[p].node_id, jobs[p].match_node_id, std::move(relations), {}};
        if (_sharded && !payloads[p].relations.matches.empty())
        {
            payloads[p].subset_pos.resize(2 * M);
            for (size_t i = 0; i < M; i++)
            {
                payloads[p].subset_pos[2 * i] = rm[i].k1;
                payloads[p].subset_pos[2 * i + 1] = rm[i].k2;
            }
        }
        if (prof)
        {
            const double tc2 = link_thread_cpu();
            link_cpu_add(4, tc1 - tc0);
            link_cpu_add(5, tc2 - tc1);
        }
    }
    {
        std::lock_guard<std::mutex> lock(_measurement_mutex);
        for (auto &pl : payloads)
            _all_inlier_measurements.emplace_back(std::move(pl));
        for (auto &d : dbg)
            debug.emplace_back(std::move(d));
    }
    lt.decompose_host += since(t0);
}

// ---- one survey over several ranks -----------------------------------------------------------------------------------
namespace
{
template <typename T> void put(std::vector<uint8_t> &out, const T *v, size_t n)
{
    const uint8_t *b = reinterpret_cast<const uint8_t *>(v);
    out.insert(out.end(), b, b + n * sizeof(T));
}
template <typename T> void put1(std::vector<uint8_t> &out, T v)
{
    put(out, &v, 1);
}
struct reader
{
    const uint8_t *p, *end;
    bool ok = true;
    template <typename T> const T *take(size_t n) // unaligned reads are done by memcpy from the returned pointer
    {
        if (!ok || (size_t)(end - p) < n * sizeof(T))
        {
            ok = false;
            return nullptr;
        }
        const T *r = reinterpret_cast<const T *>(p);
        p += n * sizeof(T);
        return r;
    }
    template <typename T> T one()
    {
        T v{};
        const T *r = take<T>(1);
        if (r)
            std::memcpy(&v, r, sizeof(T));
        return v;
    }
};
struct edge_header // 8-byte aligned: the records of one buffer follow each other padded to 8 bytes
{
    uint32_t loop_index, image_1, image_2, n_matches, n_inliers, flags; // flags: bit 0 = HOMOGRAPHY, bit 1 = 32-bit positions
    double H[9];
    double poses[4][7];
    int32_t score[4];
};
} // namespace

void LinkStage::set_remote_subset(const MeasurementGraph &graph, size_t node_id, size_t n, const uint32_t *idx, const double *xy,
                                  const uint64_t *desc)
{
    const auto it = _prepared_index.find(node_id);
    if (it == _prepared_index.end())
        return; // no link touches the image
    const size_t s = it->second;
    _subsets[s].resize(n);
    for (size_t q = 0; q < n; q++)
        _subsets[s][q] = idx[q];
    _remote[s].xy.assign(xy, xy + 2 * n);
    if (desc)
    {
        // the image takes part in this rank's pairs: descriptors for the matcher, unit rays for the decomposition's vote
        _remote[s].desc.assign(desc, desc + 8 * n);
        const image &img = graph.getNode(node_id)->payload;
        _rays[s].resize(3 * n);
        for (size_t q = 0; q < n; q++)
            image_to_3d(xy + 2 * q, *img.model, &_rays[s][3 * q]);
    }
}

void LinkStage::export_subsets(const MeasurementGraph &graph, const std::vector<size_t> &node_ids, size_t first, size_t count,
                               std::vector<uint8_t> &out) const
{
    for (size_t b = first; b < first + count; b++)
    {
        const auto it = _prepared_index.find(node_ids[b]);
        const std::vector<size_t> empty;
        const std::vector<size_t> &sub = it == _prepared_index.end() ? empty : _subsets[it->second];
        const image &img = graph.getNode(node_ids[b])->payload;
        put1<uint32_t>(out, (uint32_t)b);
        put1<uint32_t>(out, (uint32_t)sub.size());
        const size_t at = out.size(), n = sub.size();
        out.resize(at + n * (4 + 16 + 64) + (n & 1) * 4); // idx | pad to 8 | xy | desc
        uint8_t *o = out.data() + at;
        for (size_t q = 0; q < n; q++)
        {
            const uint32_t v = (uint32_t)sub[q];
            std::memcpy(o + 4 * q, &v, 4);
        }
        o += 4 * (n + (n & 1));
        for (size_t q = 0; q < n; q++)
            std::memcpy(o + 16 * q, img.features[sub[q]].location, 16);
        o += 16 * n;
        for (size_t q = 0; q < n; q++)
            std::memcpy(o + 64 * q, img.features[sub[q]].descriptor, 64);
    }
}

void LinkStage::export_edges(const std::unordered_map<size_t, uint32_t> &image_of, std::vector<uint8_t> &out) const
{
    for (const edge_payload &pl : _all_inlier_measurements)
    {
        const camera_relations &r = pl.relations;
        const size_t M = r.matches.size(), I = r.inlier_matches.size();
        bool wide = false;
        for (uint32_t v : pl.subset_pos)
            wide = wide || v > 0xFFFFu;
        edge_header h{};
        h.loop_index = (uint32_t)pl.loop_index;
        h.image_1 = image_of.at(pl.node_id);
        h.image_2 = image_of.at(pl.match_node_id);
        h.n_matches = (uint32_t)M;
        h.n_inliers = (uint32_t)I;
        h.flags = (r.relationType == camera_relations::RelationType::HOMOGRAPHY ? 1u : 0u) | (wide ? 2u : 0u);
        std::memcpy(h.H, r.ransac_relation, sizeof h.H);
        for (int i = 0; i < 4; i++)
        {
            std::memcpy(h.poses[i], r.relative_poses[i].orientation, 32);
            std::memcpy(h.poses[i] + 4, r.relative_poses[i].position, 24);
            h.score[i] = r.relative_poses[i].score;
        }
        put1(out, h);
        // per match: the two subset positions and the Hamming count (distance = count * (1.0 / 486), match_features.cpp:79)
        for (size_t i = 0; i < M; i++)
        {
            const uint16_t count = (uint16_t)std::lround(r.matches[i].distance * feature_2d::DESCRIPTOR_BITS);
            if (wide)
                put(out, &pl.subset_pos[2 * i], 2);
            else
            {
                const uint16_t k[2] = {(uint16_t)pl.subset_pos[2 * i], (uint16_t)pl.subset_pos[2 * i + 1]};
                put(out, k, 2);
            }
            put1(out, count);
        }
        for (size_t j = 0; j < I; j++)
            put1<uint32_t>(out, (uint32_t)r.inlier_matches[j].match_index);
        out.resize((out.size() + 7) & ~(size_t)7);
    }
}

bool LinkStage::import_edges(const MeasurementGraph &graph, const std::vector<size_t> &node_ids, const uint8_t *buf, size_t bytes)
{
    // first pass: where every record starts (they are variable-sized), then the records are rebuilt in parallel
    struct rec
    {
        const uint8_t *at;
        edge_header h;
    };
    std::vector<rec> recs;
    reader rd{buf, buf + bytes};
    while (rd.ok && rd.p < rd.end)
    {
        rec r;
        r.at = rd.p;
        r.h = rd.one<edge_header>();
        if (!rd.ok || r.h.image_1 >= node_ids.size() || r.h.image_2 >= node_ids.size() || r.h.n_inliers > r.h.n_matches)
        {
            rd.ok = false;
            break;
        }
        const size_t body = (size_t)r.h.n_matches * ((r.h.flags & 2u) ? 10 : 6) + (size_t)r.h.n_inliers * 4;
        rd.take<uint8_t>((body + 7) & ~(size_t)7);
        if (rd.ok)
            recs.push_back(r);
    }
    if (!rd.ok)
    {
        error = "link stage: malformed edge buffer";
        return false;
    }
    std::vector<edge_payload> payloads(recs.size());
    bool bad = false;
#pragma omp parallel for schedule(dynamic, 16) reduction(|| : bad)
    for (size_t e = 0; e < recs.size(); e++)
    {
        const edge_header &h = recs[e].h;
        const size_t id1 = node_ids[h.image_1], id2 = node_ids[h.image_2];
        edge_payload &pl = payloads[e];
        pl.loop_index = h.loop_index;
        pl.node_id = id1;
        pl.match_node_id = id2;
        camera_relations &r = pl.relations;
        r.relationType = (h.flags & 1u) ? camera_relations::RelationType::HOMOGRAPHY : camera_relations::RelationType::UNKNOWN;
        std::memcpy(r.ransac_relation, h.H, sizeof h.H);
        for (int i = 0; i < 4; i++)
        {
            std::memcpy(r.relative_poses[i].orientation, h.poses[i], 32);
            std::memcpy(r.relative_poses[i].position, h.poses[i] + 4, 24);
            r.relative_poses[i].score = h.score[i];
        }
        if (h.n_matches == 0)
            continue;
        const auto s1 = _prepared_index.find(id1), s2 = _prepared_index.find(id2);
        if (s1 == _prepared_index.end() || s2 == _prepared_index.end())
        {
            bad = true;
            continue;
        }
        const std::vector<size_t> &sub1 = _subsets[s1->second], &sub2 = _subsets[s2->second];
        auto px_of = [&](size_t id, size_t slot, uint32_t k) -> const double * {
            return !_remote[slot].xy.empty() ? &_remote[slot].xy[2 * (size_t)k]
                                             : graph.getNode(id)->payload.features[_subsets[slot][k]].location;
        };
        const bool wide = (h.flags & 2u) != 0;
        const uint8_t *m = recs[e].at + sizeof(edge_header);
        r.matches.resize(h.n_matches);
        std::vector<uint32_t> pos(2 * (size_t)h.n_matches);
        for (size_t i = 0; i < h.n_matches; i++)
        {
            uint32_t k1, k2;
            uint16_t count;
            if (wide)
            {
                std::memcpy(&k1, m + 10 * i, 4);
                std::memcpy(&k2, m + 10 * i + 4, 4);
                std::memcpy(&count, m + 10 * i + 8, 2);
            }
            else
            {
                uint16_t k[3];
                std::memcpy(k, m + 6 * i, 6);
                k1 = k[0], k2 = k[1], count = k[2];
            }
            if (k1 >= sub1.size() || k2 >= sub2.size())
            {
                bad = true;
                k1 = k2 = 0;
                if (sub1.empty() || sub2.empty())
                    break;
            }
            pos[2 * i] = k1, pos[2 * i + 1] = k2;
            r.matches[i] = feature_match{sub1[k1], sub2[k2], (size_t)count * (1.0 / feature_2d::DESCRIPTOR_BITS)};
        }
        if (sub1.empty() || sub2.empty())
            continue;
        const uint8_t *in = m + (size_t)h.n_matches * (wide ? 10 : 6);
        r.inlier_matches.resize(h.n_inliers);
        for (size_t j = 0; j < h.n_inliers; j++)
        {
            uint32_t mi;
            std::memcpy(&mi, in + 4 * j, 4);
            if (mi >= h.n_matches)
            {
                bad = true;
                mi = 0;
            }
            const double *p1 = px_of(id1, s1->second, pos[2 * mi]), *p2 = px_of(id2, s2->second, pos[2 * mi + 1]);
            feature_match_denormalized &f = r.inlier_matches[j];
            f.pixel_1[0] = p1[0], f.pixel_1[1] = p1[1];
            f.pixel_2[0] = p2[0], f.pixel_2[1] = p2[1];
            f.feature_index_1 = r.matches[mi].feature_index_1;
            f.feature_index_2 = r.matches[mi].feature_index_2;
            f.match_index = mi;
        }
    }
    if (bad)
    {
        error = "link stage: an imported edge refers to a subset this rank does not hold";
        return false;
    }
    std::lock_guard<std::mutex> lock(_measurement_mutex);
    for (auto &pl : payloads)
        _all_inlier_measurements.emplace_back(std::move(pl));
    return true;
}

std::vector<size_t> LinkStage::finalize(MeasurementGraph &graph)
{
    const auto t0 = clk::now();
    std::sort(_all_inlier_measurements.begin(), _all_inlier_measurements.end(), [](const auto &a, const auto &b) {
        return std::make_tuple(a.loop_index, a.node_id, a.match_node_id) <
               std::make_tuple(b.loop_index, b.node_id, b.match_node_id);
    });
    for (auto &measurements : _all_inlier_measurements)
        graph.addEdge(std::move(measurements.relations), measurements.node_id, measurements.match_node_id);
    _all_inlier_measurements.clear();

    std::vector<size_t> node_ids;
    node_ids.reserve(_links.size());
    for (const auto &link : _links)
        node_ids.push_back(link.node_id);
    _links.clear();
    timers.link_finalize += since(t0);
    if (ochip_verbose("link"))
        link_cpu_report();
    return node_ids;
}

} // namespace opencalibration_amd
