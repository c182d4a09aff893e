// Load and link stages overlapped, as the reference's pipeline overlaps the stages of consecutive image batches
// (Pipeline::Impl::initial_processing runs the load, link and relax closures of different batches side by side,
// src/pipeline/pipeline.cpp:522-570).  The images of one survey are extracted chunk by chunk on the device; as soon
// as every image a range of links touches (the source images and their kNN neighbours) has its features, that range
// is handed to a link runner on its own device context.  The link kernels then share the device with the extraction
// of later chunks, and the host phases of linking (ratio test + std::sort, decompose, assembleInliers) hide under
// device time that is needed anyway.  The resulting graph is the one the two stages produce one after the other:
// nodes in image order, edges in LinkStage::finalize's deterministic order.
#include "../../../include/oc_host.h"

#include "capi_graph.hpp"
#include "extract_features.hpp"
#include "load_link.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>

#include <omp.h>

using namespace opencalibration_amd;

namespace opencalibration_amd
{

std::vector<owned_pair> pair_owners(const std::vector<NodeLinks> &links)
{
    // A directed pair belongs to its source image - unless its reverse is a pair too: then both go to the later of the
    // two sources.  Either way a pair can only run once both images are extracted, so nothing starts later than it
    // could; but the two directions now always sit in the same batch (and, with one survey over several ranks, on the
    // same rank), where the device matches them from ONE pass over their distance matrix (hamming_2nn_sym_kernel).
    // With batches cut by source alone every pair that straddled a cut (13 % of them on the C3 grid) had its distances
    // computed twice, by the slower one-direction kernel.  Which batch ran a pair does not show in the graph:
    // finalize() orders the edges.
    std::vector<owned_pair> out;
    std::unordered_map<size_t, size_t> link_of; // node id -> index into links
    for (size_t i = 0; i < links.size(); i++)
        link_of.emplace(links[i].node_id, i);
    for (size_t i = 0; i < links.size(); i++)
        for (size_t m : links[i].link_ids)
        {
            size_t owner = i;
            auto it = link_of.find(m);
            if (it != link_of.end() && it->second > i)
            {
                const auto &back = links[it->second].link_ids;
                if (std::find(back.begin(), back.end(), links[i].node_id) != back.end())
                    owner = it->second;
            }
            out.push_back(owned_pair{owner, LinkStage::link_pair(i, m)});
        }
    return out;
}

// Surveys in flight.  Two calls may overlap (two host threads, a graph each, the same device context): the second one's
// extraction starts when the first one's has finished, and runs beside the first one's remaining link ranges - the tail of
// a survey's link stage (its last ranges can only start once its last images are extracted, and RANSAC is a chain of
// dependent steps per pair that leaves most of the device idle) then no longer stands alone on the device.  This is the
// reference's own schedule: its pipeline runs the load runners of one batch beside the link runners of the batch before
// (Pipeline::Impl::initial_processing, src/pipeline/pipeline.cpp:543-560).  The extraction contexts (ctx and its first
// siblings) are used by one survey at a time - the gate below -, the link runners of alternating calls use two different
// sets of sibling contexts.
namespace
{
std::mutex g_extract_gate_mu;
std::condition_variable g_extract_gate_cv;
bool g_extract_gate_busy = false;
thread_local double g_last_gate_wait = 0; // seconds the calling thread's last load_link_stream waited at the gate
// The two sets of link-runner contexts (siblings 4.. and 13.. of ctx): a call OWNS one from its first runner to its last
// (a free-list, not the parity of a call counter: calls need not finish in the order they started, and a third overlapping
// call waits here instead of driving a context set that is still in use).
std::mutex g_lane_mu;
std::condition_variable g_lane_cv;
bool g_lane_busy[2] = {false, false};
struct runner_lane
{
    int lane = -1;
    void acquire()
    {
        std::unique_lock<std::mutex> lk(g_lane_mu);
        g_lane_cv.wait(lk, [] { return !g_lane_busy[0] || !g_lane_busy[1]; });
        lane = g_lane_busy[0] ? 1 : 0;
        g_lane_busy[lane] = true;
    }
    ~runner_lane()
    {
        if (lane < 0)
            return;
        {
            std::lock_guard<std::mutex> lk(g_lane_mu);
            g_lane_busy[lane] = false;
        }
        g_lane_cv.notify_all();
    }
};
struct extract_gate
{
    bool held = false;
    void acquire()
    {
        std::unique_lock<std::mutex> lk(g_extract_gate_mu);
        g_extract_gate_cv.wait(lk, [] { return !g_extract_gate_busy; });
        g_extract_gate_busy = true;
        held = true;
    }
    void release()
    {
        if (!held)
            return;
        {
            std::lock_guard<std::mutex> lk(g_extract_gate_mu);
            g_extract_gate_busy = false;
        }
        g_extract_gate_cv.notify_all();
        held = false;
    }
    ~extract_gate() { release(); }
};
} // namespace

bool load_link_stream(och_graph *g, ochip_ctx *ctx, LinkStage &link, const std::vector<size_t> &ids, uint32_t first,
                      uint32_t count, const uint8_t *images_bgr, int width, int height, uint32_t max_keypoints,
                      bool images_on_device, const std::vector<owned_pair> &pairs, double *total_out, double *sparse_out,
                      double *t_extract_done)
{
    using clk = std::chrono::steady_clock;
    // (declared before the gate: destroyed after it - and after every runner thread below has been joined)
    runner_lane lane_guard;
    lane_guard.acquire();
    const int lane = lane_guard.lane;
    extract_gate gate;
    const auto t_gate = clk::now();
    // (released when this survey's last chunk is extracted; on every return path by the destructor.  Measured in round 6 for
    // views that come from HOST memory - bound by the PCIe link, not the device - without the gate, two surveys extracting side
    // by side: 1 271 / 1 336 images/s against 1 306 / 1 275 with it, nothing in it; OCHIP_EXTRACT_GATE=0 turns it off)
    const char *gate_env = std::getenv("OCHIP_EXTRACT_GATE");
    if (!(gate_env && gate_env[0] == '0'))
        gate.acquire();
    const auto t_begin = clk::now();
    g_last_gate_wait = std::chrono::duration<double>(t_begin - t_gate).count();
    const auto &links = link.links();
    // ---- ranges of links and the images (of this call's block) each one waits for
    std::unordered_map<size_t, uint32_t> image_of; // node id -> image index within the block
    for (uint32_t b = 0; b < count; b++)
        image_of.emplace(ids[first + b], b);
    constexpr size_t range_len = 125;
    struct range
    {
        std::vector<LinkStage::link_pair> pairs;
        uint32_t waiting; // images not ready yet
    };
    size_t owner_lo = links.size(), owner_hi = 0;
    for (const owned_pair &op : pairs)
    {
        owner_lo = std::min(owner_lo, op.owner);
        owner_hi = std::max(owner_hi, op.owner + 1);
    }
    const size_t n_ranges = pairs.empty() ? 0 : (owner_hi - owner_lo + range_len - 1) / range_len;
    std::vector<range> ranges(n_ranges);
    for (const owned_pair &op : pairs)
        ranges[(op.owner - owner_lo) / range_len].pairs.push_back(op.pair);
    std::vector<std::vector<uint32_t>> ranges_of_image(count); // image -> ranges that need it
    std::vector<uint32_t> ready_at_once;
    for (size_t k = 0; k < n_ranges; k++)
    {
        range &r = ranges[k];
        std::vector<uint32_t> need;
        for (const auto &lp : r.pairs)
        {
            auto a = image_of.find(links[lp.first].node_id);
            if (a != image_of.end())
                need.push_back(a->second);
            auto it = image_of.find(lp.second);
            if (it != image_of.end())
                need.push_back(it->second);
        }
        std::sort(need.begin(), need.end());
        need.erase(std::unique(need.begin(), need.end()), need.end());
        r.waiting = (uint32_t)need.size();
        for (uint32_t im : need)
            ranges_of_image[im].push_back((uint32_t)k);
        if (need.empty() && !r.pairs.empty())
            ready_at_once.push_back((uint32_t)k);
    }

    // ---- link runners: each owns a device context (siblings 4.. of ctx; extraction uses ctx and its first siblings)
    const char *renv = std::getenv("OCHIP_LINK_RUNNERS");
    const int n_runners = std::max(1, std::min(8, renv ? std::atoi(renv) : 3));
    const int team = omp_get_max_threads();
    const int tail_threads = std::max(1, team / 2), runner_threads = std::max(1, team / (2 * n_runners) + 1);
    std::mutex mu;
    std::condition_variable cv;
    std::deque<uint32_t> ready(ready_at_once.begin(), ready_at_once.end());
    bool no_more = false;
    std::vector<std::thread> runners;
    for (int r = 0; r < n_runners; r++)
    {
        ochip_ctx *rctx = nullptr;
        if (ochip_ctx_sibling(ctx, (uint32_t)((lane ? 13 : 4) + r), &rctx) != OCHIP_OK) // (sibling 12: the bench's relax context)
        {
            g->error = std::string("ochip_ctx_sibling: ") + ochip_last_error(ctx);
            {
                std::lock_guard<std::mutex> lk(mu);
                no_more = true;
                ready.clear();
            }
            cv.notify_all();
            for (auto &t : runners)
                t.join();
            return false;
        }
        runners.emplace_back([&, rctx]() {
            for (;;)
            {
                uint32_t k;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !ready.empty() || no_more; });
                    if (ready.empty())
                        return;
                    k = ready.front();
                    ready.pop_front();
                }
                link.run_pairs(g->graph, ranges[k].pairs, rctx, runner_threads);
            }
        });
    }

    // ---- extraction; every finished chunk fills its nodes, prepares their 40 px subsets and releases the ranges
    //      that were only waiting for these images
    double total = 0, sparse = 0;
    const bool ok = extract_features_stream(
        ctx, images_bgr, count, width, height, max_keypoints, images_on_device, tail_threads,
        [&](uint32_t chunk_first, uint32_t chunk_count, extracted_features *f) {
            std::vector<size_t> chunk_ids(chunk_count);
            for (uint32_t i = 0; i < chunk_count; i++)
            {
                image &img = g->graph.getNode(ids[first + chunk_first + i])->payload;
                total += (double)f[i].features.size();
                sparse += (double)f[i].num_sparse_features;
                img.features = std::move(f[i].features);
                img.num_sparse_features = f[i].num_sparse_features;
                img.coarse_subset = std::move(f[i].coarse_subset);
                img.coarse_spacing = f[i].coarse_spacing;
                chunk_ids[i] = ids[first + chunk_first + i];
            }
            link.prepare_images(g->graph, chunk_ids, tail_threads);
            std::lock_guard<std::mutex> lk(mu);
            for (uint32_t i = 0; i < chunk_count; i++)
                for (uint32_t k : ranges_of_image[chunk_first + i])
                    if (--ranges[k].waiting == 0)
                        ready.push_back(k);
            cv.notify_all();
        },
        &g->error);
    gate.release(); // the next survey may extract while this one's remaining ranges are linked
    if (t_extract_done)
        *t_extract_done = std::chrono::duration<double>(clk::now() - t_begin).count();
    {
        std::lock_guard<std::mutex> lk(mu);
        no_more = true; // runners drain what is queued, then stop
        if (!ok)
            ready.clear();
    }
    cv.notify_all();
    for (auto &t : runners)
        t.join();
    if (!ok)
        return false;
    if (!link.error.empty())
    {
        g->error = link.error;
        return false;
    }
    if (total_out)
        *total_out = total;
    if (sparse_out)
        *sparse_out = sparse;
    return true;
}

std::vector<size_t> add_survey_nodes(och_graph *g, uint32_t n_images, uint32_t model, const double *positions,
                                     const double *orientations, uint64_t *node_ids_out)
{
    // nodes first (positions and camera model are known before any pixel is touched): the link stage's kNN only needs
    // those, and the node ids come out in image order as with the one-after-the-other path
    std::vector<size_t> ids(n_images);
    for (uint32_t b = 0; b < n_images; b++)
    {
        image img;
        img.model = g->models[model];
        for (int i = 0; i < 3; i++)
            img.position[i] = positions[3 * (size_t)b + i];
        if (orientations)
            for (int i = 0; i < 4; i++)
                img.orientation[i] = orientations[4 * (size_t)b + i];
        img.path = "image_" + std::to_string(g->graph.size_nodes());
        ids[b] = g->graph.addNode(std::move(img));
        if (node_ids_out)
            node_ids_out[b] = ids[b];
    }
    return ids;
}

} // namespace opencalibration_amd

extern "C" int och_graph_load_link_images(och_graph *g, ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images, int width,
                                          int height, uint32_t max_keypoints, int images_on_device, uint32_t model,
                                          const double *positions, const double *orientations, uint64_t *node_ids_out,
                                          double *totals2, double *link_timers8, double *stage_seconds2)
{
    using clk = std::chrono::steady_clock;
    auto seconds_since = [](clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); };
    if (!g || !ctx || (n_images && (!images_bgr || !positions)) || model >= g->models.size())
    {
        if (g)
            g->error = "och_graph_load_link_images: bad argument";
        return -1;
    }
    const auto t_begin = clk::now();
    const std::vector<size_t> ids = add_survey_nodes(g, n_images, model, positions, orientations, node_ids_out);
    g->link = std::make_unique<LinkStage>(ctx);
    LinkStage &link = *g->link;
    link.init(g->graph, ids);
    link.prepare_index(g->graph);
    {
        // images of earlier calls are link partners of the new ones (the reference links a batch against everything
        // already in the graph, link_stage.cpp:26-35): their 40 px subsets and rays are prepared here, no range waits
        // for them
        std::unordered_map<size_t, char> mine;
        for (size_t id : ids)
            mine.emplace(id, 1);
        std::vector<size_t> earlier;
        for (const auto &n : g->graph.nodes())
            if (!mine.count(n.id) && !n.payload.features.empty())
                earlier.push_back(n.id);
        if (!earlier.empty())
            link.prepare_images(g->graph, earlier, omp_get_max_threads());
    }
    double total = 0, sparse = 0, t_extract_done = 0;
    if (!load_link_stream(g, ctx, link, ids, 0, n_images, images_bgr, width, height, max_keypoints, images_on_device != 0,
                          pair_owners(link.links()), &total, &sparse, &t_extract_done))
        return -1;
    const double gate_wait = g_last_gate_wait;
    link.finalize(g->graph);
    if (totals2)
    {
        totals2[0] = total;
        totals2[1] = sparse;
    }
    if (link_timers8)
    {
        const LinkTimers &t = link.timers;
        const double v[8] = {t.link_init,  t.subsample,     t.upload,         t.match_device,
                             t.match_host, t.ransac_device, t.decompose_host, t.link_finalize};
        for (int i = 0; i < 8; i++)
            link_timers8[i] = v[i];
    }
    if (stage_seconds2)
    {
        stage_seconds2[0] = t_extract_done;            // until the last chunk's features were final
        stage_seconds2[1] = seconds_since(t_begin) - gate_wait; // until the graph was linked (not counting the wait for the survey before to finish extracting)
    }
    return 0;
}
