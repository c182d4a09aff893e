// Pipeline::Impl::initial_processing (src/pipeline/pipeline.cpp:522-570) with the reference's own software pipelining: one
// step loads batch k (extract), links batch k - 1 against everything loaded before it and relaxes batch k - 2 as ONE group
// with two rings of context cameras ({ORIENTATION, GROUND_PLANE}, relax_all = false, disable_parallelism = true, :545-546) -
// the three stages' runners side by side (run_parallel, :42-49, :556) - and finalizes them in the reference's order
// (load: the batch's nodes enter the graph; link: the edges; relax: the orientations, :558-560).  So a link stage sees the
// nodes of the batches before its own, a relax stage sees the nodes up to one batch behind its own and the edges up to its
// own - exactly the reference's states - and nothing a runner reads is written while it runs.
//
// On the device the three stages share the GPU: the extraction's launch sequences on the context and its first siblings, the
// link runners on their own sibling contexts (host/load_link.cpp's convention: siblings 4..), the relax on sibling 12 - for
// the cameras that arrive without an orientation one resident launch (csrc/relax_chain.hip).
#include "../../../include/oc_host.h"

#include "capi_graph.hpp"
#include "extract_features.hpp"
#include "load_link.hpp"
#include "relax_stage.hpp"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include <omp.h>

using namespace opencalibration_amd;

struct och_initial_processing
{
    och_graph *g = nullptr;
    ochip_ctx *ctx = nullptr;
    std::vector<size_t> next_loaded_ids, next_linked_ids, next_relaxed_ids; // pipeline.cpp:94-96
    uint64_t steps = 0;
    std::vector<surface_model> surfaces; // (:562-565)
};

namespace
{
using clk = std::chrono::steady_clock;
double since(clk::time_point t0)
{
    return std::chrono::duration<double>(clk::now() - t0).count();
}

bool stage_priority()
{
    const char *e = std::getenv("OCHIP_IP_PRIORITY");
    return !(e && e[0] == '0');
}

// the link stage's runners over `link`'s pairs: ranges of 125 links, each on a runner thread with its own device context
bool run_link_runners(och_graph *g, ochip_ctx *ctx, LinkStage &link)
{
    const std::vector<owned_pair> pairs = pair_owners(link.links());
    if (pairs.empty())
        return true;
    constexpr size_t range_len = 125;
    size_t owner_lo = link.links().size(), owner_hi = 0;
    for (const owned_pair &op : pairs)
    {
        owner_lo = std::min(owner_lo, op.owner);
        owner_hi = std::max(owner_hi, op.owner + 1);
    }
    std::vector<std::vector<LinkStage::link_pair>> ranges((owner_hi - owner_lo + range_len - 1) / range_len);
    for (const owned_pair &op : pairs)
        ranges[(op.owner - owner_lo) / range_len].push_back(op.pair);
    const char *renv = std::getenv("OCHIP_LINK_RUNNERS");
    const int n_runners = std::max(1, std::min(8, renv ? std::atoi(renv) : 3));
    const int team = omp_get_max_threads();
    const int runner_threads = std::max(1, team / (2 * n_runners) + 1);
    std::mutex mu;
    size_t next = 0;
    std::vector<std::thread> runners;
    std::string sibling_error;
    for (int r = 0; r < n_runners && r < (int)ranges.size(); r++)
    {
        ochip_ctx *rctx = nullptr;
        if (ochip_ctx_sibling(ctx, (uint32_t)(4 + r), &rctx) != OCHIP_OK)
        {
            sibling_error = std::string("ochip_ctx_sibling: ") + ochip_last_error(ctx);
            break;
        }
        // a batch's link stage is a chain of short launches (900 pairs do not fill the device; RANSAC is as long as its
        // slowest pair) beside the extraction's launches that do: its streams go ahead of them (OCHIP_IP_PRIORITY=0: not)
        if (stage_priority())
            (void)ochip_ctx_set_priority(rctx, 1);
        runners.emplace_back([&, rctx]() {
            for (;;)
            {
                size_t k;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (next >= ranges.size())
                        return;
                    k = next++;
                }
                if (!ranges[k].empty())
                    link.run_pairs(g->graph, ranges[k], rctx, runner_threads);
            }
        });
    }
    for (auto &t : runners)
        t.join();
    if (!sibling_error.empty())
    {
        g->error = sibling_error;
        return false;
    }
    if (!link.error.empty())
    {
        g->error = link.error;
        return false;
    }
    return true;
}
} // namespace

extern "C"
{

och_initial_processing *och_initial_processing_create(och_graph *g, ochip_ctx *ctx)
{
    if (!g || !ctx)
        return nullptr;
    auto *ip = new (std::nothrow) och_initial_processing();
    if (!ip)
        return nullptr;
    ip->g = g;
    ip->ctx = ctx;
    return ip;
}

void och_initial_processing_destroy(och_initial_processing *ip)
{
    delete ip;
}

int och_initial_processing_pending(const och_initial_processing *ip)
{
    return ip && (!ip->next_loaded_ids.empty() || !ip->next_linked_ids.empty()) ? 1 : 0;
}

int och_initial_processing_step(och_initial_processing *ip, const uint8_t *images_bgr, uint32_t n_images, int width, int height,
                                uint32_t max_keypoints, int images_on_device, uint32_t model, const double *positions,
                                int sequential, uint64_t *node_ids_out, double *stats16)
{
    if (!ip)
        return -1;
    och_graph *g = ip->g;
    ochip_ctx *ctx = ip->ctx;
    if ((n_images && (!images_bgr || !positions)) || (n_images && model >= g->models.size()))
    {
        g->error = "och_initial_processing_step: bad argument";
        return -1;
    }
    const auto t_step = clk::now();
    // ---- init (pipeline.cpp:535-546): the stages take the ids the previous step's finalize calls left
    const std::vector<size_t> previous_loaded_ids = std::move(ip->next_loaded_ids);
    const std::vector<size_t> previous_linked_ids = std::move(ip->next_linked_ids);
    ip->next_loaded_ids.clear();
    ip->next_linked_ids.clear();
    const int team = omp_get_max_threads();
    LinkStage link(ctx);
    if (!previous_loaded_ids.empty())
    {
        link.init(g->graph, previous_loaded_ids);
        link.prepare_index(g->graph);
        // the 40 px subsets and rays of the batch's images and of the earlier images they are linked to
        std::vector<size_t> touched;
        std::unordered_map<size_t, char> seen;
        for (const NodeLinks &l : link.links())
        {
            if (seen.emplace(l.node_id, 1).second)
                touched.push_back(l.node_id);
            for (size_t other : l.link_ids)
                if (seen.emplace(other, 1).second)
                    touched.push_back(other);
        }
        link.prepare_images(g->graph, touched, team);
    }
    RelaxStage relax;
    if (!previous_linked_ids.empty())
    {
        RelaxConfig cfg;
        cfg.options = OPT_ORIENTATION | OPT_GROUND_PLANE;
        relax.init(g->graph, previous_linked_ids, false, true, cfg);
    }
    const double t_init = since(t_step);

    // ---- the runners of the three stages side by side (:548-556)
    std::vector<extracted_features> loaded(n_images);
    std::string load_error;
    double t_extract = 0, t_link = 0, t_relax = 0, total_features = 0, total_sparse = 0;
    bool ok_load = true, ok_link = true;
    auto load_runner = [&]() {
        if (!n_images)
            return;
        const auto t0 = clk::now();
        ok_load = extract_features_stream(
            ctx, images_bgr, n_images, width, height, max_keypoints, images_on_device != 0, std::max(1, team / 2),
            [&](uint32_t first, uint32_t count, extracted_features *f) {
                for (uint32_t i = 0; i < count; i++)
                {
                    total_features += (double)f[i].features.size();
                    total_sparse += (double)f[i].num_sparse_features;
                    loaded[first + i] = std::move(f[i]);
                }
            },
            &load_error);
        t_extract = since(t0);
    };
    auto link_runner = [&]() {
        if (previous_loaded_ids.empty())
            return;
        const auto t0 = clk::now();
        ok_link = run_link_runners(g, ctx, link);
        t_link = since(t0);
    };
    auto relax_runner = [&]() {
        if (previous_linked_ids.empty())
            return;
        const auto t0 = clk::now();
        ochip_ctx *rctx = nullptr;
        if (ochip_ctx_sibling(ctx, 12, &rctx) != OCHIP_OK)
            rctx = ctx;
        else if (stage_priority())
            (void)ochip_ctx_set_priority(rctx, 1);
        auto runners = relax.get_runners(rctx, g->graph);
        run_parallel(runners, relax.runner_contexts());
        t_relax = since(t0);
    };
    const auto t_run = clk::now();
    if (sequential)
    {
        load_runner();
        link_runner();
        relax_runner();
    }
    else
    {
        std::thread tl(link_runner), tr(relax_runner);
        load_runner();
        tl.join();
        tr.join();
    }
    const double t_runners = since(t_run);
    if (!ok_load)
    {
        g->error = load_error;
        return -1;
    }
    if (!ok_link)
        return -1;

    // ---- finalize, in the reference's order (:558-560)
    const auto t_fin = clk::now();
    if (n_images)
    {
        ip->next_loaded_ids = add_survey_nodes(g, n_images, model, positions, nullptr, node_ids_out);
        for (uint32_t i = 0; i < n_images; i++)
        {
            image &img = g->graph.getNode(ip->next_loaded_ids[i])->payload;
            img.features = std::move(loaded[i].features);
            img.num_sparse_features = loaded[i].num_sparse_features;
            img.coarse_subset = std::move(loaded[i].coarse_subset);
            img.coarse_spacing = loaded[i].coarse_spacing;
        }
    }
    if (!previous_loaded_ids.empty())
        ip->next_linked_ids = link.finalize(g->graph);
    int rc = 0;
    ip->next_relaxed_ids.clear();
    if (!previous_linked_ids.empty())
    {
        for (const auto &group : relax.finalize(g->graph))
            ip->next_relaxed_ids.insert(ip->next_relaxed_ids.end(), group.begin(), group.end());
        if (!relax.error().empty())
        {
            g->error = relax.error();
            rc = -1;
        }
        for (const surface_model &s : relax.getSurfaceModels())
            ip->surfaces.push_back(s);
    }
    ip->steps++;
    if (stats16)
    {
        const double v[16] = {since(t_step),
                              t_init,
                              t_runners,
                              since(t_fin),
                              t_extract,
                              t_link,
                              t_relax,
                              total_features,
                              total_sparse,
                              (double)previous_loaded_ids.size(),
                              (double)previous_linked_ids.size(),
                              (double)relax.timers.solves,
                              (double)relax.timers.iterations_total,
                              relax.timers.setup_host,
                              relax.timers.device,
                              (double)ip->next_linked_ids.size()};
        std::memcpy(stats16, v, sizeof v);
    }
    return rc;
}

} // extern "C"
