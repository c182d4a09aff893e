// One survey's load + link stages over several ranks (one process per GPU): the reference parallelises ONE survey over
// its workers - the load runners are one closure per image (src/pipeline/load_stage.cpp:36-50), the link runners one
// closure per directed pair (src/pipeline/link_stage.cpp:75-112), all run by Pipeline::Impl's worker loop
// (src/pipeline/pipeline.cpp:42-49,543-560).  Here a rank takes a contiguous block of the images (extract) and the
// directed pairs whose owner image (load_link.hpp: the source, or the later source when the pair is linked both ways)
// lies in that block (match, RANSAC, decomposition).  Pairs are independent units: no collective runs inside the
// stages.  Two exchanges carry data between ranks, both left to the caller's transport (RCCL all-gather in production,
// any in tests):
//   subsets  after the extraction: every image's 40 px subset (feature index, pixel, descriptor: 84 bytes per feature,
//            ~0.3 MB per image), so that a rank can match its pairs against images of other blocks;
//   edges    after the linking: every pair's result as a compact record (subset positions and Hamming counts, inlier
//            match indices, homography, decomposed poses: ~10 KB per pair); every rank imports the others' records and
//            runs LinkStage::finalize, whose sort (link_stage.cpp:123-127) makes the graph's edge list - ids included -
//            the one a single process builds.
// Afterwards every rank holds the same edges; the feature lists of the images stay with the rank that extracted them
// (the relax reads edges, positions and models only).
#include "../../../include/oc_host.h"

#include "load_link.hpp"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>

#include <omp.h>

using namespace opencalibration_amd;

struct och_shard
{
    och_graph *g = nullptr;
    ochip_ctx *ctx = nullptr;
    uint32_t rank = 0, world = 1, n_images = 0, lo = 0, hi = 0;
    std::vector<size_t> ids;
    std::unordered_map<size_t, uint32_t> image_of;
    std::vector<owned_pair> local, remote; // this rank's pairs: both images in the block / at least one outside
    std::vector<uint8_t> needs_desc;       // per image: one of `remote` touches it
    std::vector<uint8_t> buf;              // the last export
    double total = 0, sparse = 0;
    double seconds[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // extract, local links done, subsets export, import, remote links, edges export, import, finalize
    std::chrono::steady_clock::time_point t_begin;
};

static double since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

extern "C"
{

void och_shard_block(uint32_t n_images, uint32_t rank, uint32_t world, uint32_t *first, uint32_t *count)
{
    // contiguous blocks whose sizes differ by at most one (parallel.source_block)
    const uint32_t base = n_images / world, rem = n_images % world;
    *first = rank * base + std::min(rank, rem);
    *count = base + (rank < rem ? 1u : 0u);
}

och_shard *och_shard_begin(och_graph *g, ochip_ctx *ctx, uint32_t n_images, uint32_t model, const double *positions,
                           const double *orientations, uint32_t rank, uint32_t world, uint64_t *node_ids_out)
{
    if (!g || !ctx || !positions || model >= g->models.size() || world == 0 || rank >= world)
    {
        if (g)
            g->error = "och_shard_begin: bad argument";
        return nullptr;
    }
    if (g->graph.size_nodes() != 0)
    {
        // the link stage's k-NN runs over every node of the graph: a pair whose partner is an image of an earlier survey
        // has no owner among this survey's blocks and would be dropped without a word, so the sharded edge set would
        // differ from the single-process one
        g->error = "och_shard_begin: the graph already holds nodes; a sharded survey starts from an empty graph";
        return nullptr;
    }
    auto *s = new och_shard();
    s->g = g;
    s->ctx = ctx;
    s->rank = rank;
    s->world = world;
    s->n_images = n_images;
    s->t_begin = std::chrono::steady_clock::now();
    uint32_t first, count;
    och_shard_block(n_images, rank, world, &first, &count);
    s->lo = first;
    s->hi = first + count;
    s->ids = add_survey_nodes(g, n_images, model, positions, orientations, node_ids_out);
    for (uint32_t b = 0; b < n_images; b++)
        s->image_of.emplace(s->ids[b], b);
    g->link = std::make_unique<LinkStage>(ctx);
    LinkStage &link = *g->link;
    link.enable_sharding();
    link.init(g->graph, s->ids);
    link.prepare_index(g->graph);
    s->needs_desc.assign(n_images, 0);
    for (const owned_pair &op : pair_owners(link.links()))
    {
        if (op.owner < s->lo || op.owner >= s->hi)
            continue;
        const uint32_t a = (uint32_t)op.pair.first; // links are in image order
        const auto it = s->image_of.find(op.pair.second);
        if (it == s->image_of.end())
        {
            g->error = "och_shard_begin: a link's partner is not an image of this survey";
            delete s;
            return nullptr;
        }
        const uint32_t b = it->second;
        const bool a_in = a >= s->lo && a < s->hi, b_in = b >= s->lo && b < s->hi;
        if (a_in && b_in)
            s->local.push_back(op);
        else
        {
            s->remote.push_back(op);
            if (!a_in)
                s->needs_desc[a] = 1;
            if (!b_in)
                s->needs_desc[b] = 1;
        }
    }
    return s;
}

void och_shard_destroy(och_shard *s)
{
    delete s;
}

/* counts4: images of the block, pairs linked inside the block, pairs that need another block's subsets, images of other
 * blocks those pairs touch */
void och_shard_counts(const och_shard *s, uint64_t *counts4)
{
    counts4[0] = s->hi - s->lo;
    counts4[1] = s->local.size();
    counts4[2] = s->remote.size();
    counts4[3] = (uint64_t)std::count(s->needs_desc.begin(), s->needs_desc.end(), (uint8_t)1);
}

int och_shard_load_link_local(och_shard *s, const uint8_t *images_bgr, int width, int height, uint32_t max_keypoints,
                              int images_on_device)
{
    const auto t0 = std::chrono::steady_clock::now();
    double t_extract = 0;
    if (!load_link_stream(s->g, s->ctx, *s->g->link, s->ids, s->lo, s->hi - s->lo, images_bgr, width, height, max_keypoints,
                          images_on_device != 0, s->local, &s->total, &s->sparse, &t_extract))
        return -1;
    s->seconds[0] += t_extract;
    s->seconds[1] += since(t0);
    return 0;
}

/* The block's subsets as one buffer (valid until the next export or the shard's destruction) */
int och_shard_subsets_export(och_shard *s, const void **buf, uint64_t *bytes)
{
    const auto t0 = std::chrono::steady_clock::now();
    s->buf.clear();
    s->g->link->export_subsets(s->g->graph, s->ids, s->lo, s->hi - s->lo, s->buf);
    *buf = s->buf.data();
    *bytes = s->buf.size();
    s->seconds[2] += since(t0);
    return 0;
}

int och_shard_subsets_import(och_shard *s, const void *buf, uint64_t bytes)
{
    const auto t0 = std::chrono::steady_clock::now();
    // records {u32 image, u32 n, u32 idx[n] (+ pad to 8), f64 xy[2n], u64 desc[8n]}
    struct rec
    {
        uint32_t image, n;
        const uint8_t *body;
    };
    std::vector<rec> recs;
    const uint8_t *p = (const uint8_t *)buf, *end = p + bytes;
    while (p < end)
    {
        if ((size_t)(end - p) < 8)
            break;
        rec r;
        std::memcpy(&r.image, p, 4);
        std::memcpy(&r.n, p + 4, 4);
        r.body = p + 8;
        const size_t body = (size_t)r.n * 84 + (r.n & 1) * 4;
        if (r.image >= s->n_images || (size_t)(end - r.body) < body)
        {
            s->g->error = "och_shard_subsets_import: malformed buffer";
            return -1;
        }
        p = r.body + body;
        if (r.image < s->lo || r.image >= s->hi)
            recs.push_back(r);
    }
    LinkStage &link = *s->g->link;
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t k = 0; k < recs.size(); k++)
    {
        const rec &r = recs[k];
        const size_t n = r.n;
        std::vector<uint32_t> idx(n);
        std::vector<double> xy(2 * n);
        std::memcpy(idx.data(), r.body, 4 * n);
        const uint8_t *q = r.body + 4 * (n + (n & 1));
        std::memcpy(xy.data(), q, 16 * n);
        // (the buffer is 8-byte aligned record by record, so the descriptor words can be read in place)
        const uint64_t *desc = reinterpret_cast<const uint64_t *>(q + 16 * n);
        link.set_remote_subset(s->g->graph, s->ids[r.image], n, idx.data(), xy.data(), s->needs_desc[r.image] ? desc : nullptr);
    }
    s->seconds[3] += since(t0);
    return 0;
}

int och_shard_link_remote(och_shard *s)
{
    const auto t0 = std::chrono::steady_clock::now();
    LinkStage &link = *s->g->link;
    if (!s->remote.empty())
    {
        const char *renv = std::getenv("OCHIP_LINK_RUNNERS");
        const int n_runners = (int)std::min<size_t>((size_t)std::max(1, std::min(8, renv ? std::atoi(renv) : 3)),
                                                    std::max<size_t>(1, s->remote.size() / 256));
        const int threads = std::max(1, omp_get_max_threads() / n_runners);
        std::vector<std::thread> runners;
        std::vector<std::vector<LinkStage::link_pair>> part(n_runners);
        for (size_t i = 0; i < s->remote.size(); i++) // consecutive pairs (the two directions of a pair) stay together
            part[i * (size_t)n_runners / s->remote.size()].push_back(s->remote[i].pair);
        for (int r = 0; r < n_runners; r++)
        {
            ochip_ctx *rctx = nullptr;
            if (ochip_ctx_sibling(s->ctx, (uint32_t)(4 + r), &rctx) != OCHIP_OK)
            {
                s->g->error = std::string("ochip_ctx_sibling: ") + ochip_last_error(s->ctx);
                for (auto &t : runners)
                    t.join();
                return -1;
            }
            runners.emplace_back([&, r, rctx]() { link.run_pairs(s->g->graph, part[r], rctx, threads); });
        }
        for (auto &t : runners)
            t.join();
        if (!link.error.empty())
        {
            s->g->error = link.error;
            return -1;
        }
    }
    s->seconds[4] += since(t0);
    return 0;
}

int och_shard_edges_export(och_shard *s, const void **buf, uint64_t *bytes)
{
    const auto t0 = std::chrono::steady_clock::now();
    s->buf.clear();
    s->g->link->export_edges(s->image_of, s->buf);
    *buf = s->buf.data();
    *bytes = s->buf.size();
    s->seconds[5] += since(t0);
    return 0;
}

int och_shard_edges_import(och_shard *s, const void *buf, uint64_t bytes)
{
    const auto t0 = std::chrono::steady_clock::now();
    if (!s->g->link->import_edges(s->g->graph, s->ids, (const uint8_t *)buf, bytes))
    {
        s->g->error = s->g->link->error;
        return -1;
    }
    s->seconds[6] += since(t0);
    return 0;
}

/* LinkStage::finalize over the payloads of all ranks.  totals2: {features, sparse features} of this rank's block;
 * link_timers8 as och_link_stage_run; seconds9: extract, block linked, subsets export, subsets import, remote links,
 * edges export, edges import, finalize, whole stage since och_shard_begin (the exchanges themselves are the caller's). */
int och_shard_finalize(och_shard *s, double *totals2, double *link_timers8, double *seconds9)
{
    const auto t0 = std::chrono::steady_clock::now();
    LinkStage &link = *s->g->link;
    link.finalize(s->g->graph);
    s->seconds[7] += since(t0);
    if (totals2)
    {
        totals2[0] = s->total;
        totals2[1] = s->sparse;
    }
    if (link_timers8)
    {
        const LinkTimers &t = link.timers;
        const double v[8] = {t.link_init,  t.subsample,     t.upload,         t.match_device,
                             t.match_host, t.ransac_device, t.decompose_host, t.link_finalize};
        std::memcpy(link_timers8, v, sizeof v);
    }
    if (seconds9)
    {
        std::memcpy(seconds9, s->seconds, sizeof s->seconds);
        seconds9[8] = since(s->t_begin);
    }
    return 0;
}

} // extern "C"
