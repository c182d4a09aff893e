// LinkStage with the reference's interface (src/pipeline/link_stage.hpp:25-30: init / get_runners /
// finalize), re-stated for a device: instead of one closure per pair, get_runners returns a few batch runners
// (contiguous ranges of the source images), each pushing its directed pairs through libochip.so
// on its own device context, so that the host phases of one runner (ratio test + std::sort, decompose +
// assembleInliers) overlap the device phases (Hamming 2-NN, RANSAC) of the others.  The caller runs the
// runners concurrently, as the reference's pipeline does with its closures.
#pragma once

#include "match_features.hpp"
#include "ransac.hpp"
#include "types.hpp"

#include <functional>
#include <map>
#include <mutex>
#include <unordered_map>

namespace opencalibration_amd
{

struct LinkTimers // seconds; names follow the reference's PerformanceMeasure keys where one exists
{
    double link_init = 0;          // "Link init"
    double subsample = 0;          // part of "Link runner coarse match" in the reference
    double upload = 0;             // host -> HBM (descriptors, keypoints)
    double match_device = 0;       // "Link runner coarse match": device kernel + result fetch
    double match_host = 0;         // ratio test + std::sort + PROSAC order
    double ransac_device = 0;      // "Link runner coarse ransac": device
    double decompose_host = 0;     // decompose + assembleInliers
    double link_finalize = 0;      // "Link finalize"
};

class LinkStage
{
  public:
    struct pair_debug // per directed pair, in runner order; kept only when keep_debug is set
    {
        size_t node_id, match_node_id;
        std::vector<feature_match> matches;
        std::vector<uint8_t> inliers;
        double score = 0;
        uint32_t iterations = 0, improvements = 0;
        bool can_decompose = false;
    };

    // runners: number of concurrent batch runners (0: OCHIP_LINK_RUNNERS or the default of 3)
    explicit LinkStage(ochip_ctx *ctx, int runners = 0);

    // link_stage.cpp:13-38.  The reference queries the GPS KD-tree LoadStage::finalize filled with every
    // loaded image (load_stage.cpp:102-103); here the graph's nodes are searched directly (exact kNN,
    // k = 10 including the node itself).
    void init(const MeasurementGraph &graph, const std::vector<size_t> &node_ids);

    // link_stage.cpp:41-117
    std::vector<std::function<void()>> get_runners(const MeasurementGraph &graph);

    // link_stage.cpp:119-143
    std::vector<size_t> finalize(MeasurementGraph &graph);

    // Streaming use (host/load_link.cpp): after init(), prepare_index() once, then - as the features of images
    // arrive - prepare_images() for them and run_range() for every range of links whose images are all prepared
    // (any thread, one device context per concurrent call), then finalize().
    void prepare_index(const MeasurementGraph &graph);
    void prepare_images(const MeasurementGraph &graph, const std::vector<size_t> &node_ids, int threads);
    void run_range(const MeasurementGraph &graph, size_t link_begin, size_t link_end, ochip_ctx *ctx, int omp_threads);
    // the same for an explicit set of directed pairs {index into links(), id of the image to match against}: which
    // batch a pair runs in does not show in the graph (finalize() orders the edges), so a caller can put the two
    // directions of a pair into the same batch, where the device matches both from one pass over their distances
    typedef std::pair<size_t, size_t> link_pair;
    void run_pairs(const MeasurementGraph &graph, const std::vector<link_pair> &pairs, ochip_ctx *ctx, int omp_threads);
    const std::vector<NodeLinks> &links() const
    {
        return _links;
    }
    // work of the match step: directed pairs, descriptor distances they need (sum of n1 * n2 over the pairs, n = the
    // 40 px subsets), subset features summed over the images
    void match_work(double out[3]) const
    {
        out[0] = out[1] = out[2] = 0;
        for (const auto &s : _subsets)
            out[2] += (double)s.size();
        for (const NodeLinks &l : _links)
        {
            auto a = _prepared_index.find(l.node_id);
            for (size_t other : l.link_ids)
            {
                auto b = _prepared_index.find(other);
                if (a == _prepared_index.end() || b == _prepared_index.end())
                    continue;
                out[0] += 1;
                out[1] += (double)_subsets[a->second].size() * (double)_subsets[b->second].size();
            }
        }
    }

    // ---- one survey over several ranks (host/shard_link.cpp; SURVEY.md section 8e: directed pairs are independent
    //      units, so ranks own disjoint sets of pairs and no collective runs inside the stage).  A rank holds the
    //      features of its own block of images only; of every other image it is given the 40 px subset - feature
    //      indices, pixel locations and (for the images its own pairs touch) descriptors - which is all a pair needs.
    //      The payloads a rank produced travel as compact records (subset positions instead of pixels) and are imported
    //      by every other rank before finalize(), whose sort restores the serial edge order.
    void enable_sharding()
    {
        _sharded = true;
    }
    // idx / xy (2 per feature) / desc (8 words per feature, may be NULL: the image is only looked up, never matched)
    void set_remote_subset(const MeasurementGraph &graph, size_t node_id, size_t n, const uint32_t *idx, const double *xy,
                           const uint64_t *desc);
    // {u32 image, u32 n, u32 idx[n], f64 xy[2n], u64 desc[8n]} per image, appended to `out`; image = index into node_ids
    void export_subsets(const MeasurementGraph &graph, const std::vector<size_t> &node_ids, size_t first, size_t count,
                        std::vector<uint8_t> &out) const;
    // payloads this rank produced since the last export, appended to `out` (image numbers index `node_ids`)
    void export_edges(const std::unordered_map<size_t, uint32_t> &image_of, std::vector<uint8_t> &out) const;
    // payloads another rank produced; false (and `error`) on a malformed buffer or an image without a subset
    bool import_edges(const MeasurementGraph &graph, const std::vector<size_t> &node_ids, const uint8_t *buf, size_t bytes);
    size_t payload_count() const
    {
        return _all_inlier_measurements.size();
    }

    bool keep_debug = false;
    std::vector<pair_debug> debug;
    LinkTimers timers; // summed over the concurrent runners (so the phases can add up to more than the wall time)
    std::string error; // non-empty if the runner failed (the reference has no exceptions on this path)

  private:
    struct edge_payload
    {
        size_t loop_index;
        size_t node_id;
        size_t match_node_id;
        camera_relations relations;
        std::vector<uint32_t> subset_pos; // sharded runs: per match its two positions in the images' 40 px subsets
    };
    struct remote_subset // an image whose features live on another rank
    {
        std::vector<double> xy;     // 2 per subset feature
        std::vector<uint64_t> desc; // 8 per subset feature, or empty
    };
    void prepare(const MeasurementGraph &graph); // 40 px subsets + unit rays of every image the links touch
    void run_batch(const MeasurementGraph &graph, const std::vector<link_pair> &pairs, ochip_ctx *ctx, int omp_threads);

    ochip_ctx *_ctx;
    int _runners;
    std::vector<edge_payload> _all_inlier_measurements;
    std::mutex _measurement_mutex;
    std::vector<NodeLinks> _links;
    std::unordered_map<size_t, size_t> _prepared_index; // node id -> position in _subsets / _rays
    std::vector<std::vector<size_t>> _subsets;
    std::vector<std::vector<double>> _rays;
    std::vector<remote_subset> _remote; // parallel to _subsets; xy empty = the image's features are in the graph
    bool _sharded = false;
    EvalOrderCache _eval_cache;
};

} // namespace opencalibration_amd
