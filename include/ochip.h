/* ochip.h — C ABI of libochip.so, the MI355X (gfx950) device side of the opencalibration hot path.
 *
 * The reference (jkflying/opencalibration) has no plugin/FFI layer; its seam is the stage triple
 * init/get_runners/finalize (src/pipeline/link_stage.hpp:25-30, relax_stage.hpp:24-35) over
 * value-semantic free functions.  Each entry point below replaces the inner loop of one of those
 * functions and is what a reference-side binding would call (INTEGRATION.md shows the C++ stub).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * OCHIP_E* code and never throws; ochip_last_error() gives the text.  Pointers are HOST pointers
 * unless the name says `_dev`.  A context owns one HIP device, its streams and device arenas; calls
 * on one context must be serialised by the caller (the host stages call from one thread per context,
 * SURVEY.md §8b), different contexts are independent.
 */
#ifndef OCHIP_H
#define OCHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

#define OCHIP_OK 0
#define OCHIP_EINVAL (-1)  /* bad argument */
#define OCHIP_ENOMEM (-2)  /* device or host allocation failed */
#define OCHIP_EHIP (-3)    /* HIP runtime error, see ochip_last_error */
#define OCHIP_ENODEV (-4)  /* no usable gfx950 device */
#define OCHIP_ESTATE (-5)  /* call order violated (e.g. match before upload) */

#define OCHIP_DESC_WORDS64 8 /* std::bitset<486> = 8 x u64, bits 486..511 zero (feature_2d.hpp:9-21) */
#define OCHIP_DESC_BITS 486
#define OCHIP_NO_SECOND 0xFFFFu /* second_count sentinel: fewer than two reference descriptors */

    typedef struct ochip_ctx ochip_ctx;

    typedef struct ochip_pair
    {
        uint32_t image_1; /* query image (source node, link_stage.cpp:57) */
        uint32_t image_2; /* reference image (match_node_id, link_stage.cpp:67) */
    } ochip_pair;

    /* per query descriptor of image_1, in subset order */
    typedef struct ochip_match
    {
        uint32_t best_k;       /* position in image_2's uploaded subset of the nearest descriptor; lowest k on ties */
        uint16_t best_count;   /* Hamming distance (popcount of xor over 486 bits) */
        uint16_t second_count; /* distance of the 2nd nearest (== best_count on a tie); OCHIP_NO_SECOND if n2 < 2 */
    } ochip_match;

    /* kernel ids for ochip_profile_get */
    enum
    {
        OCHIP_K_MATCH = 0,
        OCHIP_K_RANSAC = 1,
        OCHIP_K_RELAX_EVAL = 2,
        OCHIP_K_RELAX_SOLVE = 3,
        OCHIP_K_AKAZE = 4,
        OCHIP_K_DENSE = 5,
        OCHIP_K_COUNT = 8
    };

    /* ---- context ------------------------------------------------------------------------------ */
    int ochip_ctx_create(int device, ochip_ctx **out);
    void ochip_ctx_destroy(ochip_ctx *ctx);
    /* The index-th sibling of ctx: another context on the same device with its own streams, scratch buffers and
     * pools, created on first use and destroyed with ctx.  Lets independent batches (e.g. the link stage's runners,
     * src/pipeline/link_stage.cpp:41-117) be in flight at once: one host thread per context.  Kernel times of
     * siblings are included in ochip_profile_get(ctx, ...). */
    int ochip_ctx_sibling(ochip_ctx *ctx, uint32_t index, ochip_ctx **out);
    /* Re-creates the context's compute stream with the highest (high != 0) or lowest stream priority of the device: a
     * latency-bound solve that shares the GPU with throughput kernels of other contexts is scheduled ahead of them.  The
     * context must be idle. */
    int ochip_ctx_set_priority(ochip_ctx *ctx, int high);
    const char *ochip_last_error(const ochip_ctx *ctx); /* ctx may be NULL: error of the last failed create */
    int ochip_device_info(const ochip_ctx *ctx, char *name, size_t name_len, int *compute_units, size_t *hbm_bytes);
    int ochip_synchronize(ochip_ctx *ctx);

    /* ---- descriptor store (replaces the packed_2 build, src/match/match_features.cpp:62-66) ----- */
    /* Size the arena once: n_images slots, total_descriptors descriptors over all images. */
    int ochip_descriptors_reserve(ochip_ctx *ctx, uint32_t n_images, uint64_t total_descriptors);
    /* Copy one image's 40-px-subset descriptors (subset order, n x 8 u64) into HBM.  The caller keeps
     * ownership of `desc`.  An image may be uploaded once. */
    int ochip_upload_descriptors(ochip_ctx *ctx, uint32_t image_id, const uint64_t *desc, uint32_t n);
    int ochip_descriptor_count(const ochip_ctx *ctx, uint32_t image_id, uint32_t *n);

    /* ---- brute-force Hamming 2-NN (replaces the double loop src/match/match_features.cpp:71-93) - */
    /* For pair p, out[out_offset[p] + i] describes query i of image_1 (i < n(image_1)).
     * out_offset has n_pairs entries; the caller sizes `out` as sum of n(image_1).  The Lowe ratio
     * test (:94), the index remap through indices_2 (:86) and the std::sort (:100-101) stay on the
     * host so libstdc++'s tie order is preserved. */
    int ochip_match_batch(ochip_ctx *ctx, const ochip_pair *pairs, uint32_t n_pairs, const uint64_t *out_offset,
                          ochip_match *out);
    /* The tail of match_features_subset (match_features.cpp:94-101) on the device, on the records ochip_match_launch left
     * in HBM (same pairs, offsets and total): Lowe's ratio test (best < 0.8 * second on count / 486 as doubles) and the
     * std::sort by descending distance - libstdc++'s permutation among equal distances, csrc/std_sort.hip.  The sorted
     * matches stay in HBM for ochip_ransac_homography_batch_sorted; counts_out[p] = matches of pair p; fallback_out[p] != 0:
     * the pair's sort needs libstdc++'s heap sort (not restated on the device) - the caller then takes the host route
     * (ochip_match_fetch, host ratio test and sort, ochip_ransac_homography_batch) for the batch. */
    int ochip_match_sort(ochip_ctx *ctx, const ochip_pair *pairs, uint32_t n_pairs, const uint64_t *out_offset, uint64_t out_total,
                         uint32_t *counts_out, uint8_t *fallback_out);
    /* Asynchronous flavour: results stay in HBM until ochip_match_fetch (lets the caller overlap). */
    int ochip_match_launch(ochip_ctx *ctx, const ochip_pair *pairs, uint32_t n_pairs, const uint64_t *out_offset,
                           uint64_t out_total);
    int ochip_match_fetch(ochip_ctx *ctx, ochip_match *out, uint64_t out_total);

    /* ---- extract: AKAZE keypoints + 486-bit M-LDB descriptors for a batch of equally sized BGR images
     *      (replaces cvtColor + resize(INTER_AREA, max side 1600) + cv::AKAZE::detectAndCompute of
     *      src/extract/extract_features.cpp:25-36; AKAZE restated from its publication) ---- */
    /* images_bgr: n_images x height x width x 3 bytes.  Per image up to max_kp keypoints are written, in
     * cv::AKAZE's detection order (evolution level, then row, then column of the extremum - the order the
     * unstable std::sort of extract_features.cpp:55-56 starts from, which decides the result when responses tie):
     * kp6[(i*max_kp + k)*6] = {x, y, diameter, angle (radians), response, evolution level}
     * in pixels of the working (downscaled) image, desc[(i*max_kp + k)*8] = descriptor words (bit j = word j>>6,
     * bit j&63, the packing of extract_features.cpp:47-51); counts[i] = keypoints of image i;
     * work_wh = working width, height. */
    int ochip_akaze_batch(ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images, int width, int height,
                          uint32_t max_kp, float *kp6, uint64_t *desc, uint32_t *counts, int *work_wh);

    /* Same with the images already resident in HBM (device pointer): the PCIe upload is outside the call. */
    int ochip_akaze_batch_dev(ochip_ctx *ctx, const uint8_t *images_bgr_dev, uint32_t n_images, int width, int height,
                              uint32_t max_kp, float *kp6, uint64_t *desc, uint32_t *counts, int *work_wh);

    /* ---- extract_features' tail (src/extract/extract_features.cpp:38-87) on the device ---- */
    /* The reference rescales the keypoints, std::sorts them by response (unstable: libstdc++'s order among equal
     * responses, restated move for move on the device, csrc/std_sort.hip), runs a greedy 8 px suppression in that order
     * and emits [sparse..., dense...].  All arrays are the caller's (host; page-locked ones copy at link speed):
     *   records    [n][max_kp + 1][88]  the image's output list [sparse..., dense...] as feature_2d records (location =
     *                                   pt / scale as doubles, strength, 4 bytes padding, 8 descriptor words):
     *                                   counts[i] + 1 records (the strongest feature heads both lists: the reference
     *                                   visits its seed again, :60-66)
     *   response   [n][max_kp]          responses in detection order
     *   slot       [n][max_kp]          slot[s] = where the keypoint of detection index s lies in `records` (the seed:
     *                                   its sparse slot, 0)
     *   num_sparse [n]                  length of the sparse list
     *   conflict   [n]                  non-zero: this image's responses drive introsort to its depth limit, where
     *                                   libstdc++ switches to a heap sort the device does not restate; `records` then
     *                                   holds the features in no particular order and the host orders and suppresses
     *                                   them itself (host/extract_features.cpp: extract_tail_prepared), using `response`
     *                                   and `slot`.  Never seen on detector output; organ-pipe responses do it. */
    /* Optionally also spatially_subsample_feature_indices(features, subset_spacing, num_sparse)
     * (src/match/match_features.cpp:8-52: the sparse features' indices std::sorted by strength, then accepted greedily when
     * no accepted feature lies within the spacing) - what LinkStage computes per image before matching (link_stage.cpp:63-65
     * with 40 px):
     *   subset          [n][OCHIP_SUBSET_CAP]  accepted indices into the feature list, in the function's output order
     *   num_subset      [n]                    their number (> OCHIP_SUBSET_CAP: row truncated, the host computes it)
     *   subset_conflict [n]                    non-zero: this sort hit the depth limit, the host computes the subset
     * Leave subset NULL (or subset_spacing 0) to skip it. */
#define OCHIP_SUBSET_CAP 16384
    typedef struct ochip_feature_lists
    {
        uint8_t *records;
        float *response;
        uint32_t *slot;
        uint32_t *num_sparse;
        uint8_t *conflict;
        uint32_t *subset;
        uint32_t *num_subset;
        uint8_t *subset_conflict;
        double subset_spacing;
    } ochip_feature_lists;
    /* ochip_akaze_batch / _dev with the tail prepared: counts as there, `lists` instead of kp6 / desc.
     * scale = working / original size (the reference's `scale`, :26), nms_radius in working pixels (8, :58). */
    int ochip_akaze_features(ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images, int width, int height, uint32_t max_kp,
                             double nms_radius, uint32_t *counts, const ochip_feature_lists *lists, int *work_wh);
    int ochip_akaze_features_dev(ochip_ctx *ctx, const uint8_t *images_bgr_dev, uint32_t n_images, int width, int height,
                                 uint32_t max_kp, double nms_radius, uint32_t *counts, const ochip_feature_lists *lists, int *work_wh);
    /* the same preparation for keypoints the caller already has (kp6 / desc / counts as ochip_akaze_batch returns them) */
    int ochip_feature_lists_from_keypoints(ochip_ctx *ctx, const float *kp6, const uint64_t *desc, const uint32_t *counts, uint32_t n_images,
                            uint32_t max_kp, int work_w, int work_h, double scale, double nms_radius,
                            const ochip_feature_lists *out);

    /* ---- synthetic views: benchmark / test DATA generated directly in HBM (no algorithm of the path) ---- */
    /* A jittered ground lattice of Gaussian blobs on the plane z = a x + b y rendered through pinhole cameras;
     * all views of one seed show the same ground.  cams: n x {pos3, quat4 (x y z w)}; model3 = {f, ppx, ppy};
     * plane2 = {a, b}; lattice3 = {x0, y0, spacing}.  Images are BGR (grey replicated), n x height x width x 3. */
    int ochip_synth_views_alloc(ochip_ctx *ctx, uint32_t n_images, int width, int height, uint8_t **images_dev);
    void ochip_synth_views_free(ochip_ctx *ctx, uint8_t *images_dev);
    int ochip_synth_views_read(ochip_ctx *ctx, const uint8_t *images_dev, uint32_t index, int width, int height,
                               uint8_t *host_out); /* copy view `index` back to the host (for the CPU checker) */
    int ochip_synth_render_views(ochip_ctx *ctx, uint8_t *images_dev, uint32_t first_image, uint32_t n_images, int width,
                                 int height, const double *cams, const double *model3, const double *plane2,
                                 const double *lattice3, uint32_t seed);

    /* ---- keypoints -> unit rays (replaces image_to_3d of src/distort/distort_keypoints.cpp:68-103,
     *      hoisted from once per match to once per keypoint) ------------------------------------------ */
    /* xy: n x 2 pixel locations in the same subset order as the descriptors of image_id (n must equal
     * the uploaded descriptor count).  model8 = {focal_length_pixels, pp_x, pp_y, k1, k2, k3, p1, p2}
     * (DifferentiableCameraModel, include/opencalibration/types/camera_model.hpp:22-60). */
    int ochip_upload_keypoints(ochip_ctx *ctx, uint32_t image_id, const double *xy, uint32_t n, const double *model8);

    /* One-call flavour of reserve + upload_descriptors + upload_keypoints for a whole batch: image i owns
     * counts[i] consecutive rows of desc_all (x 8 u64) and xy_all (x 2 f64); models8 is n_images x 8. */
    int ochip_upload_batch(ochip_ctx *ctx, uint32_t n_images, const uint32_t *counts, const uint64_t *desc_all,
                           const double *xy_all, const double *models8);

    /* Page-locked host memory for the large transfers (match results, RANSAC inputs): PCIe copies from
     * pageable memory run at a fraction of the link rate. */
    int ochip_host_alloc(ochip_ctx *ctx, size_t bytes, void **out);
    void ochip_host_free(ochip_ctx *ctx, void *p);

    /* ---- homography RANSAC, one job per directed pair (replaces ransac<homography_model>,
     *      src/model_inliers/ransac.cpp:53-257 + homography_model.cpp:19-136) -------------------------- */
    typedef struct ochip_ransac_match
    {
        uint32_t k1, k2;   /* positions of the matched keypoints in the uploaded subsets of image_1 / image_2 */
        uint16_t count;    /* Hamming distance; correspondence.quality = count * (1.0/486) */
        uint16_t reserved; /* 0 */
    } ochip_ransac_match;

    typedef struct ochip_ransac_job
    {
        uint32_t image_1, image_2;
        uint32_t n;            /* number of matches = correspondences, in match_features_subset's output order */
        uint32_t rng_state;    /* std::default_random_engine(42) state after std::shuffle(eval_order) */
        uint64_t match_offset; /* into matches[], sorted_idx[] and inliers[] */
        uint64_t eval_offset;  /* into eval_order[]: iota(n) after std::shuffle (ransac.cpp:158) */
    } ochip_ransac_job;

    typedef struct ochip_ransac_result
    {
        double H[9];           /* row-major homography (model.homography), NaN if nothing was fitted */
        double score;          /* return value of ransac(): evaluate(best) / n */
        uint32_t iterations;   /* loop trips executed */
        uint32_t n_inliers;
        uint32_t improvements; /* times a hypothesis beat the best score */
        uint32_t reserved;
    } ochip_ransac_result;

    /* sorted_idx: per job, iota(n) std::sort-ed by quality ascending (the PROSAC order, ransac.cpp:83-90);
     * both it and eval_order come from the host's libstdc++ so tie order is the reference's.
     * inliers: total_matches bytes, 1 = inlier, indexed like matches[]. */
    int ochip_ransac_homography_batch(ochip_ctx *ctx, const ochip_ransac_job *jobs, uint32_t n_jobs,
                                      const ochip_ransac_match *matches, const uint32_t *sorted_idx,
                                      uint64_t total_matches, const uint32_t *eval_order, uint64_t eval_total,
                                      double inlier_threshold, ochip_ransac_result *results, uint8_t *inliers);

    /* ---- the fundamental- and essential-matrix models under the same RANSAC loop (replaces
     *      ransac<fundamental_matrix_model> / ransac<essential_matrix_model>, src/model_inliers/ransac.cpp:260-261 over
     *      fundamental_matrix_model.cpp:12-217 (8-point fit, rank-2 constraint, Sampson error, DEGENSAC) and
     *      essential_matrix_model.cpp:12-123 (5-point linear fit, equal singular values); tests-only in the reference).
     *      model: 0 = fundamental (threshold 0.01), 1 = essential.  corr6: per correspondence {measurement1,
     *      measurement2} (any scale: both are divided by their z); sorted_idx / eval_order as for the homography batch
     *      (sorted_idx is read only by jobs with has_quality != 0).  results[j].H = the matrix, .score = evaluate / n. */
    typedef struct ochip_epipolar_job
    {
        uint32_t n;           /* correspondences */
        uint32_t rng_state;   /* std::default_random_engine(42) state after std::shuffle(eval_order) */
        uint32_t has_quality; /* any correspondence.quality != 0: PROSAC sampling over sorted_idx */
        uint32_t reserved;
        uint64_t corr_offset; /* into corr6 (x 6), sorted_idx and inliers */
        uint64_t eval_offset; /* into eval_order */
    } ochip_epipolar_job;
    int ochip_ransac_epipolar_batch(ochip_ctx *ctx, int model, const ochip_epipolar_job *jobs, uint32_t n_jobs,
                                    const double *corr6, const uint32_t *sorted_idx, uint64_t total,
                                    const uint32_t *eval_order, uint64_t eval_total, double inlier_threshold,
                                    ochip_ransac_result *results, uint8_t *inliers);

    /* ---- re-fit of accepted edges after the camera models changed (replaces the per-edge loop of
     *      RelaxGroup::finalize, src/relax/relax_group.cpp:137-177: distort_keypoints with the current models, then
     *      `rounds` (= 3 there) times fitInliers + evaluate starting from the previous inliers).  jobs / matches as for
     *      ochip_ransac_homography_batch (rng_state and eval_offset unused); the keypoints and the CURRENT models must
     *      have been uploaded (ochip_upload_keypoints / ochip_upload_batch).  inliers: in = the previous inlier flags,
     *      out = the flags of the last evaluate; results[j].H = the last fit, .n_inliers, .score = evaluate / n. */
    /* homography_model::decompose (homography_model.cpp:138-185) of a job's result: the four poses of
     * cv::decomposeHomographyMat(H, I) as {orientation x y z w, position x y z, score = cheirality votes of the inliers} in
     * std::stable_sort's order (absent solutions: NaN, score -1), can_decompose = decompose()'s return value, n_inliers =
     * number of inlier flags set. */
    typedef struct ochip_decomposition
    {
        double pose[4][8];
        uint32_t can_decompose, n_inliers;
    } ochip_decomposition;
    /* The same on the matches ochip_match_sort left in HBM (decomp_out: NULL or one ochip_decomposition per job): jobs[j] is pair j of that call with n = counts_out[j]; the
     * correspondences (their order is the sort's), the PROSAC order (ransac.cpp:83-90, the same device std::sort) are built
     * on the device.  matches_out[match_offset + i] = correspondence i of the job (for the host's feature_match list);
     * fallback_out[j] != 0: the PROSAC order of job j needs the host (then nothing of this call is to be used). */
    int ochip_ransac_homography_batch_sorted(ochip_ctx *ctx, const ochip_ransac_job *jobs, uint32_t n_jobs, uint64_t total_matches,
                                             const uint32_t *eval_order, uint64_t eval_total, double inlier_threshold,
                                             ochip_ransac_result *results, uint8_t *inliers, ochip_ransac_match *matches_out,
                                             uint8_t *fallback_out, ochip_decomposition *decomp_out);
    /* The two lists an accepted edge keeps (camera_relations.matches / .inlier_matches, link_stage.cpp:99-107), gathered on
     * the device right after ochip_ransac_homography_batch_sorted (same context, same batch):
     *   feature_match_out  [total_matches] x {u64 feature_index_1, u64 feature_index_2, f64 distance}  (types/feature_match.hpp:
     *                      feature_match, 24 bytes), job j's matches at its match_offset, in the sort's order
     *   inlier_match_out   [total_inliers] x {f64 pixel_1[2], pixel_2[2], u64 feature_index_1, feature_index_2, match_index}
     *                      (feature_match_denormalized, 56 bytes): job j's inliers in match order at inlier_offset[j] (the
     *                      caller sums ochip_decomposition.n_inliers)
     * feature_index[k]: index in its image's feature list of the k-th keypoint of the upload (ochip_upload_batch order) -
     * the 40 px subset's indices. */
    int ochip_edge_lists(ochip_ctx *ctx, uint32_t n_jobs, uint64_t total_matches, const uint32_t *feature_index, uint64_t n_keypoints,
                         const uint64_t *inlier_offset, uint64_t total_inliers, void *feature_match_out, void *inlier_match_out);
    int ochip_refit_homography_batch(ochip_ctx *ctx, const ochip_ransac_job *jobs, uint32_t n_jobs,
                                     const ochip_ransac_match *matches, uint64_t total_matches, uint32_t rounds,
                                     double inlier_threshold, ochip_ransac_result *results, uint8_t *inliers);

    /* ---- relax: ground-plane bundle adjustment (replaces ceres::Solver::Solve on the problem
     *      RelaxProblem::setupGroundPlaneProblem builds, src/relax/relax_problem.cpp:61-81,1390-1420) ---- */
    typedef struct ochip_relax_problem ochip_relax_problem;

    typedef struct ochip_relax_desc
    {
        uint32_t n_cams;
        const double *cam_pos;       /* n_cams x 3, constants (GPS positions are never optimised) */
        const double *cam_q;         /* n_cams x 4 initial orientation, Eigen coefficient order x y z w */
        const uint8_t *cam_optimize; /* n_cams: 1 = in _nodes_to_optimize, 0 = constant context camera */
        double plane_xy[6];          /* the three plane corners' x,y (initializeGroundPlane, :1189-1242) */
        double plane_z[3];           /* initial corner heights (the structure parameters) */
        uint8_t z_optimize[3];
        uint32_t n_blocks;           /* 2-ray PlaneIntersectionAngleCost blocks (relax_cost_function.hpp:658-684) */
        const uint32_t *blk_cam_a;   /* n_blocks: source camera */
        const uint32_t *blk_cam_b;   /* n_blocks: dest camera */
        const double *blk_rays;      /* n_blocks x 6: camera-frame unit rays of the source and dest keypoint */
        uint32_t n_prior;            /* PointsDownwardsPrior blocks (relax_cost_function.hpp:21-49) */
        const uint32_t *prior_cam;
        double huber_a;              /* HuberLoss scale of the 2-ray blocks (1 degree in radians, :68) */
        double prior_weight;         /* 1e-3 (:1297) */
    } ochip_relax_desc;

    typedef struct ochip_relax_options /* the ceres::Solver::Options fields the reference sets or relies on */
    {
        int max_num_iterations;             /* 100 (:32) */
        double initial_trust_region_radius; /* 1 (:36) */
        double function_tolerance;          /* 1e-6 */
        double gradient_tolerance;          /* 1e-10 */
        double parameter_tolerance;         /* 1e-8 */
    } ochip_relax_options;

    enum
    {
        OCHIP_RELAX_NO_PARAMETERS = 0,
        OCHIP_RELAX_CONVERGENCE_GRADIENT = 1,
        OCHIP_RELAX_CONVERGENCE_PARAMETER = 2,
        OCHIP_RELAX_CONVERGENCE_FUNCTION = 3,
        OCHIP_RELAX_CONVERGENCE_RADIUS = 4,
        OCHIP_RELAX_NO_CONVERGENCE = 5,
        OCHIP_RELAX_FAILURE = 6
    };

    typedef struct ochip_relax_summary
    {
        int termination;
        int iterations; /* summary.iterations.size(): iteration 0 + every trust-region step attempted */
        int successful_steps, unsuccessful_steps;
        int num_parameters; /* tangent dimension of the reduced program */
        int num_residual_blocks;
        double initial_cost, final_cost;
    } ochip_relax_summary;

    /* ---- set-up of the ground-plane relax: gridFilterMatchesPerImage + the residual-block list -------------------
     * Replaces, for the edges of one RelaxProblem::setupGroundPlaneProblem, src/relax/relax_problem.cpp:234-309 (score every
     * inlier match, keep the best one per cell of each image's grid: GridFilter::addMeasurement,
     * include/opencalibration/relax/grid_filter.hpp:33-51) and :388-560 with fixed intrinsics (a kept match whose rays'
     * closest approach lies over the plane's border triangle becomes a residual block), one wavefront per edge. */
    typedef struct ochip_plane_edge
    {
        uint32_t cam_a, cam_b;     /* rows of cam_pos / cam_q: the source and the destination image's pose */
        uint32_t model_a, model_b; /* rows of models10 */
        uint32_t n_inliers;
        uint32_t flags;            /* bit 0: relationType == HOMOGRAPHY (H scores the match, :279-287) */
        uint64_t inlier_offset;    /* first record of the edge in `inliers` */
        double H[9];               /* camera_relations::ransac_relation, row-major */
    } ochip_plane_edge;
    typedef struct ochip_plane_inlier
    {
        double px1[2], px2[2];   /* feature_match_denormalized::pixel_1 / pixel_2 */
        double descriptor_score; /* 1 - matches[match_index].distance (1 if the index is out of range), :273-275 */
    } ochip_plane_inlier;
    typedef struct ochip_plane_setup ochip_plane_setup;
    /* Scores and filters.  cam_pos [n_cams][3], cam_q [n_cams][4] (x, y, z, w); models10 [n_models][10] = f, ppx, ppy,
     * k1, k2, k3, p1, p2, pixels_cols, pixels_rows; triangle_xy6 = the border triangle's corners in the searcher's order
     * (after its orientation fix-up, src/surface/intersect.cpp:56-163); grid_fraction = 0.15 (:66).
     * keep_out[i] (n_inliers bytes): bit 0 = match i is on the source image's whitelist, bit 1 = on the destination's.
     * inexact_out[e] != 0 (n_edges bytes): two matches of edge e share the best score of a cell (the reference's unstable
     * std::sort decides, grid_filter.hpp:33-51) or a match lies outside the cell table - the caller computes that edge's
     * flags itself and passes them to ochip_plane_setup_override.  The handle holds device blocks of `ctx`. */
    int ochip_plane_setup_create(ochip_ctx *ctx, const ochip_plane_edge *edges, uint32_t n_edges, const ochip_plane_inlier *inliers,
                                 uint64_t n_inliers, const double *cam_pos, const double *cam_q, uint32_t n_cams,
                                 const double *models10, uint32_t n_models, const double *triangle_xy6, double grid_fraction,
                                 uint8_t *keep_out, uint8_t *inexact_out, ochip_plane_setup **out);
    int ochip_plane_setup_override(ochip_plane_setup *s, uint64_t first_inlier, uint64_t n, const uint8_t *keep);
    /* The residual blocks in edge order, matches in index order: cameras (rows of cam_pos) and the two camera-frame unit
     * rays (6 doubles) of each - the arrays ochip_relax_desc takes.  All three arrays NULL: only *n_blocks is returned; it
     * is set even when capacity is too small (OCHIP_ENOMEM). */
    int ochip_plane_setup_blocks(ochip_plane_setup *s, uint32_t *blk_cam_a, uint32_t *blk_cam_b, double *blk_rays, uint64_t capacity,
                                 uint64_t *n_blocks);
    void ochip_plane_setup_destroy(ochip_plane_setup *s);

    /* ---- the NaN bootstrap of runGroundPlane as one resident launch (csrc/relax_chain.hip) -------------------------
     * Replaces the loop of src/relax/relax.cpp:52-80: every camera of a batch that arrives without an orientation takes
     * the orientation of the pose in front of it and is relaxed - setupGroundPlaneProblem, relaxObservedModelOnly, solve -
     * before the next one.  The group's cameras, edges (in edges_to_optimize's order) and inlier matches are uploaded
     * once; ochip_plane_chain_run walks the steps on the device without a host round trip in between.
     * A step: `cam` takes the current orientation of `prev_cam` (prev_cam < 0: prev_q), the plane is the triangle tri_xy
     * (corners in the mesh searcher's order, initializeGroundPlane :1189-1242) at height z0, the edges [0, n_filter) take
     * part (gridFilterMatchesPerImage stops at the first edge without usable poses, :251-252).  mode 0: `cam` alone is
     * optimised, every other camera is constant (relax.cpp:61-68: the caller puts the first edge that touches a camera
     * without an orientation in the graph at n_filter); mode 1: every camera with cam_optimize set (relax.cpp:70-75);
     * mode 2: the group's own solve behind the loop (relax.cpp:81-84): as mode 1, but nobody takes an orientation (cam, prev_*
     * unused). */
    typedef struct ochip_plane_chain_step
    {
        uint32_t cam, mode, n_filter;
        int32_t prev_cam;
        double prev_q[4];
        double tri_xy[6];
        double z0;
    } ochip_plane_chain_step;
    typedef struct ochip_plane_chain_result
    {
        uint32_t steps_done; /* the steps before this one are finished and in cam_q_out */
        int32_t status;      /* 0: every step done; 1: step `steps_done` needs the host's grid filter (a tie for a cell's best
                                score, a match outside the cell table); 2: given up (a wait on the device ran into its limit).
                                The caller continues from that step with its own loop. */
        int32_t solves, iterations_total, last_iterations, last_residual_blocks; /* as ochip_relax_summary counts them */
        double last_initial_cost, last_final_cost;
        int32_t grid_syncs, workgroups;
        double plane_z[3];   /* the last step's corner heights, in tri_xy's order */
    } ochip_plane_chain_result;
    typedef struct ochip_plane_chain ochip_plane_chain;
    /* cam_pos [n_cams][3], cam_q [n_cams][4] (NaN: not oriented yet), cam_optimize [n_cams] (mode 1), models10 as for
     * ochip_plane_setup_create; huber_a, prior_weight as in ochip_relax_desc.  Fails with OCHIP_EINVAL for what the chain
     * does not take (more than 340 optimised cameras, a grid finer than 1 / 15, an edge from a camera to itself). */
    int ochip_plane_chain_create(ochip_ctx *ctx, const ochip_plane_edge *edges, uint32_t n_edges, const ochip_plane_inlier *inliers,
                                 uint64_t n_inliers, const double *cam_pos, const double *cam_q, const uint8_t *cam_optimize,
                                 uint32_t n_cams, const double *models10, uint32_t n_models, double grid_fraction, double huber_a,
                                 double prior_weight, const ochip_plane_chain_step *steps, uint32_t n_steps, ochip_plane_chain **out);
    /* stepped != 0: one phase per launch (the test route; same code, same numbers).  cam_q_out [n_cams][4]: the state after
     * the last finished step, normalised as RelaxProblem::solve leaves it (:1410-1413). */
    int ochip_plane_chain_run(ochip_plane_chain *chain, int stepped, double *cam_q_out, ochip_plane_chain_result *result);
    void ochip_plane_chain_destroy(ochip_plane_chain *chain);

    /* A problem (this one and the ochip_relaxg_ / ochip_relaxp_ ones below) holds device blocks and a page-locked host
       block of its context: destroy it before the context. */
    int ochip_relax_problem_create(ochip_ctx *ctx, const ochip_relax_desc *desc, ochip_relax_problem **out);
    void ochip_relax_problem_destroy(ochip_relax_problem *p);
    /* relaxObservedModelOnly (:931-984): freeze every camera, leave the plane heights variable (or undo) */
    int ochip_relax_set_cameras_constant(ochip_relax_problem *p, int constant);
    /* Levenberg-Marquardt trust-region solve; the state stays in HBM */
    int ochip_relax_solve(ochip_relax_problem *p, const ochip_relax_options *opt, ochip_relax_summary *summary);
    /* Sharded evaluation of one relax problem over `world` ranks (one process per GPU, every rank creates the same
     * problem): rank r evaluates the residual blocks of camera pairs [r * chunk, (r + 1) * chunk),
     * chunk = ceil(n_pairs / world); after every evaluation `exchange` must all-gather the record arrays in place
     * (rank r's slice of an array starts at ptr + r * bytes_per_rank; acc_bytes_per_rank is 0 for a cost-only
     * evaluation) - an RCCL all-gather over xGMI in production (torch.distributed backend "nccl"), any transport
     * in tests - and return 0 once the gathered data is visible to later work on the context's stream.  The library
     * synchronises its stream before the call.  `exchange` is called whenever it is given, also with world == 1.  Every rank then runs the same deterministic assembly and linear solve on the same
     * records, so the result is bit-identical to the unsharded solve.  This is the one exchange step of the path
     * (single-group global relax, src/pipeline/pipeline.cpp:653-655; Ceres itself is single-process). */
    typedef int (*ochip_relax_exchange_fn)(void *user, void *acc_dev, uint64_t acc_bytes_per_rank, void *cost_dev,
                                           uint64_t cost_bytes_per_rank, void *fail_dev, uint64_t fail_bytes_per_rank);
    int ochip_relax_set_shard(ochip_relax_problem *p, uint32_t rank, uint32_t world, ochip_relax_exchange_fn exchange,
                              void *user);
    /* The native transport for that exchange: RCCL all-gathers, in place, on the context's compute stream (no host
     * wait: the solver's next kernels are on the same stream).  librccl is resolved at run time; the calls fail with
     * OCHIP_EHIP where it is absent.  Rank 0 draws the id (ncclGetUniqueId) and hands it to the other ranks by any means;
     * every rank then creates its communicator on the context its relax problem lives on and passes
     * ochip_rccl_relax_exchange / the communicator as `exchange` / `user` of ochip_relax_set_shard.
     * Stands where SURVEY.md section 8b sketches ochip_allreduce_normal_eq over ncclAllReduce: the per-pair records are
     * 4 MB per evaluation at C3 against 72 MB for the dense normal equations. */
#define OCHIP_RCCL_ID_BYTES 128
    typedef struct ochip_rccl_comm ochip_rccl_comm;
    int ochip_rccl_unique_id(ochip_ctx *ctx, uint8_t *id /* OCHIP_RCCL_ID_BYTES */);
    int ochip_rccl_comm_create(ochip_ctx *ctx, const uint8_t *id, uint32_t rank, uint32_t world, ochip_rccl_comm **out);
    void ochip_rccl_comm_destroy(ochip_rccl_comm *comm);
    int ochip_rccl_comm_stats(const ochip_rccl_comm *comm, uint64_t *exchanges, uint64_t *bytes_gathered);
    int ochip_rccl_relax_exchange(void *user, void *acc_dev, uint64_t acc_bytes_per_rank, void *cost_dev,
                                  uint64_t cost_bytes_per_rank, void *fail_dev, uint64_t fail_bytes_per_rank);
    /* cam_q: n_cams x 4; ochip_relax_solve leaves the optimised cameras' quaternions normalised exactly as
     * RelaxProblem::solve does after every Solve (:1410-1413); plane_z: 3 */
    int ochip_relax_get_state(ochip_relax_problem *p, double *cam_q, double *plane_z);

    /* ---- relax, general form: ground mesh, multi-ray tracks, shared intrinsics (replaces ceres::Solver::Solve on the
     *      problems RelaxProblem::setupGroundMeshProblem / setupGroundPlaneProblem build, src/relax/relax_problem.cpp:61-120:
     *      residual blocks of 2..5 rays on a triangle of the surface mesh (relax_cost_function.hpp:601-790), mesh priors
     *      (:51-69, :119-155; relax_problem.cpp:1303-1366), downward prior (:21-49), distortion monotonicity (:157-185)).
     *      Unknowns: 3 per optimised camera (quaternion tangent), 1 per optimised mesh vertex (height), and the shared
     *      inverse lens model: focal (bounded, :496-497), principal point, radial k1..k_n (SubsetManifold, :22-23). ---- */
    typedef struct ochip_relaxg_problem ochip_relaxg_problem;

    typedef struct ochip_relaxg_desc
    {
        uint32_t n_cams;
        const double *cam_pos;       /* n_cams x 3 */
        const double *cam_q;         /* n_cams x 4, x y z w */
        const uint8_t *cam_optimize; /* n_cams */
        uint32_t n_verts;            /* mesh vertices; their heights are the structure parameters */
        const double *vert_xy;       /* n_verts x 2 */
        const double *vert_z;        /* n_verts */
        const uint8_t *vert_optimize;
        uint32_t n_blocks;           /* ray blocks, in the order the reference adds them (tracks, then 2-ray blocks) */
        const uint8_t *blk_n;        /* rays of block b: 2 (Huber loss huber_a) or 3..5 (no loss) */
        const uint8_t *blk_intr;     /* 1 = the FocalRadial functor on the shared model (:501-566), NULL = none */
        const uint32_t *blk_ray_off; /* n_blocks + 1 */
        const uint32_t *blk_tri;     /* n_blocks x 3 vertex indices (the triangle under the rays) */
        const uint32_t *ray_cam;     /* per ray */
        const double *ray_dir;       /* per ray x 3: camera-frame unit ray (used by blocks with blk_intr = 0) */
        const double *ray_px;        /* per ray x 2: pixel (used by blocks with blk_intr = 1); may be NULL */
        uint32_t n_down;             /* PointsDownwardsPrior */
        const uint32_t *down_cam;
        double down_weight;          /* 1e-3 */
        uint32_t n_diff;             /* DifferenceCost between the heights of the two ends of a mesh edge */
        const uint32_t *diff_v;      /* n_diff x 2 */
        double diff_weight;          /* 1e-4 */
        double anchor_weight;        /* DifferenceCost of every vertex against its initial height (1e-5); 0 = none */
        uint32_t n_smooth;           /* AdjacentTriangleNormalCost per inner mesh edge: vertices A B C D */
        const uint32_t *smooth_v;    /* n_smooth x 4 */
        double smooth_weight;        /* 1e-4 */
        double huber_a;              /* 1 degree in radians */
        /* shared inverse lens model (InverseDifferentiableCameraModel) of the blk_intr blocks */
        double model[8];             /* f, ppx, ppy, k1, k2, k3, p1, p2 */
        uint8_t opt_focal, opt_principal, n_radial_free; /* n_radial_free: 0..3 leading radial coefficients are variable */
        double focal_lo, focal_hi;   /* 100, 20000 */
        uint32_t mono_observations;  /* DistortionMonotonicityCost: weight sqrt(n / 10); 0 = none */
        double mono_r_max;
        /* sharded evaluation over `shard_world` ranks (0 or 1 = not sharded): every rank creates the same problem and
         * evaluates its run of the ray blocks; ochip_relaxg_set_exchange names the transport */
        uint32_t shard_rank, shard_world;
        /* relation blocks of setupDecompositionProblem (src/relax/relax_problem.cpp:40-59,311-350):
         * MultiDecomposedRotationCost (relax_cost_function.hpp:188-307) between cameras rel_cam[2i], rel_cam[2i + 1] over
         * the four homography decompositions rel_pose[32 i ..] = 4 x {q xyzw, t xyz, score}, HuberLoss(rel_huber_a) */
        uint32_t n_rel;
        const uint32_t *rel_cam;
        const double *rel_pose;
        double rel_huber_a; /* 10 degrees in radians */
    } ochip_relaxg_desc;

    int ochip_relaxg_problem_create(ochip_ctx *ctx, const ochip_relaxg_desc *desc, ochip_relaxg_problem **out);
    void ochip_relaxg_problem_destroy(ochip_relaxg_problem *p);
    /* The exchange step of a problem created with shard_world > 1 (the reference's single global group is
     * {ORIENTATION, GROUND_MESH}, src/pipeline/pipeline.cpp:645-664): after every evaluation `exchange` all-gathers the
     * ranks' runs of the record array in place - same contract as ochip_relax_set_shard; ochip_rccl_relax_exchange with a
     * communicator created on the problem's context is the native transport.  The padded layout keeps every rank's run the
     * same size; the assembly and the cost sum then run on identical arrays in an order that does not depend on the
     * sharding, so the solve is bit-identical to the unsharded one on every rank.  A ray block's record is 0.4 - 2.6 KB
     * (the packed J'J of its 9 - 24 columns), so an evaluation exchanges a few hundred MB at C3 size: the sharding buys
     * nothing against an evaluation of ~1 ms on one GPU; it exists for memory (a survey whose records do not fit one
     * device) and for the contract of BASELINE config C4. */
    int ochip_relaxg_set_exchange(ochip_relaxg_problem *p, ochip_relax_exchange_fn exchange, void *user);
    /* relaxObservedModelOnly (:931-984): 1 = everything but the mesh heights held constant, 0 = undo */
    int ochip_relaxg_set_structure_only(ochip_relaxg_problem *p, int on);
    int ochip_relaxg_solve(ochip_relaxg_problem *p, const ochip_relax_options *opt, ochip_relax_summary *summary);
    /* cam_q: n_cams x 4 (optimised cameras normalised as RelaxProblem::solve leaves them), vert_z: n_verts, model: 8;
     * any may be NULL */
    int ochip_relaxg_get_state(ochip_relaxg_problem *p, double *cam_q, double *vert_z, double *model);
    /* One evaluation at the current state for tests: total cost, and (when not NULL) the dense J'J (n x n) and J'r (n) of
     * the reduced system in the library's unknown order; order_out (n_cams + n_verts + 3 entries) receives the first
     * unknown of every camera / vertex / f / pp / k or -1. */
    int ochip_relaxg_evaluate(ochip_relaxg_problem *p, double *cost, int *n_out, double *JtJ, double *Jtr, int32_t *order_out);

    /* ---- relax with 3-D points: reprojection bundle adjustment, the points eliminated by a per-point 3 x 3 Schur
     *      complement (replaces ceres::Solver::Solve with SPARSE_SCHUR on the problem RelaxProblem::setup3dPointProblem
     *      builds, src/relax/relax_problem.cpp:122-145,986-1187: PixelErrorCost_Orientation[Focal[Radial[Tangential]]],
     *      relax_cost_function.hpp:309-500, HuberLoss(10 px), focal bounds, SubsetManifold of the radial block,
     *      DistortionMonotonicityCost :157-185).  Every point is seen by exactly two cameras - the reference makes one
     *      point per whitelisted inlier of an edge - and the points of one edge form a group: points
     *      [grp_first[g], grp_first[g + 1]) are seen by cameras grp_cam[2g] (pixels obs_px[2p]) and grp_cam[2g + 1]
     *      (obs_px[2p + 1]).  Unknowns of the reduced system: 3 per optimised camera (quaternion tangent) and the shared
     *      lens model's free parameters; the points are back-substituted. ---- */
    typedef struct ochip_relaxp_problem ochip_relaxp_problem;
    typedef struct ochip_relaxp_desc
    {
        uint32_t n_cams;
        const double *cam_pos;       /* n_cams x 3 */
        const double *cam_q;         /* n_cams x 4, x y z w */
        const uint8_t *cam_optimize; /* n_cams */
        uint32_t n_points;
        const double *point_xyz;     /* n_points x 3: the triangulated starting points */
        uint32_t n_groups;
        const uint32_t *grp_first;   /* n_groups + 1 */
        const uint32_t *grp_cam;     /* n_groups x 2 */
        const double *obs_px;        /* (2 n_points) x 2 pixels */
        int functor;                 /* 0 Orientation, 1 OrientationFocal, 2 ...Radial, 3 ...RadialTangential */
        double model[8];             /* f, ppx, ppy, k1, k2, k3, p1, p2 of the shared forward lens model */
        uint8_t opt_focal, opt_principal, n_radial_free; /* functor >= 1: f / pp variable; functor >= 2: 0..3 leading radial
                                                            coefficients variable (3 = a free Euclidean block) */
        double focal_lo, focal_hi;   /* bounds of a variable focal length (100, 20000) */
        double huber_a;              /* 10 px */
        uint32_t mono_observations;  /* DistortionMonotonicityCost (functor >= 2): weight sqrt(n / 10); 0 = none */
        double mono_r_max;
    } ochip_relaxp_desc;
    int ochip_relaxp_problem_create(ochip_ctx *ctx, const ochip_relaxp_desc *desc, ochip_relaxp_problem **out);
    void ochip_relaxp_problem_destroy(ochip_relaxp_problem *p);
    /* relaxObservedModelOnly (:931-984): 1 = cameras and lens model held constant, only the points move; 0 = undo */
    int ochip_relaxp_set_structure_only(ochip_relaxp_problem *p, int on);
    int ochip_relaxp_solve(ochip_relaxp_problem *p, const ochip_relax_options *opt, ochip_relax_summary *summary);
    /* cam_q: n_cams x 4 (optimised cameras normalised), point_xyz: n_points x 3, model: 8; any may be NULL */
    int ochip_relaxp_get_state(ochip_relaxp_problem *p, double *cam_q, double *point_xyz, double *model);

    /* ---- profiling: HIP-event time of every launch of a kernel since the last reset ------------- */
    int ochip_profile_reset(ochip_ctx *ctx);
    int ochip_profile_get(ochip_ctx *ctx, int kernel_id, uint64_t *launches, double *total_ms);
    /* descriptor distances of the match launches since the last ochip_profile_reset (siblings included): computed (a pair
     * matched in both directions from one pass counts its n1 x n2 distances once) and delivered (twice) */
    int ochip_match_work(ochip_ctx *ctx, uint64_t *computed, uint64_t *delivered);
    /* fp64 flops the relax solves issued on the matrix cores (the panel and trailing-update GEMMs of their Cholesky
     * factorisations, inside the block envelope) since the last ochip_profile_reset; their time is OCHIP_K_RELAX_SOLVE's */
    int ochip_relax_work(ochip_ctx *ctx, double *mfma_flops);
    /* roofline bookkeeping since the last ochip_profile_reset (siblings included): counters3[0] = sum over the homography
     * RANSAC jobs of loop trips x correspondences (an upper bound of the (hypothesis, correspondence) errors evaluated -
     * the SPRT exit of src/model_inliers/ransac.cpp:197-200 leaves a hypothesis early; time: OCHIP_K_RANSAC),
     * [1] / [2] = 2-ray residual blocks the ground-plane engine evaluated with / without Jacobians (OCHIP_K_RELAX_EVAL) */
    int ochip_work_counters(ochip_ctx *ctx, uint64_t *counters3);
    /* the largest reduced system a relax on this context (or a sibling) has held: its unknowns, the bytes of J'J and its
     * factor as stored (64 x 64 tiles of the block envelope, lower triangle) and what the same two matrices take dense -
     * the reference hands this system to ceres::SPARSE_NORMAL_CHOLESKY (src/relax/relax_problem.cpp:30-37) */
    int ochip_relax_memory(ochip_ctx *ctx, uint64_t *unknowns, uint64_t *stored_bytes, uint64_t *dense_bytes);

    /* ---- diagnostics ----------------------------------------------------------------------------- */
    /* libstdc++'s std::sort on the device (csrc/std_sort.hip), for the tests: segment s = [offsets[s], offsets[s + 1]) of
     * (keys, payload), sorted as std::sort with comp(a, b) = key(a) > key(b) leaves them - the permutation among equal keys
     * included.  fallback_out[s] != 0: the segment hit introsort's depth limit (libstdc++ heap-sorts there) and is NOT
     * sorted; callers sort such a segment on the host. */
    int ochip_debug_std_sort(ochip_ctx *ctx, const uint32_t *keys, const uint32_t *payload, const uint32_t *offsets, uint32_t n_segs,
                             uint32_t *keys_out, uint32_t *payload_out, uint8_t *fallback_out);
    /* out[i] = x[i] op y[i] computed on the device with the hot-path kernels' compile flags:
     * op 0: x/y, 1: sqrt(x), 2: log(x), 3: x*y+x (unfused), 4: x+y.  Host pointers. */
    int ochip_debug_fp64(ochip_ctx *ctx, int op, const double *x, const double *y, size_t n, double *out);

    /* ---- dense guided matching: the descriptor search of densifyMesh (src/dense/dense_stereo.cpp:245-283) ------------
     * The index holds, for every image, its dense features sorted by the cell of a uniform grid over the image (cell edge
     * cell_size > the search radius): feat_off (n_images + 1) offsets into desc8 (8 u64 per feature) / loc2 (x, y), per
     * image a table of ncx * ncy + 1 cell starts (relative to the image's first feature) at cell_off[i] in cell_start,
     * grid2 = {ncx, ncy} and origin2 = the grid's corner per image; a feature at (x, y) lies in cell
     * (floor((x - ox) / cell_size), floor((y - oy) / cell_size)).  A query names a feature of the index (its descriptor is
     * the probe), the image to search and the predicted pixel; the result is the nearest and second nearest Hamming
     * distance among that image's features with squared pixel distance < radius^2, the nearest one's position in the
     * image's sorted order, and how many features the disc held.  0xFFFF = no such neighbour. */
    typedef struct ochip_dense_index ochip_dense_index;
    typedef struct ochip_dense_query
    {
        uint32_t src_feature; /* position in the index (all images concatenated) */
        uint32_t cand_image;
        double px, py;
    } ochip_dense_query;
    typedef struct ochip_dense_result
    {
        uint32_t best_feature; /* position within cand_image's features */
        uint16_t best_count, second_count;
        uint32_t nearby;
    } ochip_dense_result;
    int ochip_dense_index_create(ochip_ctx *ctx, uint32_t n_images, const uint64_t *feat_off, const uint64_t *desc8,
                                 const double *loc2, const uint64_t *cell_off, const uint32_t *cell_start, const int32_t *grid2,
                                 const double *origin2, double cell_size, ochip_dense_index **out);
    void ochip_dense_index_destroy(ochip_dense_index *ix);
    int ochip_dense_match(ochip_dense_index *ix, const ochip_dense_query *queries, uint64_t n_queries, double radius,
                          ochip_dense_result *out);
    /* densifyMesh after the mesh intersections, whole survey, on the device (replaces the rest of the per-feature loop of
     * src/dense/dense_stereo.cpp:174-297 - the k nearest cameras of the hit point, its projection into each, the disc
     * search, the accept rule - and the union of the matched measurements; ochip_dense_match stays for callers that
     * build their queries themselves).
     * cams17 [n_images][17]: position (3), orientation.inverse() (4, xyzw), model10 (f ppx ppy k1 k2 k3 p1 p2 cols rows);
     * id_of_pos [total]: the reference's measurement id (image offset + dense feature number) of every index position - a
     * permutation of [0, total), anything else is OCHIP_EINVAL;
     * hits3 [total][3] by index position: where the feature's ray meets the mesh, x = NaN: nowhere (:196-207).
     * max_candidates = MAX_CANDIDATE_IMAGES (10), descriptor_bits = 486, ratio = 0.85, max_abs = 0.35 (:51-56).
     * root_out [total] by measurement id: the smallest member of the measurement's track, 0xFFFFFFFF = unmatched;
     * counts2: queries issued, matches accepted; slot_dst_out (may be NULL) [total][11]: per feature and candidate in
     * nearest-camera order the accepted match's measurement id or 0xFFFFFFFF (tests: the reference's match order). */
    int ochip_dense_link(ochip_dense_index *ix, const double *cams17, const uint32_t *id_of_pos, const double *hits3, double radius,
                         uint32_t max_candidates, uint32_t descriptor_bits, double ratio, double max_abs, uint32_t *root_out,
                         uint64_t *counts2, uint32_t *slot_dst_out);

    /* The tracks' 3-D points (dense_stereo.cpp:299-340 with triangulateTrack, :112-172) after ochip_dense_link on the same
     * index: track t = the measurement ids track_member[track_start[t] .. track_start[t + 1]) in ascending order; its first two
     * rays meet in a point, members reprojecting within max_reprojection_error px are inliers, fewer than two give no point,
     * fewer than all give the point of the first two inliers (a member >= the index's measurement count: OCHIP_EINVAL).
     * cam_q4 [n_images][4]: the images' orientations (x y z w).
     * points3_out [n_tracks][3], valid_out [n_tracks] (0: the track has no point). */
    int ochip_dense_triangulate(ochip_dense_index *ix, const double *cam_q4, uint32_t n_tracks, const uint32_t *track_start,
                                const uint32_t *track_member, double max_reprojection_error, double *points3_out, uint8_t *valid_out);

#ifdef __cplusplus
}
#endif
#endif /* OCHIP_H */
