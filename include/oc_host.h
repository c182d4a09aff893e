/* oc_host.h — C driver API of liboc_host.so, the C++17 host side of the MI355X hot path.
 *
 * liboc_host.so holds the host code that sits above libochip.so (include/ochip.h): the reference's
 * stage classes and value types re-stated for flat-array device calls (opencalibration_amd/csrc/host).
 * A C++ application links the classes directly; this flat API exists so the same code can be driven
 * from Python (tests/, bench.py) without a C++ test harness.
 */
#ifndef OC_HOST_H
#define OC_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "ochip.h"

#ifdef __cplusplus
extern "C"
{
#endif

    /* spatially_subsample_feature_indices (src/match/match_features.cpp:8-52); out sized >= n */
    size_t och_subsample(const double *loc, const float *strength, size_t n, double spacing, size_t count,
                         uint64_t *out);

    /* ratio test + remap + std::sort of match_features_subset (src/match/match_features.cpp:94-101)
     * over the device kernel's raw output for one pair; outputs sized >= n1 */
    size_t och_matches_from_device(const ochip_match *raw, const uint64_t *idx1, size_t n1, const uint64_t *idx2,
                                   size_t n2, uint64_t *out_i1, uint64_t *out_i2, double *out_dist);

    /* ---- MeasurementGraph + LinkStage driver (opencalibration_amd/csrc/host/link_stage.hpp) ---- */
    typedef struct och_graph och_graph;
    och_graph *och_graph_create(void);
    void och_graph_destroy(och_graph *g);
    const char *och_last_error(const och_graph *g);
    /* m10 = {f, ppx, ppy, k1, k2, k3, p1, p2, pixels_cols, pixels_rows}; returns the model handle */
    uint32_t och_graph_add_model(och_graph *g, const double *m10);
    /* one image = what extract_features hands to the link stage; returns the node id */
    uint64_t och_graph_add_image(och_graph *g, const double *loc, const float *strength, const uint64_t *desc, size_t n,
                                 size_t num_sparse, uint32_t model, const double *position3);
    /* graph.addEdge(relations, source, dest) from flat arrays (a deserialised graph, a test): inl_px4 n x {pixel_1, pixel_2},
     * inl_idx3 n x {feature_index_1, feature_index_2, match_index}; matches: match_idx2 (may be NULL), match_dist (may be
     * NULL); poses32: 4 x {q xyzw, t xyz, score} or NULL.  Returns the edge id (0 + och_last_error on an unknown node). */
    uint64_t och_graph_add_edge(och_graph *g, uint64_t source_id, uint64_t dest_id, const double *H9, int is_homography,
                                size_t n_inliers, const double *inl_px4, const uint64_t *inl_idx3, size_t n_matches,
                                const uint64_t *match_idx2, const double *match_dist, const double *poses32);
    void och_graph_get_orientations(const och_graph *g, double *ori /* n_nodes x 4, node order */);
    /* work of the last link stage's match step: {directed pairs, descriptor distances needed = sum n1 * n2, subset features} */
    void och_link_match_work(const och_graph *g, double *out3);
    size_t och_graph_num_nodes(const och_graph *g);
    size_t och_graph_num_edges(const och_graph *g);
    void och_graph_node_ids(const och_graph *g, uint64_t *out);
    /* LinkStage::init + the batch runner + finalize; timers = 8 doubles {link_init, subsample, upload,
     * match_device, match_host, ransac_device, decompose_host, link_finalize} seconds */
    int och_link_stage_run(och_graph *g, ochip_ctx *ctx, const uint64_t *node_ids, size_t n, int keep_debug,
                           double *timers);
    size_t och_link_debug_count(const och_graph *g);
    void och_link_debug_pair(const och_graph *g, size_t p, uint64_t *ids2, uint64_t *n_matches, double *score,
                             uint32_t *iters3);
    void och_link_debug_matches(const och_graph *g, size_t p, uint64_t *i1, uint64_t *i2, double *dist, uint8_t *inl);
    void och_graph_edge_info(const och_graph *g, size_t e, uint64_t *ids2, uint64_t *counts2, double *H, double *poses);
    void och_graph_edge_inliers(const och_graph *g, size_t e, uint64_t *f1, uint64_t *f2, uint64_t *match_index,
                                double *px4);
    void och_graph_edge_match_distances(const och_graph *g, size_t e, double *out); /* relations.matches[i].distance */
    /* relations.matches of edge e: feature index pairs (n_matches x 2) and distances; relationType == HOMOGRAPHY */
    void och_graph_edge_matches(const och_graph *g, size_t e, uint64_t *idx2, double *dist, int *is_homography);
    void och_graph_set_orientations(och_graph *g, const double *ori /* n_nodes x 4, node order */);

    /* ---- extract (opencalibration_amd/csrc/host/extract_features.hpp): extract_features(cv::Mat) of
     *      src/extract/extract_features.cpp:11-88 for a batch of equally sized BGR images ------------------ */
    /* Per image up to max_out features at stride max_out: loc (x, y in full-resolution pixels), strength, desc
     * (8 u64); counts[i] features of image i, the first num_sparse[i] of which passed the 8 px NMS.
     * images_on_device != 0: images_bgr is a device pointer (images already resident in HBM). */
    int och_extract_features_batch(ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images, int width, int height,
                                   uint32_t max_keypoints, uint32_t max_out, double *loc, float *strength,
                                   uint64_t *desc, uint32_t *counts, uint32_t *num_sparse, int images_on_device);
    const char *och_extract_last_error(void);
    /* Host pieces of the link step, callable without a device (tests): homography_model::decompose on the inlier rays
     * (n x {measurement1, measurement2}; poses 4 x {q xyzw, t xyz, score}; returns can_decompose) and image_to_3d. */
    int och_homography_decompose(const double *H9, const double *m1m2, size_t n, double *poses);
    void och_image_to_3d(const double *px, size_t n, const double *model10, double *rays);
    /* ransac<fundamental_matrix_model> (model 0) / ransac<essential_matrix_model> (model 1) on the device
     * (ochip_ransac_epipolar_batch) for one set of correspondences: rays6 n x {measurement1, measurement2}, quality n or
     * NULL (PROSAC), threshold = the model's inlier_threshold (0.01).  M9: the matrix (row-major), inliers: n flags,
     * counts3: {iterations, improvements, inliers}.  Returns ransac()'s score, NAN on a device error. */
    double och_ransac_epipolar(ochip_ctx *ctx, int model, const double *rays6, const double *quality, size_t n, double threshold,
                               double *M9, uint8_t *inliers, uint32_t *counts3);
    /* The host tail of extract_features alone (extract_features.cpp:38-87), no device involved: kp6 rows
     * {x, y, size, angle, response, level} in cv::AKAZE's order -> loc / strength / desc as above; returns the count. */
    size_t och_extract_tail(const float *kp6, const uint64_t *desc, uint32_t n, double scale, double *loc, float *strength,
                            uint64_t *desc_out, uint64_t *num_sparse);
    /* The same tail for ONE image on the lists the device prepared (ochip_akaze_features / ochip_feature_lists_from_keypoints,
     * include/ochip.h: records, response, slot, num_sparse of that image): the host's part is the std::sort order, the copy
     * and the re-seating inside groups of equal responses; conflict != 0 runs the suppression here as well. */
    size_t och_extract_tail_prepared(const uint8_t *records, const float *response, const uint32_t *slot, uint32_t num_sparse_in,
                                     int conflict, uint32_t n, double scale, double *loc, float *strength, uint64_t *desc_out,
                                     uint64_t *num_sparse);
    /* Cumulative CPU seconds of that tail's phases, recorded while OCHIP_VERBOSE=extract is set: ordering, NMS, feature
     * records, total, and the number of images whose responses tied (they take std::sort's route). */
    void och_extract_tail_profile(double *out5);
    /* The strength order of that tail alone: the permutation of 0..n-1 that std::sort by descending response produces
     * (extract_features.cpp:55-56; ties keep whatever order libstdc++'s introsort leaves them in).  use_std != 0 calls
     * std::sort itself, 0 the library's own implementation of the same sequence of moves (host/sort_like_std.hpp). */
    void och_sort_by_response(const float *response, uint32_t n, uint32_t *order_out, int use_std);
    /* The load stage's job for a batch of equally sized images (src/pipeline/load_stage.cpp:43-110: extract_features,
     * then one graph node per image): extract on the device in chunks (host tail of chunk k overlapped with the device
     * work of chunk k + 1), then addNode in image order.  positions: n x 3; node_ids_out: n (may be NULL);
     * totals2 (may be NULL) receives {features, sparse features} summed over the images.  Returns 0, or -1 with the
     * message in och_last_error(g). */
    int och_graph_load_images(och_graph *g, ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images, int width,
                              int height, uint32_t max_keypoints, int images_on_device, uint32_t model,
                              const double *positions, uint64_t *node_ids_out, double *totals2);

    /* Load and link overlapped, the way the reference's pipeline overlaps the stages of consecutive image batches
     * (src/pipeline/pipeline.cpp:522-570): nodes are created first (positions, optional orientations n x 4, model),
     * extraction streams chunk by chunk, and every range of links whose images all have their features is linked on
     * its own device context while later chunks are still being extracted.  The graph is the one och_graph_load_images
     * followed by och_link_stage_run produces.  link_timers8 as och_link_stage_run; stage_seconds2 = {seconds until
     * the last features were final, seconds until the graph was linked}.  Any output pointer may be NULL. */
    int och_graph_load_link_images(och_graph *g, ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images, int width,
                                   int height, uint32_t max_keypoints, int images_on_device, uint32_t model,
                                   const double *positions, const double *orientations, uint64_t *node_ids_out,
                                   double *totals2, double *link_timers8, double *stage_seconds2);

    /* ---- INITIAL_PROCESSING with the reference's software pipelining (Pipeline::Impl::initial_processing,
     *      src/pipeline/pipeline.cpp:522-570; opencalibration_amd/csrc/host/initial_processing.cpp): one step loads (extracts)
     *      the batch it is given, links the batch of the step before against everything loaded before it, and relaxes the
     *      batch of two steps before as ONE group with two rings of context cameras ({ORIENTATION, GROUND_PLANE},
     *      disable_parallelism, :545-546) - the three stages' runners side by side (:548-556), finalized in the reference's
     *      order (:558-560).  Images arrive WITHOUT orientations (types/image.hpp:31); the relax stage initialises them
     *      (src/relax/relax.cpp:44-87).  n_images == 0 drains the pipeline: call until och_initial_processing_pending() is 0.
     *      sequential != 0: the three stages one after the other on the calling thread (the test route: the same graph).
     *      stats16 (may be NULL): seconds of the step, its init, its runners, its finalize; seconds of the load, link and
     *      relax runners; features and sparse features extracted; images linked and relaxed in this step; the relax
     *      stage's solves, LM iterations, host set-up seconds, device seconds; images handed to the next step's relax. */
    typedef struct och_initial_processing och_initial_processing;
    och_initial_processing *och_initial_processing_create(och_graph *g, ochip_ctx *ctx);
    void och_initial_processing_destroy(och_initial_processing *ip);
    int och_initial_processing_pending(const och_initial_processing *ip);
    int och_initial_processing_step(och_initial_processing *ip, const uint8_t *images_bgr, uint32_t n_images, int width, int height,
                                    uint32_t max_keypoints, int images_on_device, uint32_t model, const double *positions,
                                    int sequential, uint64_t *node_ids_out, double *stats16);

    /* ---- ONE survey's load + link stages over `world` ranks, one process per GPU (opencalibration_amd/csrc/host/shard_link.cpp).
     *      The reference parallelises one survey over its workers: one load closure per image
     *      (src/pipeline/load_stage.cpp:36-50), one link closure per directed pair (src/pipeline/link_stage.cpp:75-112),
     *      run by the worker loop of src/pipeline/pipeline.cpp:42-49.  Rank r extracts a contiguous block of the images and
     *      links the directed pairs owned by that block; pairs are independent units, no collective runs inside a stage.
     *      Two exchanges between the calls are the caller's (an all-gather over RCCL, or any transport):
     *        och_shard_load_link_local  extract the block, link the pairs whose two images are both in it (streamed)
     *        och_shard_subsets_export   -> all-gather -> och_shard_subsets_import (once per other rank's buffer)
     *        och_shard_link_remote      the rank's pairs that touch another block
     *        och_shard_edges_export     -> all-gather -> och_shard_edges_import (once per other rank's buffer)
     *        och_shard_finalize         LinkStage::finalize: the same edge list, ids included, on every rank
     *      Buffers handed to the import calls must be 8-byte aligned.  The graph must hold the camera model and no nodes
     *      of this survey yet; afterwards every rank's graph has all nodes and all edges, and the feature lists of the
     *      rank's own block. ------------------------------------------------------------------------------------------- */
    typedef struct och_shard och_shard;
    void och_shard_block(uint32_t n_images, uint32_t rank, uint32_t world, uint32_t *first, uint32_t *count);
    och_shard *och_shard_begin(och_graph *g, ochip_ctx *ctx, uint32_t n_images, uint32_t model, const double *positions,
                               const double *orientations, uint32_t rank, uint32_t world, uint64_t *node_ids_out);
    void och_shard_destroy(och_shard *s);
    /* counts4: images of the block, pairs linked inside the block, pairs that touch another block, images of other
     * blocks those pairs need descriptors of */
    void och_shard_counts(const och_shard *s, uint64_t *counts4);
    /* images_bgr: the BLOCK's images (host or device pointer) */
    int och_shard_load_link_local(och_shard *s, const uint8_t *images_bgr, int width, int height, uint32_t max_keypoints,
                                  int images_on_device);
    /* *buf stays valid until the next export on this shard */
    int och_shard_subsets_export(och_shard *s, const void **buf, uint64_t *bytes);
    int och_shard_subsets_import(och_shard *s, const void *buf, uint64_t bytes);
    int och_shard_link_remote(och_shard *s);
    int och_shard_edges_export(och_shard *s, const void **buf, uint64_t *bytes);
    int och_shard_edges_import(och_shard *s, const void *buf, uint64_t bytes);
    /* totals2: {features, sparse features} of the block; link_timers8 as och_link_stage_run; seconds9: extract, block
     * linked, subsets export, subsets import, remote links, edges export, edges import, finalize, whole stage */
    int och_shard_finalize(och_shard *s, double *totals2, double *link_timers8, double *seconds9);

    /* ---- relax (opencalibration_amd/csrc/host/relax.hpp): relax(graph, nodes, cam_models, edges,
     *      {ORIENTATION, GROUND_PLANE}, {}) of src/relax/relax.cpp:122-134 ------------------------------ */
    /* Stand-alone problem from flat arrays.  graph: n_nodes x {pos3, ori4 xyzw (may be NaN)} + one shared
     * camera model; poses: node indices + orientations (in/out, NaN = uninitialised); edges: src/dst node
     * index, H (9), is_homography, inlier offsets, per inlier {px1 xy, px2 xy} + match_index, optional
     * per-edge match distances; opt_edges: whitelist order.  plane_out: 3 corners x (x,y,z).
     * summary_out (8): solves, iterations_total, last_iterations, last_initial_cost, last_final_cost,
     * last_residual_blocks, host setup seconds, device seconds.  Returns 0 or -1. */
    int och_relax_ground_plane(ochip_ctx *ctx, size_t n_nodes, const double *node_pos, const double *node_ori,
                               const double *model10, size_t n_poses, const uint64_t *pose_node, double *pose_ori,
                               size_t n_edges, const uint64_t *edge_src, const uint64_t *edge_dst,
                               const double *edge_H, const uint8_t *edge_is_homography, const uint64_t *inl_off,
                               const double *inl_px, const uint64_t *inl_match_index, const uint64_t *dist_off,
                               const double *dist, size_t n_opt_edges, const uint64_t *opt_edges, double *plane_out,
                               double *summary_out);
    const char *och_relax_last_error(void);
    /* Test hook.  on = 1: every ground-plane relax set-up from now on repeats gridFilterMatchesPerImage and the block
     * assembly (relax_problem.cpp:234-309, :388-560) with the host code and fails unless the blocks the device built
     * (ochip_plane_setup_*, include/ochip.h) equal them bit for bit; 0: off (default); < 0: unchanged.  Returns the
     * number of set-ups compared so far. */
    int och_debug_relax_setup_check(int on);
    /* Every node of a linked graph as one group, every edge whitelisted (the single-group global relax,
     * src/pipeline/pipeline.cpp:653-655).  ori_inout: n_nodes x 4 in node order. */
    int och_graph_relax_ground_plane(och_graph *g, ochip_ctx *ctx, double *ori_inout, double *plane_out,
                                     double *summary_out);
    /* The same relax with the residual-block evaluation sharded over `world` ranks (one process per GPU, each holding
     * the same graph); `exchange` all-gathers the per-pair records, see ochip_relax_set_shard in ochip.h.  Results
     * are bit-identical to the unsharded call on every rank. */
    int och_graph_relax_ground_plane_sharded(och_graph *g, ochip_ctx *ctx, double *ori_inout, double *plane_out,
                                             double *summary_out, uint32_t rank, uint32_t world,
                                             ochip_relax_exchange_fn exchange, void *user);

    /* ---- relax, every flavour the device runs (opencalibration_amd/csrc/host/relax_mesh.hpp):
     *      relax(graph, nodes, cam_models, edges, config, previousSurfaces) of src/relax/relax.cpp:118-134 with
     *      config.options = bits of include/opencalibration/types/relax_options.hpp:9-33 in enum order (ORIENTATION = 1,
     *      POSITION = 2, GROUND_PLANE = 4, GROUND_MESH = 8, ..., MINIMAL_MESH = 4096). ------------------------------ */
    typedef struct och_surface och_surface; /* surface_model: mesh (vertices, edges with their opposite vertices) + cloud */
    och_surface *och_surface_create(void);
    void och_surface_destroy(och_surface *s);
    void och_surface_counts(const och_surface *s, size_t *n_vertices, size_t *n_edges, size_t *n_cloud);
    /* vertices n x 3; edges n x 5 {source, dest, border, opposite 0, opposite 1} (UINT64_MAX = none); cloud n x 3 */
    void och_surface_get(const och_surface *s, double *vertices, uint64_t *edges5, double *cloud);
    void och_surface_set(och_surface *s, size_t n_vertices, const double *vertices, size_t n_edges, const uint64_t *edges5,
                         size_t n_cloud, const double *cloud);
    /* new heights for the mesh vertices (n_vertices doubles); topology and the order of the mesh's containers stay */
    void och_surface_set_heights(och_surface *s, const double *z);
    /* rebuildMesh / buildMinimalMesh (src/surface/expand_mesh.cpp) from camera positions and an optional previous surface */
    void och_rebuild_mesh(const double *cam_xyz, size_t n, const och_surface *previous, int minimal, och_surface *out);
    /* Stand-alone problem from flat arrays, as och_relax_ground_plane plus what the mesh flavour reads: per node its
     * feature locations (feat_off n_nodes + 1, feat_xy), per inlier the two feature indices (inl_feat n x 2).
     * previous / surface_out may be NULL.  summary_out (12): solves, iterations_total, last_iterations,
     * last_initial_cost, last_final_cost, last_residual_blocks, host setup seconds, device seconds, track blocks,
     * 2-ray blocks, mesh vertices, unknowns of the last solve.  model10_inout (may be NULL): the caller's cam_models entry of
     * the shared camera model, read before and written after the relax (the intrinsics flavours change it). */
    int och_relax(ochip_ctx *ctx, size_t n_nodes, const double *node_pos, const double *node_ori, const double *model10,
                  const uint64_t *feat_off, const double *feat_xy, size_t n_poses, const uint64_t *pose_node,
                  double *pose_ori, size_t n_edges, const uint64_t *edge_src, const uint64_t *edge_dst, const double *edge_H,
                  const uint8_t *edge_is_homography, const uint64_t *inl_off, const double *inl_px,
                  const uint64_t *inl_feat, const uint64_t *inl_match_index, const uint64_t *dist_off, const double *dist,
                  size_t n_opt_edges, const uint64_t *opt_edges, uint32_t options, double grid_fraction,
                  const och_surface *previous, och_surface *surface_out, double *summary_out, double *model10_inout);
    /* och_relax with what the two flavours the reference only reaches from its tests need: edge_poses32 (may be NULL) =
     * per edge the four homography decompositions 4 x {q xyzw, t xyz, score} of camera_relations::relative_poses - the
     * relative-orientation flavour (no GROUND_*, no POINTS_3D option: runRelativeOrientation, src/relax/relax.cpp:14-42)
     * reads them; points_mode >= 0 with POINTS_3D runs TestRelaxProblem (test/test_relax.cpp:470-483) instead of the driver:
     * setup3dPointProblem, then nothing (0), solve (1) or relaxObservedModelOnly (2); points_before / points_after
     * (points_cap x 3, may be NULL) receive the tracks' 3-D points after the set-up and at the end, *n_points_out their
     * number.  points_mode < 0: exactly och_relax (POINTS_3D then runs runPoints, relax.cpp:103-115). */
    int och_relax_ex(ochip_ctx *ctx, size_t n_nodes, const double *node_pos, const double *node_ori, const double *model10,
                     const uint64_t *feat_off, const double *feat_xy, size_t n_poses, const uint64_t *pose_node,
                     double *pose_ori, size_t n_edges, const uint64_t *edge_src, const uint64_t *edge_dst, const double *edge_H,
                     const uint8_t *edge_is_homography, const uint64_t *inl_off, const double *inl_px,
                     const uint64_t *inl_feat, const uint64_t *inl_match_index, const uint64_t *dist_off, const double *dist,
                     size_t n_opt_edges, const uint64_t *opt_edges, uint32_t options, double grid_fraction,
                     const och_surface *previous, och_surface *surface_out, double *summary_out, double *model10_inout,
                     const double *edge_poses32, int points_mode, double *points_before, double *points_after,
                     size_t points_cap, size_t *n_points_out);
    /* Every node of a linked graph as one group, every edge whitelisted, any flavour.  ori_inout: n_nodes x 4. */
    int och_graph_relax(och_graph *g, ochip_ctx *ctx, double *ori_inout, uint32_t options, double grid_fraction,
                        const och_surface *previous, och_surface *surface_out, double *summary_out);

    /* The same with the evaluation of the residual blocks sharded over `world` ranks (one process per GPU, each holding the
     * same graph edges): the single global group of the reference's FINAL_GLOBAL_RELAX ({ORIENTATION, GROUND_MESH},
     * src/pipeline/pipeline.cpp:645-664) and the plane flavour alike; `exchange` as for
     * och_graph_relax_ground_plane_sharded.  Bit-identical to the unsharded call on every rank. */
    int och_graph_relax_sharded(och_graph *g, ochip_ctx *ctx, double *ori_inout, uint32_t options, double grid_fraction,
                                const och_surface *previous, och_surface *surface_out, double *summary_out, uint32_t rank,
                                uint32_t world, ochip_relax_exchange_fn exchange, void *user);

    /* RelaxStage (src/pipeline/relax_stage.cpp): init (partition into floor(n / 50) groups - 150 with free intrinsics -
     * by spectral clustering of the link graph, or one group with two rings of context cameras when disable_parallelism),
     * the groups' runners (concurrently, on sibling device contexts), finalize (write-back + merged surface).
     * node_ids may be NULL with relax_all != 0.  max_groups > 0: trim_groups(max_groups) before running.
     * group_of_node (n_nodes, may be NULL) receives for every node the group it is a primary node of (0 = largest) or -1.
     * summary_out as och_relax (sums over the groups), summary_out[12] = number of groups run. */
    int och_relax_stage_run(och_graph *g, ochip_ctx *ctx, const uint64_t *node_ids, size_t n_ids, int relax_all,
                            int disable_parallelism, uint32_t options, double grid_fraction, size_t max_groups,
                            const och_surface *previous, och_surface *surface_out, int64_t *group_of_node,
                            double *summary_out);
    /* The same stage in steps, for the groups of one survey over `world` ranks (one process per GPU, the same graph on every
     * rank): groups are independent during their solves (src/pipeline/relax_stage.cpp:95-111), rank r runs groups
     * r, r + world, ... of the largest-first list and nothing is exchanged inside a solve.  begin = init (+ trim_groups);
     * run_groups = this rank's runners; export -> all-gather -> import (once per other rank's buffer) moves the groups'
     * results (orientations, camera models, surfaces); end = finalize (write-back and mergeSurfaceModels - the
     * point-count-weighted vertex mean of src/surface/refine_mesh.cpp:931-1010 - over ALL groups in group order, so the
     * result is the single-process one) and destroys the handle.  och_relax_stage_run is begin + run_groups(0, 1) + end. */
    typedef struct och_relax_stage och_relax_stage;
    och_relax_stage *och_relax_stage_begin(och_graph *g, const uint64_t *node_ids, size_t n_ids, int relax_all,
                                           int disable_parallelism, uint32_t options, double grid_fraction, size_t max_groups,
                                           const och_surface *previous, int64_t *group_of_node);
    size_t och_relax_stage_num_groups(const och_relax_stage *st);
    int och_relax_stage_run_groups(och_relax_stage *st, ochip_ctx *ctx, uint32_t rank, uint32_t world);
    int och_relax_stage_export(och_relax_stage *st, uint32_t rank, uint32_t world, const void **buf, uint64_t *bytes);
    int och_relax_stage_import(och_relax_stage *st, const void *buf, uint64_t bytes);
    int och_relax_stage_end(och_relax_stage *st, och_surface *surface_out, double *summary_out);
    /* the partition alone (no device): group_of_node as above, position_in_group (may be NULL) the node's place in its
     * group's list; returns the number of groups */
    size_t och_relax_partition(const och_graph *g, size_t num_groups, int64_t *group_of_node, int64_t *position_in_group);
    /* convertModel (src/distort/invert_distortion.cpp:105-191): the inverse lens model fitted to a forward one (to_inverse),
     * or the forward model fitted to an inverse one; m10 as och_graph_add_model */
    void och_convert_model(const double *m10, int to_inverse, double *out10);
    /* mergeSurfaceModels of n surfaces (src/surface/refine_mesh.cpp:916-1016) */
    void och_merge_surfaces(const och_surface *const *surfaces, size_t n, och_surface *out);

    /* ---- after a relax changed a camera model: the write-back half of RelaxGroup::finalize
     *      (src/relax/relax_group.cpp:125-177).  och_graph_set_model replaces the intrinsics of model `model` (m10 as for
     *      och_graph_add_model; every image sharing the model sees the change, as with the reference's
     *      shared_ptr<CameraModel>); och_graph_refit_edges then re-fits EVERY edge of the graph on its previous inliers:
     *      correspondences from the current models, three rounds of fitInliers + evaluate on the device
     *      (ochip_refit_homography_batch), decomposition and inlier assembly on the host. */
    int och_graph_set_model(och_graph *g, uint32_t model, const double *m10);
    int och_graph_refit_edges(och_graph *g, ochip_ctx *ctx);

    /* ---- the reference's on-disk formats (opencalibration_amd/csrc/host/graph_io.hpp): the MeasurementGraph as
     *      graph.json (serialize / deserialize, src/io/serialize_MeasurementGraph.cpp:204-591,
     *      src/io/deserialize_MeasurementGraph.cpp:30-272), a surface mesh as ASCII PLY (src/io/serialize_MeshGraph.cpp,
     *      src/io/deserialize_MeshGraph.cpp) and the checkpoint directory (saveCheckpoint / loadCheckpoint /
     *      validateCheckpoint, src/io/checkpoint.cpp:155-337).  Loading REPLACES the handle's graph and model table;
     *      0 = ok, -1 + och_last_error(g) otherwise. ------------------------------------------------------------------ */
    char *och_graph_to_json(const och_graph *g, size_t *len); /* malloc'd, NUL-terminated: och_free */
    void och_free(void *p);
    int och_graph_from_json(och_graph *g, const char *text, size_t len);
    int och_graph_save_json(och_graph *g, const char *path);
    int och_graph_load_json(och_graph *g, const char *path);
    /* per node in the graph's order (any pointer may be NULL): id, index into the model table, features, sparse features */
    void och_graph_node_table(const och_graph *g, uint64_t *ids, uint32_t *model_index, uint64_t *n_features, uint64_t *n_sparse);
    /* one node's payload by its place in the graph's order (any pointer may be NULL): feature locations n x 2, strengths,
     * descriptors n x 8 words, position 3, orientation 4 (x y z w); its image path */
    int och_graph_node_payload(const och_graph *g, size_t index, double *loc, float *strength, uint64_t *desc,
                               double *position3, double *orientation4);
    const char *och_graph_node_path(const och_graph *g, size_t index);
    int och_graph_set_node_path(och_graph *g, size_t index, const char *path);
    size_t och_graph_num_models(const och_graph *g);
    int och_graph_get_model(const och_graph *g, uint32_t index, double *m11); /* och_graph_add_model's ten, then the id */
    int och_surface_save_ply(const och_surface *s, const char *path);
    int och_surface_load_ply(och_surface *s, const char *path);
    size_t och_surface_num_clouds(const och_surface *s);
    void och_surface_cloud_sizes(const och_surface *s, uint64_t *sizes);
    void och_surface_set_clouds(och_surface *s, size_t n_clouds, const uint64_t *sizes, const double *xyz);
    typedef struct och_checkpoint och_checkpoint;
    int och_checkpoint_validate(const char *dir); /* 1 = metadata.json and graph.json exist */
    /* state: a PipelineState name (types/pipeline_state.hpp:25-55); info4: state_run_count, origin latitude, longitude, 0 */
    int och_checkpoint_save(const char *dir, och_graph *g, const och_surface *const *surfaces, size_t n_surfaces,
                            const char *state, const double *info4);
    och_checkpoint *och_checkpoint_load(const char *dir, och_graph *g); /* NULL + och_last_error(g) on failure */
    void och_checkpoint_destroy(och_checkpoint *cp);
    size_t och_checkpoint_num_surfaces(const och_checkpoint *cp);
    const char *och_checkpoint_state(const och_checkpoint *cp);
    void och_checkpoint_info(const och_checkpoint *cp, double *info4);
    int och_checkpoint_get_surface(const och_checkpoint *cp, size_t index, och_surface *out);

    /* ---- dense guided matching (opencalibration_amd/csrc/host/dense_stereo.hpp): densifyMesh(graph, surfaces) of
     *      src/dense/dense_stereo.cpp:66-403 with the surface's mesh as surfaces[0]; the triangulated points are appended
     *      to the surface as one more cloud.  stats10 (may be NULL): images, dense features, queries sent to the device,
     *      accepted matches, tracks, points, then seconds {index build, rays + predictions, device, tracks}.  match_pairs
     *      (may be NULL, match_cap pairs): accepted matches as measurement ids (image offset + dense feature number). */
    int och_densify_mesh(och_graph *g, ochip_ctx *ctx, och_surface *surface, double *stats10, uint64_t *match_pairs,
                         size_t match_cap);
    uint32_t och_hilbert_xy2d(int order, int x, int y); /* types/hilbert.hpp:9-28 */

    /* ---- mesh refinement (opencalibration_amd/csrc/host/refine_mesh.hpp; src/surface/refine_mesh.cpp:15-909) ----------
     * refineByPointDensity / refineAtPoint on the surface's mesh with its clouds; countPointsPerTriangle as rows of
     * (three vertices, {count, distance variance}) in the order the triangles first receive a point; the triangle under
     * points (TriangleLocator).  och_mesh_refinement_run: the MESH_REFINEMENT state of the pipeline
     * (src/pipeline/pipeline.cpp:666-819) - minimal mesh, RelaxStage {ORIENTATION, GROUND_MESH} at grid fraction
     * 0.1 / 2^level on the device, count, refine or advance the level - repeated until the state is left or max_steps
     * runs were made; log8 rows {level, grid fraction, gsd, triangles above threshold, max points per triangle,
     * triangles created, mesh vertices, repeat flag}.  Returns the steps made, -1 + och_last_error(g) on a device error. */
    size_t och_refine_by_point_density(och_surface *s, size_t max_points_per_triangle, double min_distance_variance,
                                       int max_iterations, double min_triangle_size);
    size_t och_refine_at_point(och_surface *s, double x, double y, int levels);
    size_t och_count_points_per_triangle(const och_surface *s, uint64_t *tri3, double *stats2, size_t cap);
    void och_surface_locate(const och_surface *s, const double *xy, size_t n, uint64_t *tri3);
    int och_mesh_refinement_run(och_graph *g, ochip_ctx *ctx, och_surface *surface, int max_steps, double *log8);

#ifdef __cplusplus
}
#endif
#endif
