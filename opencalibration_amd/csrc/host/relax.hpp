// Host side of the relax step (reference: include/opencalibration/relax/relax.hpp:12-15,
// relax_problem.hpp).  Problem assembly (pose lookup, grid filter, triangle lookup) stays on the host
// as in the reference; the Ceres solve is replaced by ochip_relax_solve on the device.
#pragma once

#include "types.hpp"

#include "../../../include/ochip.h"

namespace opencalibration_amd
{

struct NodePose // include/opencalibration/types/node_pose.hpp
{
    size_t node_id;
    double orientation[4]; // x y z w
    double position[3];
};

struct surface_model_plane // the part of surface_model the ground-plane flavour produces: one triangle
{
    double corner[3][3] = {{NAN, NAN, NAN}, {NAN, NAN, NAN}, {NAN, NAN, NAN}};
};

struct RelaxTimers
{
    double setup_host = 0; // grid filter + block assembly
    double device = 0;     // problem upload + LM solves
    int solves = 0;
    int iterations_total = 0; // sum of summary.iterations.size() over solves ("LM iters")
    int last_iterations = 0;
    double last_initial_cost = 0, last_final_cost = 0;
    int last_residual_blocks = 0;
};

// One process per GPU, every rank calling relax_ground_plane on the same graph: the residual-block evaluation is
// sharded over the ranks and `exchange` all-gathers the per-pair records (ochip_relax_set_shard, include/ochip.h).
struct RelaxShard
{
    uint32_t rank = 0, world = 1;
    ochip_relax_exchange_fn exchange = nullptr;
    void *user = nullptr;
};

// relax(graph, nodes, cam_models, edges_to_optimize, {ORIENTATION, GROUND_PLANE}, {}) — src/relax/relax.cpp:122-134
// routed to runGroundPlane (:44-87).  edges_to_optimize holds edge ids in whitelist order.
// Returns false (poses untouched) if the device reported an error; `error` then has the text.
bool relax_ground_plane(ochip_ctx *ctx, const MeasurementGraph &graph, std::vector<NodePose> &nodes,
                        const std::vector<size_t> &edges_to_optimize, surface_model_plane *surface,
                        RelaxTimers *timers, std::string *error, const RelaxShard *shard = nullptr);

// Test hook: on = 1 makes every ground-plane set-up repeat the grid filter and the block assembly with the host code
// (relax_util.hpp) and fail unless the device's blocks equal them bit for bit; on < 0 leaves the switch alone.  Returns
// the number of set-ups compared so far.
int relax_setup_check(int on);
bool relax_setup_check_on();   // (for the other flavours' set-ups)
void relax_setup_check_passed();

} // namespace opencalibration_amd
