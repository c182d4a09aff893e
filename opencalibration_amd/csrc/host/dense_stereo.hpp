// Dense guided matching (reference: include/opencalibration/dense/dense_stereo.hpp, src/dense/dense_stereo.cpp:66-403;
// SURVEY.md §8 f1): every dense feature's ray is intersected with the surface mesh, the point is predicted into the ten
// nearest other cameras, the dense features around the prediction are searched by descriptor ON THE DEVICE
// (ochip_dense_match, csrc/dense.hip), matches become tracks (union-find), tracks become 3-D points appended to the
// first surface's cloud.
#pragma once

#include "relax_mesh.hpp"

namespace opencalibration_amd
{

struct DenseStats // not in the reference
{
    size_t images = 0, dense_features = 0, queries = 0, matches = 0, tracks = 0, points = 0;
    double index_seconds = 0, rays_seconds = 0, device_seconds = 0, tracks_seconds = 0;
};

// densifyMesh(graph, surfaces, progress).  Returns false with `error` set when the device reports an error (the surfaces
// are then untouched).  matches_out (optional): every accepted match as a pair of measurement ids (image offset + dense
// feature number, the reference's numbering), in the order the reference's loops produce them.
bool densifyMesh(ochip_ctx *ctx, const MeasurementGraph &graph, std::vector<surface_model> &surfaces, DenseStats *stats,
                 std::string *error, std::vector<std::pair<size_t, size_t>> *matches_out = nullptr);

// types/hilbert.hpp:9-28
uint32_t hilbert_xy2d(int order, int x, int y);

} // namespace opencalibration_amd
