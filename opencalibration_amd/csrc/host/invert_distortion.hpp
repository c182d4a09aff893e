// Forward <-> inverse lens model (reference: include/opencalibration/distort/invert_distortion.hpp,
// src/distort/invert_distortion.cpp:105-191).  The relax flavours with free intrinsics optimise the INVERSE model
// (pixel -> ray in closed form, include/opencalibration/distort/distort_keypoints.hpp:97-116) and convert back afterwards.
#pragma once

#include "types.hpp"

namespace opencalibration_amd
{

// InverseDifferentiableCameraModel<double>: the same fields, the distortion polynomial applied to (px - pp) / f
struct InverseCameraModel : CameraModel
{
};

InverseCameraModel convertModel(const CameraModel &standardModel);
CameraModel convertModel(const InverseCameraModel &invertedModel, size_t id);

void image_to_3d(const double keypoint[2], const InverseCameraModel &model, double ray[3]); // closed form
void image_from_3d(const double ray[3], const CameraModel &model, double pixel[2]);         // forward model

} // namespace opencalibration_amd
