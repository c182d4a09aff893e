// Shared definition of the opaque och_graph handle of include/oc_host.h.
#pragma once

#include "link_stage.hpp"
#include "relax_mesh.hpp"

struct och_graph
{
    opencalibration_amd::MeasurementGraph graph;
    std::vector<std::shared_ptr<opencalibration_amd::CameraModel>> models;
    std::unique_ptr<opencalibration_amd::LinkStage> link;
    std::string error;
};

struct och_surface // surface_model: mesh + point clouds
{
    opencalibration_amd::surface_model s;
};
