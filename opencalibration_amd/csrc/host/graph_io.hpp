// The reference's on-disk formats for the data either side of the hot path: the MeasurementGraph as
// graph.json (src/io/serialize_MeasurementGraph.cpp:204-591, src/io/deserialize_MeasurementGraph.cpp:30-272), a surface
// mesh as ASCII PLY (src/io/serialize_MeshGraph.cpp:31-138, src/io/deserialize_MeshGraph.cpp:28-149) and the checkpoint
// directory that holds them (src/io/checkpoint.cpp:22-337).  A graph.json written by the reference after any pipeline
// state loads here as the input of the link / relax stages, and what is written here loads in the reference.
#pragma once

#include "relax_mesh.hpp"

#include <iosfwd>

namespace opencalibration_amd
{

// io/serialize.hpp:14, io/deserialize.hpp:8.  The writer lays the text out as rapidjson's PrettyWriter with
// kFormatSingleLineArray does; numbers are the shortest decimal that reads back to the same double (rapidjson's
// Grisu2 prints the same digits except in the rare cases where it is one digit longer - the VALUES are identical).
// The thumbnail (a PNG, OpenCV is not here) and the metadata object are carried as text.
bool serialize(const MeasurementGraph &graph, std::ostream &out);
bool deserialize(const std::string &json, MeasurementGraph &graph, std::string *error = nullptr);

// io/serialize.hpp:15, io/deserialize.hpp:9.  Mesh node and edge ids are insertion indices here (relax_mesh.hpp): the
// writer uses them as the file's ids, the reader numbers the file's vertices and edges in file order and maps the
// opposite-corner ids through the same table.  Vertex coordinates go through `ostream << double` (6 significant
// digits) exactly as in the reference.
bool serialize(const MeshGraph &graph, std::ostream &out);
bool deserialize(std::istream &ply, MeshGraph &graph);

struct CheckpointData // io/checkpoint.hpp:12-20; the pipeline state by its name (types/pipeline_state.hpp:25-55)
{
    MeasurementGraph graph;
    std::vector<surface_model> surfaces;
    double origin_latitude = 0.0, origin_longitude = 0.0;
    std::string state = "INITIAL_PROCESSING";
    uint64_t state_run_count = 0;
};
bool saveCheckpoint(const CheckpointData &data, const std::string &checkpoint_dir, std::string *error = nullptr);
bool loadCheckpoint(const std::string &checkpoint_dir, CheckpointData &data, std::string *error = nullptr);
bool validateCheckpoint(const std::string &checkpoint_dir);

} // namespace opencalibration_amd
