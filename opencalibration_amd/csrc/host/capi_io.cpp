// liboc_host.so: C ABI of the on-disk formats (graph_io.hpp) - graph.json, surface PLY, checkpoint directory.
#include "../../../include/oc_host.h"

#include "capi_graph.hpp"
#include "graph_io.hpp"

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

using namespace opencalibration_amd;

struct och_checkpoint
{
    CheckpointData data;
};

namespace
{
// the handle's model table: one entry per distinct camera model, in the order the nodes first use them
void adopt_graph(och_graph *g, MeasurementGraph &&graph)
{
    g->graph = std::move(graph);
    g->link.reset();
    g->models.clear();
    for (const auto &node : g->graph.nodes())
    {
        const auto &m = node.payload.model;
        bool known = false;
        for (const auto &have : g->models)
            known = known || have == m;
        if (m && !known)
            g->models.push_back(m);
    }
}
} // namespace

extern "C"
{

void och_free(void *p)
{
    std::free(p);
}

char *och_graph_to_json(const och_graph *g, size_t *len)
{
    std::ostringstream out;
    if (!serialize(g->graph, out))
        return nullptr;
    const std::string s = out.str();
    char *buf = (char *)std::malloc(s.size() + 1);
    if (!buf)
        return nullptr;
    std::memcpy(buf, s.data(), s.size());
    buf[s.size()] = '\0';
    if (len)
        *len = s.size();
    return buf;
}

int och_graph_from_json(och_graph *g, const char *text, size_t len)
{
    MeasurementGraph graph;
    std::string why;
    if (!deserialize(std::string(text, len), graph, &why))
    {
        g->error = "graph.json: " + why;
        return -1;
    }
    adopt_graph(g, std::move(graph));
    return 0;
}

int och_graph_save_json(och_graph *g, const char *path)
{
    std::ofstream out(path, std::ios::binary);
    if (!out.is_open() || !serialize(g->graph, out))
    {
        g->error = std::string("cannot write ") + path;
        return -1;
    }
    return 0;
}

int och_graph_load_json(och_graph *g, const char *path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open())
    {
        g->error = std::string("cannot open ") + path;
        return -1;
    }
    const std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    return och_graph_from_json(g, text.data(), text.size());
}

// per node, in the graph's order: id, model index (into the handle's model table), feature count, sparse feature count
void och_graph_node_table(const och_graph *g, uint64_t *ids, uint32_t *model_index, uint64_t *n_features, uint64_t *n_sparse)
{
    size_t i = 0;
    for (const auto &node : g->graph.nodes())
    {
        if (ids)
            ids[i] = node.id;
        if (model_index)
        {
            uint32_t mi = 0;
            for (size_t k = 0; k < g->models.size(); k++)
                if (g->models[k] == node.payload.model)
                    mi = (uint32_t)k;
            model_index[i] = mi;
        }
        if (n_features)
            n_features[i] = node.payload.features.size();
        if (n_sparse)
            n_sparse[i] = node.payload.num_sparse_features;
        i++;
    }
}

/* one node's payload by its place in the graph's order (any pointer may be NULL): feature locations n x 2, strengths n,
 * descriptors n x 8 words, position 3, orientation 4 (x y z w) */
int och_graph_node_payload(const och_graph *g, size_t index, double *loc, float *strength, uint64_t *desc, double *position3,
                           double *orientation4)
{
    if (index >= g->graph.size_nodes())
        return -1;
    const image &img = g->graph.nodes()[index].payload;
    for (size_t i = 0; i < img.features.size(); i++)
    {
        if (loc)
            loc[2 * i] = img.features[i].location[0], loc[2 * i + 1] = img.features[i].location[1];
        if (strength)
            strength[i] = img.features[i].strength;
        if (desc)
            std::memcpy(desc + 8 * i, img.features[i].descriptor, 64);
    }
    if (position3)
        std::memcpy(position3, img.position, 24);
    if (orientation4)
        std::memcpy(orientation4, img.orientation, 32);
    return 0;
}
const char *och_graph_node_path(const och_graph *g, size_t index)
{
    return index < g->graph.size_nodes() ? g->graph.nodes()[index].payload.path.c_str() : "";
}
int och_graph_set_node_path(och_graph *g, size_t index, const char *path)
{
    if (index >= g->graph.size_nodes())
        return -1;
    g->graph.nodes()[index].payload.path = path;
    return 0;
}

size_t och_graph_num_models(const och_graph *g)
{
    return g->models.size();
}

/* m11: och_graph_add_model's ten numbers, then the model id */
int och_graph_get_model(const och_graph *g, uint32_t index, double *m11)
{
    if (index >= g->models.size())
        return -1;
    const CameraModel &m = *g->models[index];
    m11[0] = m.focal_length_pixels;
    m11[1] = m.principle_point[0];
    m11[2] = m.principle_point[1];
    for (int i = 0; i < 3; i++)
        m11[3 + i] = m.radial_distortion[i];
    m11[6] = m.tangential_distortion[0];
    m11[7] = m.tangential_distortion[1];
    m11[8] = (double)m.pixels_cols;
    m11[9] = (double)m.pixels_rows;
    m11[10] = (double)m.id;
    return 0;
}

int och_surface_save_ply(const och_surface *s, const char *path)
{
    std::ofstream out(path, std::ios::binary);
    return (out.is_open() && serialize(s->s.mesh, out)) ? 0 : -1;
}

int och_surface_load_ply(och_surface *s, const char *path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open())
        return -1;
    MeshGraph mesh;
    if (!deserialize(in, mesh))
        return -1;
    s->s.mesh = std::move(mesh);
    return 0;
}

/* the surface's point clouds one by one (a checkpoint stores them as separate files): sizes[n_clouds] then all points */
size_t och_surface_num_clouds(const och_surface *s)
{
    return s->s.cloud.size();
}
void och_surface_cloud_sizes(const och_surface *s, uint64_t *sizes)
{
    for (size_t i = 0; i < s->s.cloud.size(); i++)
        sizes[i] = s->s.cloud[i].size();
}
void och_surface_set_clouds(och_surface *s, size_t n_clouds, const uint64_t *sizes, const double *xyz)
{
    s->s.cloud.assign(n_clouds, point_cloud());
    size_t k = 0;
    for (size_t i = 0; i < n_clouds; i++)
        for (uint64_t j = 0; j < sizes[i]; j++, k++)
            s->s.cloud[i].push_back({xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2]});
}

int och_checkpoint_validate(const char *dir)
{
    return validateCheckpoint(dir) ? 1 : 0;
}

/* info4: state_run_count, origin_latitude, origin_longitude, unused */
int och_checkpoint_save(const char *dir, och_graph *g, const och_surface *const *surfaces, size_t n_surfaces, const char *state,
                        const double *info4)
{
    CheckpointData data;
    data.graph = g->graph; // a copy: the handle keeps its graph
    for (size_t i = 0; i < n_surfaces; i++)
        data.surfaces.push_back(surfaces[i]->s);
    if (state)
        data.state = state;
    if (info4)
    {
        data.state_run_count = (uint64_t)info4[0];
        data.origin_latitude = info4[1];
        data.origin_longitude = info4[2];
    }
    std::string why;
    if (!saveCheckpoint(data, dir, &why))
    {
        g->error = why;
        return -1;
    }
    return 0;
}

/* loads the directory: the graph goes into g, the rest stays in the returned handle (NULL + och_last_error(g) on failure) */
och_checkpoint *och_checkpoint_load(const char *dir, och_graph *g)
{
    auto *cp = new och_checkpoint();
    std::string why;
    if (!loadCheckpoint(dir, cp->data, &why))
    {
        g->error = why;
        delete cp;
        return nullptr;
    }
    adopt_graph(g, std::move(cp->data.graph));
    cp->data.graph = MeasurementGraph();
    return cp;
}

void och_checkpoint_destroy(och_checkpoint *cp)
{
    delete cp;
}

size_t och_checkpoint_num_surfaces(const och_checkpoint *cp)
{
    return cp->data.surfaces.size();
}

const char *och_checkpoint_state(const och_checkpoint *cp)
{
    return cp->data.state.c_str();
}

void och_checkpoint_info(const och_checkpoint *cp, double *info4)
{
    info4[0] = (double)cp->data.state_run_count;
    info4[1] = cp->data.origin_latitude;
    info4[2] = cp->data.origin_longitude;
    info4[3] = 0;
}

int och_checkpoint_get_surface(const och_checkpoint *cp, size_t index, och_surface *out)
{
    if (index >= cp->data.surfaces.size())
        return -1;
    out->s = cp->data.surfaces[index];
    return 0;
}

} // extern "C"
