// ORACLE — test infrastructure only (see oracle.hpp).
//
// The partitioning half of the relax stage restated:
//   include/opencalibration/geometry/KMeans.hpp:12-263           KMeans<size_t, 3> (k-center seeding, redistribution of
//                                                                small clusters' centres, sort by size after every step)
//   include/opencalibration/geometry/spectral_cluster.hpp:16-254 SpectralClustering<size_t, 3>: normalised Laplacian of the
//                                                                link graph, its D + 1 = 4 smallest eigenpairs, rows of
//                                                                the first D eigenvectors normalised -> k-means; connected
//                                                                components get their own share of the clusters
//   src/pipeline/relax_stage.cpp:28-112                          RelaxStage::init: cluster count, links, groups largest first
//   src/surface/refine_mesh.cpp:572-825, :916-1016               TriangleLocator, countPointsPerTriangle, mergeSurfaceModels
// [3P] Spectra::SymEigsSolver (external/spectra is an empty submodule here) is replaced by a dense symmetric
// eigen-decomposition (cyclic Jacobi); Spectra returns its nev = 4 smallest eigenvalues sorted largest first, so
// evectors.block<1, 3>(i, 0) are the eigenvectors of the 4th, 3rd and 2nd smallest eigenvalue, in that order.
#include "relax_full.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <queue>
#include <tuple>
#include <unordered_map>
#include <unordered_set>

namespace oracle
{
namespace rx
{

namespace
{

constexpr size_t D = 3;
using point3 = std::array<double, D>;

class KMeans // KMeans.hpp
{
  public:
    struct cluster
    {
        point3 centroid{};
        std::vector<std::pair<point3, size_t>> points;
        bool operator<(const cluster &c) const
        {
            return points.size() < c.points.size();
        }
    };
    explicit KMeans(size_t k)
    {
        _clusters.resize(k);
    }
    void reset(size_t k)
    {
        _clusters.clear();
        _clusters.resize(k);
        _initialized = false;
    }
    void add(const point3 &location, size_t value)
    {
        if (_initialized)
        {
            const size_t c = nearest_cluster(location);
            const size_t n = _clusters[c].points.size();
            for (size_t i = 0; i < D; i++)
                _clusters[c].centroid[i] = (location[i] * 1. + _clusters[c].centroid[i] * n) / (n + 1);
            _clusters[c].points.emplace_back(location, value);
        }
        else
            _clusters[0].points.emplace_back(location, value);
    }
    bool iterate()
    {
        if (!_initialized)
            return initialize();
        reassign_centroids();
        reassign_points();
        recalculate_centroids();
        std::sort(_clusters.begin(), _clusters.end());
        return true;
    }
    const std::vector<cluster> &getClusters() const
    {
        return _clusters;
    }

  private:
    static double distance_sq(const point3 &a, const point3 &b)
    {
        double dist = 0;
        for (size_t i = 0; i < D; i++)
            dist += (a[i] - b[i]) * (a[i] - b[i]);
        return dist;
    }
    std::vector<std::pair<point3, size_t>> collect_all_points()
    {
        std::vector<std::pair<point3, size_t>> points;
        for (auto &c : _clusters)
        {
            points.insert(points.end(), c.points.begin(), c.points.end());
            c.points.clear();
        }
        return points;
    }
    bool initialize()
    {
        if (_clusters[0].points.size() < _clusters.size())
            return false;
        auto points = collect_all_points();
        std::vector<size_t> seeds{0};
        std::vector<double> min_dists(points.size(), std::numeric_limits<double>::max());
        for (size_t k = 1; k < _clusters.size(); ++k)
        {
            const point3 &last = points[seeds.back()].first;
            size_t furthest = 0;
            double max_min = -1.0;
            for (size_t i = 0; i < points.size(); ++i)
            {
                min_dists[i] = std::min(min_dists[i], distance_sq(points[i].first, last));
                if (min_dists[i] > max_min)
                {
                    max_min = min_dists[i];
                    furthest = i;
                }
            }
            seeds.push_back(furthest);
        }
        for (size_t k = 0; k < _clusters.size(); ++k)
            _clusters[k].centroid = points[seeds[k]].first;
        for (const auto &p : points)
        {
            size_t best = 0;
            double min_dist = std::numeric_limits<double>::infinity();
            for (size_t k = 0; k < _clusters.size(); ++k)
            {
                const double d = distance_sq(p.first, _clusters[k].centroid);
                if (d < min_dist)
                {
                    min_dist = d;
                    best = k;
                }
            }
            _clusters[best].points.push_back(p);
        }
        recalculate_centroids();
        std::sort(_clusters.begin(), _clusters.end());
        _initialized = true;
        return true;
    }
    void recalculate_centroids()
    {
        for (auto &c : _clusters)
        {
            if (c.points.empty())
                continue;
            c.centroid.fill(0);
            for (const auto &p : c.points)
                for (size_t i = 0; i < D; i++)
                    c.centroid[i] += p.first[i];
            for (size_t i = 0; i < D; i++)
                c.centroid[i] /= c.points.size();
        }
    }
    size_t nearest_cluster(const point3 &location)
    {
        size_t idx = 0;
        double nearest = std::numeric_limits<double>::infinity();
        for (size_t i = 0; i < _clusters.size(); i++)
        {
            const double d = distance_sq(location, _clusters[i].centroid);
            if (d < nearest)
            {
                idx = i;
                nearest = d;
            }
        }
        return idx;
    }
    void reassign_centroids()
    {
        size_t n = 0;
        const double ratio = 2.71828;
        for (; n < _clusters.size() / 2; n++)
            if (_clusters[n].points.size() * ratio > _clusters[_clusters.size() - 1 - n].points.size())
                break;
        for (size_t i = 0; i < n; i++)
            for (size_t j = 0; j < D; j++)
            {
                const double sign = (i + j) % 2 == 0 ? 1 : -1;
                _clusters[i].centroid[j] = _clusters[_clusters.size() - 1 - i].centroid[j] * (1 + sign * 1e-9);
            }
    }
    void reassign_points()
    {
        auto points = collect_all_points();
        for (const auto &p : points)
            _clusters[nearest_cluster(p.first)].points.push_back(p);
    }
    std::vector<cluster> _clusters;
    bool _initialized = false;
};

// all eigenpairs of a dense symmetric matrix (row-major n x n, destroyed) by cyclic Jacobi rotations; eigenvalues
// ascending, eigenvectors as columns of V (row-major n x n)
void jacobi_eigen(std::vector<double> &A, int n, std::vector<double> &w, std::vector<double> &V)
{
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++)
        V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++)
    {
        double off = 0;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++)
                off += A[(size_t)p * n + q] * A[(size_t)p * n + q];
        if (off < 1e-26)
            break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++)
            {
                const double apq = A[(size_t)p * n + q];
                if (std::abs(apq) < 1e-300)
                    continue;
                const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++)
                {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++)
                {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++)
                {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; i++)
        order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return A[(size_t)a * n + a] < A[(size_t)b * n + b]; });
    w.resize(n);
    std::vector<double> Vs((size_t)n * n);
    for (int j = 0; j < n; j++)
    {
        w[j] = A[(size_t)order[j] * n + order[j]];
        for (int k = 0; k < n; k++)
            Vs[(size_t)k * n + j] = V[(size_t)k * n + order[j]];
    }
    V.swap(Vs);
}

class SpectralClustering // spectral_cluster.hpp
{
  public:
    explicit SpectralClustering(size_t k) : _kmeans(k)
    {
    }
    void add(const point3 &location, size_t value)
    {
        _items.emplace_back(location, value);
    }
    void addLink(size_t a, size_t b, double w)
    {
        _links.emplace_back(a, b, w);
    }
    bool spectralize()
    {
        if (_kmeans.getClusters().size() <= 1)
            return false;
        const size_t n = _items.size();
        std::unordered_map<size_t, size_t> lookup;
        for (size_t i = 0; i < n; i++)
            lookup.emplace(_items[i].second, i);
        std::vector<double> degree(n, 0.0);
        std::vector<std::unordered_map<size_t, double>> adj(n); // adjacency.setFromTriplets sums duplicates
        for (const auto &l : _links)
        {
            auto a = lookup.find(std::get<0>(l)), b = lookup.find(std::get<1>(l));
            if (a == lookup.end() || b == lookup.end())
                continue;
            const size_t i0 = a->second, i1 = b->second;
            const double w = std::get<2>(l);
            adj[i0][i1] += w;
            adj[i1][i0] += w;
            degree[i0] += w;
            degree[i1] += w;
        }
        for (double d : degree)
            if (d == 0.)
                return false; // isolated nodes
        // connected components (breadth first over the adjacency's non-zeros, column order = ascending index)
        std::vector<std::vector<size_t>> components;
        {
            std::vector<char> visited(n, 0);
            for (size_t start = 0; start < n; start++)
            {
                if (visited[start])
                    continue;
                components.emplace_back();
                std::queue<size_t> q;
                q.push(start);
                visited[start] = 1;
                while (!q.empty())
                {
                    const size_t node = q.front();
                    q.pop();
                    components.back().push_back(node);
                    std::vector<size_t> nb;
                    for (const auto &kv : adj[node])
                        nb.push_back(kv.first);
                    std::sort(nb.begin(), nb.end());
                    for (size_t v : nb)
                        if (!visited[v])
                        {
                            visited[v] = 1;
                            q.push(v);
                        }
                }
            }
        }
        if (components.size() > 1)
            return spectralizeComponents(components);
        // normalised Laplacian I - D^-1/2 A D^-1/2 (Ng, Jordan, Weiss)
        std::vector<double> L(n * n, 0.0), w, V;
        for (size_t i = 0; i < n; i++)
        {
            L[i * n + i] = 1.0;
            for (const auto &kv : adj[i])
                L[i * n + kv.first] -= kv.second / (std::sqrt(degree[i]) * std::sqrt(degree[kv.first]));
        }
        if (n < D + 3)
            return false; // (Spectra needs nev < ncv <= n)
        jacobi_eigen(L, (int)n, w, V);
        for (size_t i = 0; i < n; i++)
        {
            point3 loc = {V[i * n + 3], V[i * n + 2], V[i * n + 1]};
            const double z = loc[0] * loc[0] + loc[1] * loc[1] + loc[2] * loc[2];
            if (z > 0)
                for (double &c : loc)
                    c /= std::sqrt(z);
            _kmeans.add(loc, _items[i].second);
        }
        return true;
    }
    void fallback()
    {
        for (const auto &item : _items)
            _kmeans.add(item.first, item.second);
    }
    void iterate()
    {
        if (!_sub.empty())
        {
            for (auto &s : _sub)
                s.iterate();
            rebuild();
        }
        else
            _kmeans.iterate();
    }
    const std::vector<KMeans::cluster> &getClusters() const
    {
        return _sub.empty() ? _kmeans.getClusters() : _combined;
    }

  private:
    bool spectralizeComponents(const std::vector<std::vector<size_t>> &components)
    {
        const size_t k = _kmeans.getClusters().size();
        std::vector<size_t> alloc(components.size(), 1);
        for (size_t extra = components.size(); extra < k; extra++)
        {
            size_t best = 0;
            double best_ratio = 0;
            for (size_t i = 0; i < components.size(); i++)
            {
                const double ratio = static_cast<double>(components[i].size()) / alloc[i];
                if (ratio > best_ratio)
                {
                    best_ratio = ratio;
                    best = i;
                }
            }
            alloc[best]++;
        }
        for (size_t c = 0; c < components.size(); c++)
        {
            std::unordered_set<size_t> ids;
            for (size_t idx : components[c])
                ids.insert(_items[idx].second);
            _sub.emplace_back(alloc[c]);
            auto &sub = _sub.back();
            for (size_t idx : components[c])
                sub.add(_items[idx].first, _items[idx].second);
            for (const auto &l : _links)
                if (ids.count(std::get<0>(l)) && ids.count(std::get<1>(l)))
                    sub.addLink(std::get<0>(l), std::get<1>(l), std::get<2>(l));
            const bool ok = (alloc[c] > 1) && sub.spectralize();
            if (!ok)
                sub.fallback();
        }
        rebuild();
        return true;
    }
    void rebuild()
    {
        _combined.clear();
        for (const auto &s : _sub)
            for (const auto &c : s.getClusters())
                _combined.push_back(c);
        std::sort(_combined.begin(), _combined.end());
    }
    std::vector<std::pair<point3, size_t>> _items;
    std::vector<std::tuple<size_t, size_t, double>> _links;
    KMeans _kmeans;
    std::vector<SpectralClustering> _sub;
    std::vector<KMeans::cluster> _combined;
};

} // namespace

// RelaxStage::init (relax_stage.cpp:28-112): the groups' primary node ids, largest group first, and the context depth
std::vector<std::vector<size_t>> relax_stage_groups(const MeasurementGraph &graph, const std::vector<size_t> &node_ids,
                                                    bool relax_all, bool disable_parallelism, uint32_t options,
                                                    size_t *graph_connection_depth)
{
    std::vector<size_t> ids = node_ids;
    if (relax_all)
    {
        ids.clear();
        for (size_t i = 0; i < graph.nodes.size(); i++)
            ids.push_back(i);
    }
    const bool global_params = (options & (FOCAL_LENGTH | PRINCIPAL_POINT | LENS_DISTORTIONS_RADIAL | LENS_DISTORTIONS_TANGENTIAL)) != 0;
    const int optimal = global_params ? 150 : 50;
    const size_t num_groups =
        disable_parallelism ? 1 : std::max<size_t>(1, static_cast<size_t>(std::floor(ids.size() / optimal)));
    SpectralClustering k(num_groups);
    for (size_t id : ids)
    {
        const Vec3 &p = graph.nodes[id].position;
        k.add({p.x, p.y, p.z}, id);
    }
    if (num_groups > 1)
    {
        for (size_t id : ids)
        {
            k.addLink(id, id, 0.1);
            for (size_t e : graph.nodes[id].edges)
                k.addLink(graph.edges[e].source, graph.edges[e].dest, 1);
        }
        if (!k.spectralize())
            k.fallback();
        for (int i = 0; i < 10; i++)
            k.iterate();
    }
    else
        k.fallback();
    *graph_connection_depth = num_groups > 1 ? 0 : 2;
    const auto &clusters = k.getClusters();
    std::vector<std::vector<size_t>> groups;
    for (auto it = clusters.rbegin(); it != clusters.rend(); ++it)
    {
        std::vector<size_t> g;
        for (const auto &p : it->points)
            g.push_back(p.second);
        groups.push_back(std::move(g));
    }
    return groups;
}

} // namespace rx
} // namespace oracle

extern "C"
{
// KMeans<size_t, 3> as oracle/_ref's ref_kmeans3 drives the reference's header
void ocx_kmeans3(const double *xyz, size_t n, size_t k, int iterations, uint64_t *assignment, double *centroids, uint64_t *sizes)
{
    oracle::rx::KMeans km(k);
    for (size_t i = 0; i < n; i++)
        km.add({xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}, i);
    for (int i = 0; i < iterations; i++)
        km.iterate();
    const auto &cl = km.getClusters();
    for (size_t c = 0; c < cl.size(); c++)
    {
        for (int d = 0; d < 3; d++)
            centroids[3 * c + d] = cl[c].centroid[d];
        sizes[c] = cl[c].points.size();
        for (const auto &p : cl[c].points)
            assignment[p.second] = c;
    }
}
}
