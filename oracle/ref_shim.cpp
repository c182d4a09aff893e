// ORACLE — test infrastructure only.  oracle/_ref/libref.so
//
// Thin C ABI over the std-only headers of the reference that compile in this image without any
// stand-ins (included from /root/reference at build time, never copied into this repo):
//   external/jk-tree/include/jk/KDTree.h, external/unordered_dense/include/ankerl/unordered_dense.h,
//   include/opencalibration/relax/grid_filter.hpp, types/union_find.hpp, geometry/KMeans.hpp,
//   combinatorics/interleave.hpp.  It lets the tests check the restated nearest-neighbour logic
// (oracle/match.cpp hash grid; the pair selection of link_stage.cpp:22-38) against the reference's
// own KD-tree, including its tie behaviour.
#include <ankerl/unordered_dense.h>
#include <jk/KDTree.h>
#include <opencalibration/combinatorics/interleave.hpp>
#include <opencalibration/geometry/KMeans.hpp>
#include <opencalibration/relax/grid_filter.hpp>
#include <opencalibration/types/hilbert.hpp>
#include <opencalibration/types/union_find.hpp>

#include <algorithm>
#include <array>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <vector>

extern "C"
{

// Same driver loop as src/match/match_features.cpp:8-52 but on the reference's KDTree<size_t,2,8>.
size_t ref_subsample_kdtree(const double *loc, const float *strength, size_t n, double spacing, size_t count,
                            uint64_t *out)
{
    if (count == 0)
        count = n;
    if (count == 0)
        return 0;
    std::vector<size_t> sorted(count);
    for (size_t i = 0; i < count; i++)
        sorted[i] = i;
    std::sort(sorted.begin(), sorted.end(), [&](size_t a, size_t b) { return strength[a] > strength[b]; });
    jk::tree::KDTree<size_t, 2, 8> tree;
    size_t kept = 0;
    for (size_t idx : sorted)
    {
        const std::array<double, 2> p{loc[2 * idx], loc[2 * idx + 1]};
        bool accept = true;
        if (tree.size() != 0)
        {
            auto searcher = tree.searcher();
            const auto &nn = searcher.search(p, std::numeric_limits<double>::infinity(), 1);
            accept = nn[0].distance > spacing * spacing;
        }
        if (accept)
        {
            tree.addPoint(p, kept);
            out[kept++] = idx;
        }
    }
    return kept;
}

// kNN as LoadStage::finalize (load_stage.cpp:102-103: addPoint in node order) + LinkStage::init
// (link_stage.cpp:26: searchKnn(position, k)) use it.  out: n x k payloads (SIZE_MAX padded).
void ref_knn(const double *xy, size_t n, size_t k, uint64_t *out)
{
    jk::tree::KDTree<size_t, 2> tree;
    for (size_t i = 0; i < n; i++)
        tree.addPoint({xy[2 * i], xy[2 * i + 1]}, i);
    for (size_t i = 0; i < n; i++)
    {
        auto knn = tree.searchKnn({xy[2 * i], xy[2 * i + 1]}, k);
        for (size_t j = 0; j < k; j++)
            out[i * k + j] = j < knn.size() ? knn[j].payload : UINT64_MAX;
    }
}


// GridFilter<size_t> (grid_filter.hpp:16-62): measurements added in the given order; out_best = 1 for the values left in
// getBestMeasurementsPerCell().  Values are 0..n-1.
void ref_grid_filter(const double *xy, const double *score, size_t n, double resolution, uint8_t *out_best)
{
    opencalibration::GridFilter<size_t> f;
    f.setResolution(resolution);
    for (size_t i = 0; i < n; i++)
        f.addMeasurement(xy[2 * i], xy[2 * i + 1], score[i], i);
    for (size_t i = 0; i < n; i++)
        out_best[i] = 0;
    for (size_t v : f.getBestMeasurementsPerCell())
        out_best[v] = 1;
}
// the same with caller-chosen values (a value may be added to several cells: the set semantics of _best)
void ref_grid_filter_values(const double *xy, const double *score, const uint64_t *value, size_t n, double resolution,
                            uint64_t *out_values, size_t *n_out)
{
    opencalibration::GridFilter<size_t> f;
    f.setResolution(resolution);
    for (size_t i = 0; i < n; i++)
        f.addMeasurement(xy[2 * i], xy[2 * i + 1], score[i], (size_t)value[i]);
    size_t k = 0;
    for (size_t v : f.getBestMeasurementsPerCell())
        out_values[k++] = v;
    *n_out = k;
}
uint64_t ref_grid_cell_key(int i, int j)
{
    return opencalibration::gridCellKey(i, j);
}

// UnionFind (union_find.hpp): unite the pairs in order, then roots[i] = find(i)
void ref_union_find(size_t n, const uint64_t *pairs, size_t n_pairs, uint64_t *roots)
{
    opencalibration::UnionFind uf(n);
    for (size_t i = 0; i < n_pairs; i++)
        uf.unite(pairs[2 * i], pairs[2 * i + 1]);
    for (size_t i = 0; i < n; i++)
        roots[i] = uf.find(i);
}

// KMeans<size_t, 3> (KMeans.hpp): add every point (values 0..n-1), `iterations` x iterate(); out: per point the index of
// its cluster in getClusters() order, centroids k x 3, sizes k
void ref_kmeans3(const double *xyz, size_t n, size_t k, int iterations, uint64_t *assignment, double *centroids,
                 uint64_t *sizes)
{
    opencalibration::KMeans<size_t, 3> km(k);
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

// interleave (interleave.hpp) of up to three index lists
size_t ref_interleave3(const uint64_t *a, size_t na, const uint64_t *b, size_t nb, const uint64_t *c, size_t nc,
                       int full_dispersal, uint64_t *out)
{
    std::vector<uint64_t> va(a, a + na), vb(b, b + nb), vc(c, c + nc);
    const std::vector<uint64_t> r =
        opencalibration::interleave<std::vector<uint64_t>>({std::ref(va), std::ref(vb), std::ref(vc)}, full_dispersal != 0);
    for (size_t i = 0; i < r.size(); i++)
        out[i] = r[i];
    return r.size();
}

// the Hilbert index of densifyMesh's feature walk (types/hilbert.hpp)
uint32_t ref_hilbert_xy2d(int order, int x, int y)
{
    return opencalibration::xy2d(order, x, y);
}

// the 3-D nearest-camera query of densifyMesh (dense_stereo.cpp:104-109,212-213): payloads of the k nearest points
size_t ref_knn3(const double *xyz, size_t n, const double *query3, size_t k, uint64_t *out)
{
    jk::tree::KDTree<size_t, 3, 8> tree;
    for (size_t i = 0; i < n; i++)
        tree.addPoint({xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}, i);
    auto searcher = tree.searcher();
    const auto &res = searcher.search({query3[0], query3[1], query3[2]}, std::numeric_limits<double>::max(), k);
    for (size_t i = 0; i < res.size(); i++)
        out[i] = res[i].payload;
    return res.size();
}

// the disc query around a predicted pixel (dense_stereo.cpp:250-252): payloads inside, in the tree's order
size_t ref_ball2(const double *xy, size_t n, const double *query2, double radius_sq, uint64_t *out)
{
    jk::tree::KDTree<size_t, 2, 8> tree;
    for (size_t i = 0; i < n; i++)
        tree.addPoint({xy[2 * i], xy[2 * i + 1]}, i);
    auto searcher = tree.searcher();
    const auto &res = searcher.search({query2[0], query2[1]}, radius_sq, std::numeric_limits<size_t>::max());
    for (size_t i = 0; i < res.size(); i++)
        out[i] = res[i].payload;
    return res.size();
}

// ankerl::unordered_dense::set<size_t>: the iteration order after a sequence of inserts (op >= 0: insert op) and erases
// (op < 0: erase -op - 1) - what the reference's mesh (DirectedGraph's edge map, a vertex's edge set) iterates in
size_t ref_dense_set_order(const int64_t *ops, size_t n, uint64_t *out)
{
    ankerl::unordered_dense::set<size_t> s;
    for (size_t i = 0; i < n; i++)
    {
        if (ops[i] >= 0)
            s.insert((size_t)ops[i]);
        else
            s.erase((size_t)(-ops[i] - 1));
    }
    size_t k = 0;
    for (size_t v : s)
        out[k++] = v;
    return k;
}

} // extern "C"
