// ORACLE — test infrastructure only.  oracle/_ref/libref.so
//
// Thin C ABI over the std-only header of the reference that compiles in this image without any
// stand-ins: external/jk-tree/include/jk/KDTree.h (included from /root/reference at build time, never
// copied into this repo).  It lets the tests check the restated nearest-neighbour logic
// (oracle/match.cpp hash grid; the pair selection of link_stage.cpp:22-38) against the reference's
// own KD-tree, including its tie behaviour.
#include <jk/KDTree.h>

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

} // extern "C"
