#include "dense_stereo.hpp"

#include "invert_distortion.hpp"
#include "relax_util.hpp"
#include "triangle_walker.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <numeric>

namespace opencalibration_amd
{

using namespace relax_detail;

namespace
{

constexpr double SEARCH_RADIUS_PIXELS = 150.0; // dense_stereo.cpp:51-55
constexpr double RATIO_THRESHOLD = 0.85;
constexpr int MAX_CANDIDATE_IMAGES = 10;
constexpr double MAX_ABSOLUTE_DESCRIPTOR_DISTANCE = 0.35;
constexpr double MAX_REPROJECTION_ERROR_PIXELS = 8.0;
constexpr double CELL_SIZE = SEARCH_RADIUS_PIXELS + 1.0; // grid of the device index: a disc touches at most 3 x 3 cells

double seconds_since(const std::chrono::steady_clock::time_point &t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// camera_tree.searcher().search(point, max, k) (jk::KDTree, 3-D): the k cameras nearest to a point, nearest first.
// Exact: cameras binned on an x-y grid, rings of cells around the query until the k-th best squared distance is below
// the squared x-y distance every camera of an unvisited ring must have.  Ties (measure zero) go to the camera added first.
class CameraGrid
{
  public:
    void build(const std::vector<std::array<double, 3>> &positions)
    {
        _pos = &positions;
        const size_t n = positions.size();
        _x0 = _y0 = INFINITY;
        double x1 = -INFINITY, y1 = -INFINITY;
        for (const auto &p : positions)
        {
            _x0 = std::min(_x0, p[0]), _y0 = std::min(_y0, p[1]);
            x1 = std::max(x1, p[0]), y1 = std::max(y1, p[1]);
        }
        const double area = std::max((x1 - _x0) * (y1 - _y0), 1e-12);
        _cell = std::max(std::max(std::sqrt(area / std::max<size_t>(n, 1)), std::max(x1 - _x0, y1 - _y0) / 4000.0), 1e-6);
        _nx = (long)((x1 - _x0) / _cell) + 1;
        _ny = (long)((y1 - _y0) / _cell) + 1;
        _start.assign((size_t)_nx * _ny + 1, 0);
        std::vector<uint32_t> cell_of(n);
        for (size_t i = 0; i < n; i++)
        {
            const long cx = std::min(_nx - 1, (long)((positions[i][0] - _x0) / _cell)), cy = std::min(_ny - 1, (long)((positions[i][1] - _y0) / _cell));
            cell_of[i] = (uint32_t)(cy * _nx + cx);
            _start[cell_of[i] + 1]++;
        }
        for (size_t c = 0; c < (size_t)_nx * _ny; c++)
            _start[c + 1] += _start[c];
        _items.resize(n);
        std::vector<uint32_t> fill(_start.begin(), _start.end() - 1);
        for (size_t i = 0; i < n; i++)
            _items[fill[cell_of[i]]++] = (uint32_t)i;
    }
    // out: up to k (squared distance, camera) pairs, ascending
    void nearest(const v3 &p, size_t k, std::vector<std::pair<double, uint32_t>> &out) const
    {
        out.clear();
        const long cx = (long)std::floor((p.x - _x0) / _cell), cy = (long)std::floor((p.y - _y0) / _cell);
        const long far = std::max(std::max(std::labs(cx), std::labs(cx - (_nx - 1))), std::max(std::labs(cy), std::labs(cy - (_ny - 1))));
        for (long r = 0; r <= far; r++)
        {
            auto visit = [&](long x, long y) {
                if (x < 0 || y < 0 || x >= _nx || y >= _ny)
                    return;
                const size_t c = (size_t)(y * _nx + x);
                for (uint32_t it = _start[c]; it < _start[c + 1]; it++)
                {
                    const uint32_t cam = _items[it];
                    const auto &q = (*_pos)[cam];
                    const double dx = q[0] - p.x, dy = q[1] - p.y, dz = q[2] - p.z;
                    const std::pair<double, uint32_t> e{dx * dx + dy * dy + dz * dz, cam};
                    if (out.size() < k)
                        out.insert(std::upper_bound(out.begin(), out.end(), e), e);
                    else if (e < out.back())
                    {
                        out.pop_back();
                        out.insert(std::upper_bound(out.begin(), out.end(), e), e);
                    }
                }
            };
            if (r == 0)
                visit(cx, cy);
            else
            {
                for (long x = std::max(0L, cx - r); x <= std::min(_nx - 1, cx + r); x++) // the ring's two rows
                {
                    visit(x, cy - r);
                    visit(x, cy + r);
                }
                for (long y = std::max(0L, cy - r + 1); y <= std::min(_ny - 1, cy + r - 1); y++) // and two columns
                {
                    visit(cx - r, y);
                    visit(cx + r, y);
                }
            }
            if (out.size() >= k && out.back().first < (double)r * _cell * (double)r * _cell)
                break;
        }
    }

  private:
    const std::vector<std::array<double, 3>> *_pos = nullptr;
    double _x0 = 0, _y0 = 0, _cell = 1;
    long _nx = 1, _ny = 1;
    std::vector<uint32_t> _start, _items;
};

struct DenseImage
{
    const image *img = nullptr;
    size_t node_index = 0;   // place in the graph's node order
    size_t offset = 0;       // first measurement id (reference numbering: image offset + dense feature number)
    size_t n_dense = 0;
    uint64_t feat_base = 0;  // first position in the device index
    std::vector<uint32_t> sorted_to_dense, dense_to_sorted; // cell order <-> dense feature number
    double q_inv[4];         // orientation.inverse()
};

inline void project(const v3 &point, const DenseImage &im, double pixel[2]) // image_from_3d(point, model, position, orientation)
{
    const v3 rel{point.x - im.img->position[0], point.y - im.img->position[1], point.z - im.img->position[2]};
    const v3 r = rotate(im.q_inv, rel);
    const double ray[3] = {r.x, r.y, r.z};
    image_from_3d(ray, *im.img->model, pixel);
}

} // namespace

uint32_t hilbert_xy2d(int order, int x, int y)
{
    uint32_t d = 0;
    for (int s = order / 2; s > 0; s /= 2)
    {
        const int rx = (x & s) > 0 ? 1 : 0, ry = (y & s) > 0 ? 1 : 0;
        d += (uint32_t)(s * s * ((3 * rx) ^ ry));
        if (ry == 0)
        {
            if (rx == 1)
            {
                x = s - 1 - x;
                y = s - 1 - y;
            }
            std::swap(x, y);
        }
    }
    return d;
}

bool densifyMesh(ochip_ctx *ctx, const MeasurementGraph &graph, std::vector<surface_model> &surfaces, DenseStats *stats, std::string *error,
                 std::vector<std::pair<size_t, size_t>> *matches_out)
{
    DenseStats st;
    auto finish = [&](bool ok) {
        if (stats)
            *stats = st;
        return ok;
    };
    if (surfaces.empty())
        return finish(true);
    // images with dense features, a camera model and a finite pose (:78-88), in the graph's order
    std::vector<DenseImage> images;
    for (size_t ni = 0; ni < graph.size_nodes(); ni++)
    {
        const image &img = graph.nodes()[ni].payload;
        bool nan = false;
        for (int k = 0; k < 3; k++)
            nan = nan || std::isnan(img.position[k]);
        for (int k = 0; k < 4; k++)
            nan = nan || std::isnan(img.orientation[k]);
        if (img.features.size() > img.num_sparse_features && img.model && !nan)
        {
            DenseImage d;
            d.img = &img;
            d.node_index = ni;
            d.n_dense = img.features.size() - img.num_sparse_features;
            const double *q = img.orientation;
            const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
            if (n2 > 0)
                d.q_inv[0] = -q[0] / n2, d.q_inv[1] = -q[1] / n2, d.q_inv[2] = -q[2] / n2, d.q_inv[3] = q[3] / n2;
            else
                d.q_inv[0] = d.q_inv[1] = d.q_inv[2] = d.q_inv[3] = 0;
            images.push_back(std::move(d));
        }
    }
    if (images.empty())
        return finish(true);
    st.images = images.size();
    const size_t n_img = images.size();
    size_t total = 0;
    for (DenseImage &d : images)
    {
        d.offset = total;
        d.feat_base = total;
        total += d.n_dense;
    }
    st.dense_features = total;
    if (total >= (1ull << 32))
    {
        if (error)
            *error = "densifyMesh: more than 2^32 dense features";
        return finish(false);
    }

    // ---- the device index: every image's dense features sorted by grid cell
    auto t0 = std::chrono::steady_clock::now();
    std::vector<uint64_t> feat_off(n_img + 1), cell_off(n_img + 1);
    std::vector<int32_t> grid2(2 * n_img);
    std::vector<double> origin2(2 * n_img);
    std::vector<uint64_t> desc8(8 * total);
    std::vector<double> loc2(2 * total);
    for (size_t i = 0; i < n_img; i++)
    {
        const DenseImage &d = images[i];
        const image &img = *d.img;
        double x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
        for (size_t k = img.num_sparse_features; k < img.features.size(); k++)
        {
            x0 = std::min(x0, img.features[k].location[0]), y0 = std::min(y0, img.features[k].location[1]);
            x1 = std::max(x1, img.features[k].location[0]), y1 = std::max(y1, img.features[k].location[1]);
        }
        origin2[2 * i] = std::floor(x0), origin2[2 * i + 1] = std::floor(y0);
        grid2[2 * i] = (int32_t)std::floor((x1 - origin2[2 * i]) / CELL_SIZE) + 1;
        grid2[2 * i + 1] = (int32_t)std::floor((y1 - origin2[2 * i + 1]) / CELL_SIZE) + 1;
        feat_off[i] = d.feat_base;
        cell_off[i + 1] = (uint64_t)grid2[2 * i] * grid2[2 * i + 1] + 1; // sizes first, offsets below
    }
    feat_off[n_img] = total;
    cell_off[0] = 0;
    for (size_t i = 0; i < n_img; i++)
        cell_off[i + 1] += cell_off[i];
    std::vector<uint32_t> cell_start(cell_off[n_img]);
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t i = 0; i < n_img; i++)
    {
        DenseImage &d = images[i];
        const image &img = *d.img;
        const int ncx = grid2[2 * i], ncy = grid2[2 * i + 1];
        uint32_t *cs = cell_start.data() + cell_off[i];
        std::fill(cs, cs + (size_t)ncx * ncy + 1, 0u);
        std::vector<uint32_t> cell_of(d.n_dense);
        for (size_t k = 0; k < d.n_dense; k++)
        {
            const double *l = img.features[img.num_sparse_features + k].location;
            const int cx = (int)std::floor((l[0] - origin2[2 * i]) / CELL_SIZE), cy = (int)std::floor((l[1] - origin2[2 * i + 1]) / CELL_SIZE);
            cell_of[k] = (uint32_t)(cy * ncx + cx);
            cs[cell_of[k] + 1]++;
        }
        for (size_t c = 0; c < (size_t)ncx * ncy; c++)
            cs[c + 1] += cs[c];
        std::vector<uint32_t> fill(cs, cs + (size_t)ncx * ncy);
        d.sorted_to_dense.resize(d.n_dense);
        d.dense_to_sorted.resize(d.n_dense);
        for (size_t k = 0; k < d.n_dense; k++)
        {
            const uint32_t pos = fill[cell_of[k]]++;
            d.sorted_to_dense[pos] = (uint32_t)k;
            d.dense_to_sorted[k] = pos;
            const feature_2d &f = img.features[img.num_sparse_features + k];
            std::memcpy(&desc8[8 * (d.feat_base + pos)], f.descriptor, 64);
            loc2[2 * (d.feat_base + pos)] = f.location[0];
            loc2[2 * (d.feat_base + pos) + 1] = f.location[1];
        }
    }
    ochip_dense_index *index = nullptr;
    if (ochip_dense_index_create(ctx, (uint32_t)n_img, feat_off.data(), desc8.data(), loc2.data(), cell_off.data(), cell_start.data(),
                                 grid2.data(), origin2.data(), CELL_SIZE, &index) != OCHIP_OK)
    {
        if (error)
            *error = std::string("ochip_dense_index_create: ") + ochip_last_error(ctx);
        return finish(false);
    }
    std::vector<uint64_t>().swap(desc8);
    std::vector<double>().swap(loc2);
    st.index_seconds = seconds_since(t0);

    std::vector<std::array<double, 3>> cam_pos(n_img);
    for (size_t i = 0; i < n_img; i++)
        cam_pos[i] = {images[i].img->position[0], images[i].img->position[1], images[i].img->position[2]};
    CameraGrid camera_grid;
    camera_grid.build(cam_pos);

    const MeshGraph &mesh = surfaces[0].mesh;
    // union-find over the measurements, lock-free: the larger root goes under the smaller one, so a component's root is its
    // smallest member and the partition does not depend on the order of the unions (the reference unites under a mutex
    // in whatever order its OpenMP threads finish; only the partition reaches the result)
    std::unique_ptr<std::atomic<uint32_t>[]> parent(new std::atomic<uint32_t>[std::max<size_t>(total, 1)]);
    std::unique_ptr<std::atomic<uint8_t>[]> matched(new std::atomic<uint8_t>[std::max<size_t>(total, 1)]);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < total; i++)
    {
        parent[i].store((uint32_t)i, std::memory_order_relaxed);
        matched[i].store(0, std::memory_order_relaxed);
    }
    auto find = [&](uint32_t x) {
        while (true)
        {
            uint32_t p = parent[x].load(std::memory_order_relaxed);
            if (p == x)
                return x;
            const uint32_t gp = parent[p].load(std::memory_order_relaxed);
            if (gp != p)
                parent[x].compare_exchange_weak(p, gp, std::memory_order_relaxed);
            x = p;
        }
    };
    auto unite = [&](uint32_t a, uint32_t b) {
        matched[a].store(1, std::memory_order_relaxed);
        matched[b].store(1, std::memory_order_relaxed);
        while (true)
        {
            a = find(a);
            b = find(b);
            if (a == b)
                return;
            if (a < b)
                std::swap(a, b);
            uint32_t expected = a;
            if (parent[a].compare_exchange_strong(expected, b, std::memory_order_relaxed))
                return;
        }
    };
    bool device_failed = false;
    // batches of source images: rays, mesh intersections and predictions on the host threads, the descriptor search of
    // the whole batch in one device call, the accept decisions and the union-find on the host
    const size_t BATCH = 64;
    for (size_t b0 = 0; b0 < n_img && !device_failed; b0 += BATCH)
    {
        const size_t b1 = std::min(n_img, b0 + BATCH);
        auto t1 = std::chrono::steady_clock::now();
        std::vector<std::vector<ochip_dense_query>> per_image(b1 - b0);
#pragma omp parallel for schedule(dynamic, 1)
        for (size_t si = b0; si < b1; si++)
        {
            const DenseImage &src = images[si];
            const image &img = *src.img;
            TriangleWalker walker;
            if (!walker.init(mesh))
                continue;
            // hilbertFeatureOrder (:24-49)
            const int w = (int)img.model->pixels_cols, h = (int)img.model->pixels_rows;
            int order = 1;
            while (order < std::max(w, h))
                order *= 2;
            std::vector<std::pair<uint32_t, size_t>> indexed(src.n_dense);
            for (size_t k = 0; k < src.n_dense; k++)
            {
                const double *l = img.features[img.num_sparse_features + k].location;
                indexed[k] = {hilbert_xy2d(order, std::clamp((int)l[0], 0, w - 1), std::clamp((int)l[1], 0, h - 1)), k};
            }
            std::sort(indexed.begin(), indexed.end());
            std::vector<ochip_dense_query> &queries = per_image[si - b0];
            queries.reserve(src.n_dense * MAX_CANDIDATE_IMAGES);
            std::vector<std::pair<double, uint32_t>> cams;
            const v3 origin{img.position[0], img.position[1], img.position[2]};
            for (const auto &entry : indexed)
            {
                const size_t k = entry.second;
                double ray[3];
                image_to_3d(img.features[img.num_sparse_features + k].location, *img.model, ray);
                const v3 dir = rotate(img.orientation, v3{ray[0], ray[1], ray[2]});
                if (walker.find(dir, origin) != TriangleWalker::INTERSECTION)
                    continue;
                const v3 pt3d = walker.hit;
                camera_grid.nearest(pt3d, MAX_CANDIDATE_IMAGES + 1, cams);
                for (const auto &cand : cams)
                {
                    if (cand.second == si)
                        continue;
                    const DenseImage &ci = images[cand.second];
                    double px[2];
                    project(pt3d, ci, px);
                    if (px[0] < 0 || px[0] >= (double)ci.img->model->pixels_cols || px[1] < 0 || px[1] >= (double)ci.img->model->pixels_rows)
                        continue;
                    ochip_dense_query q;
                    q.src_feature = (uint32_t)(src.feat_base + src.dense_to_sorted[k]);
                    q.cand_image = cand.second;
                    q.px = px[0];
                    q.py = px[1];
                    queries.push_back(q);
                }
            }
        }
        std::vector<ochip_dense_query> queries;
        {
            size_t n = 0;
            for (const auto &v : per_image)
                n += v.size();
            queries.reserve(n);
            for (const auto &v : per_image)
                queries.insert(queries.end(), v.begin(), v.end());
        }
        st.rays_seconds += seconds_since(t1);
        st.queries += queries.size();
        t1 = std::chrono::steady_clock::now();
        std::vector<ochip_dense_result> results(queries.size());
        if (ochip_dense_match(index, queries.data(), queries.size(), SEARCH_RADIUS_PIXELS, results.data()) != OCHIP_OK)
        {
            if (error)
                *error = std::string("ochip_dense_match: ") + ochip_last_error(ctx);
            device_failed = true;
            break;
        }
        st.device_seconds += seconds_since(t1);
        t1 = std::chrono::steady_clock::now();
        // the accept decisions (dense_stereo.cpp:278-283) and the unions, one source image per task
        std::vector<size_t> q_off(per_image.size() + 1, 0);
        for (size_t i = 0; i < per_image.size(); i++)
            q_off[i + 1] = q_off[i] + per_image[i].size();
        std::vector<size_t> accepted(per_image.size(), 0);
        std::vector<std::vector<std::pair<size_t, size_t>>> kept(matches_out ? per_image.size() : 0);
#pragma omp parallel for schedule(dynamic, 1)
        for (size_t k = 0; k < per_image.size(); k++)
        {
            const DenseImage &src = images[b0 + k];
            for (size_t qi = q_off[k]; qi < q_off[k + 1]; qi++)
            {
                const ochip_dense_result &r = results[qi];
                if (r.nearby == 0)
                    continue;
                const double best_dist = r.best_count * (1.0 / feature_2d::DESCRIPTOR_BITS);
                const double second_best_dist =
                    r.second_count == 0xFFFF ? INFINITY : r.second_count * (1.0 / feature_2d::DESCRIPTOR_BITS);
                const bool good_match =
                    r.nearby >= 2 ? best_dist < RATIO_THRESHOLD * second_best_dist : best_dist < MAX_ABSOLUTE_DESCRIPTOR_DISTANCE;
                if (!good_match)
                    continue;
                const DenseImage &dst = images[queries[qi].cand_image];
                const size_t src_id = src.offset + src.sorted_to_dense[queries[qi].src_feature - src.feat_base];
                const size_t dst_id = dst.offset + dst.sorted_to_dense[r.best_feature];
                if (matches_out)
                    kept[k].emplace_back(src_id, dst_id);
                accepted[k]++;
                unite((uint32_t)src_id, (uint32_t)dst_id);
            }
        }
        for (size_t k = 0; k < per_image.size(); k++)
        {
            st.matches += accepted[k];
            if (matches_out)
                matches_out->insert(matches_out->end(), kept[k].begin(), kept[k].end());
        }
        st.tracks_seconds += seconds_since(t1);
    }
    ochip_dense_index_destroy(index);
    if (device_failed)
        return finish(false);

    // ---- tracks, in the order of their smallest member (:299-340), triangulated from their first two rays
    auto t2 = std::chrono::steady_clock::now();
    std::vector<std::vector<size_t>> multi_tracks;
    {
        // a matched measurement's component: tracks in the order of their smallest member (= their root), members ascending
        std::vector<uint32_t> root(total);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < total; i++)
            root[i] = matched[i].load(std::memory_order_relaxed) ? find((uint32_t)i) : UINT32_MAX;
        std::vector<size_t> track_of_root(total, (size_t)-1);
        for (size_t i = 0; i < total; i++)
        {
            if (root[i] == UINT32_MAX) // is_singleton
                continue;
            size_t &t = track_of_root[root[i]];
            if (t == (size_t)-1)
            {
                t = multi_tracks.size();
                multi_tracks.emplace_back();
            }
            multi_tracks[t].push_back(i);
        }
    }
    st.tracks = multi_tracks.size();
    auto image_of // measurement id -> image: offsets ascend
         = [&](size_t id) {
        size_t lo = 0, hi = n_img - 1;
        while (lo < hi)
        {
            const size_t mid = (lo + hi + 1) / 2;
            if (images[mid].offset <= id)
                lo = mid;
            else
                hi = mid - 1;
        }
        return lo;
    };
    const double max_err_sq = MAX_REPROJECTION_ERROR_PIXELS * MAX_REPROJECTION_ERROR_PIXELS;
    std::vector<std::array<double, 3>> track_results(multi_tracks.size());
    std::vector<char> track_valid(multi_tracks.size(), 0);
#pragma omp parallel for schedule(dynamic, 64)
    for (size_t ti = 0; ti < multi_tracks.size(); ti++)
    {
        const auto &ids = multi_tracks[ti];
        if (ids.size() < 2)
            continue;
        struct RayMeasurement
        {
            v3 dir, origin;
            const double *pixel;
            const DenseImage *im;
        };
        std::vector<RayMeasurement> ms;
        ms.reserve(ids.size());
        for (size_t id : ids)
        {
            const DenseImage &im = images[image_of(id)];
            const image &img = *im.img;
            const double *px = img.features[img.num_sparse_features + (id - im.offset)].location;
            double ray[3];
            image_to_3d(px, *img.model, ray);
            ms.push_back({rotate(img.orientation, v3{ray[0], ray[1], ray[2]}), v3{img.position[0], img.position[1], img.position[2]}, px, &im});
        }
        v3 point;
        double err;
        ray_intersection(ms[0].dir, ms[0].origin, ms[1].dir, ms[1].origin, &point, &err); // only the first two rays (:145-161)
        auto finite = [](const v3 &p) { return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z); };
        if (!finite(point) || err < 0)
            continue;
        std::vector<size_t> inliers;
        for (size_t i = 0; i < ms.size(); i++)
        {
            double reproj[2];
            project(point, *ms[i].im, reproj);
            const double ex = reproj[0] - ms[i].pixel[0], ey = reproj[1] - ms[i].pixel[1];
            if (ex * ex + ey * ey <= max_err_sq)
                inliers.push_back(i);
        }
        if (inliers.size() < 2)
            continue;
        if (inliers.size() < ms.size())
        {
            ray_intersection(ms[inliers[0]].dir, ms[inliers[0]].origin, ms[inliers[1]].dir, ms[inliers[1]].origin, &point, &err);
            if (!finite(point) || err < 0)
                continue;
        }
        track_results[ti] = {point.x, point.y, point.z};
        track_valid[ti] = 1;
    }
    point_cloud merged;
    for (size_t ti = 0; ti < multi_tracks.size(); ti++)
        if (track_valid[ti])
            merged.push_back(track_results[ti]);
    st.points = merged.size();
    if (!merged.empty())
        surfaces[0].cloud.push_back(std::move(merged));
    st.tracks_seconds += seconds_since(t2);
    return finish(true);
}

} // namespace opencalibration_amd
