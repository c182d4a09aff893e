// ORACLE — test infrastructure only.  densifyMesh (src/dense/dense_stereo.cpp:66-403) restated: for every dense feature of
// every image the ray through it is intersected with the surface mesh, the 3-D point is projected into the 10 nearest other
// cameras, the dense features of that camera within 150 px of the prediction are compared by descriptor (best < 0.85 x
// second best, or < 0.35 when there is only one), matches are united into tracks, tracks are triangulated from their
// first two rays and filtered by an 8 px reprojection error.
//
// The reference's jk::KDTree queries (nearest cameras in 3-D, features in a disc) are exhaustive scans here: the disc
// query is `squared distance < radius^2` (KDTree.h:398,404), the nearest-camera query returns the closest
// MAX_CANDIDATE_IMAGES + 1 cameras; exact distance ties between cameras (measure zero) go to the camera inserted first.
// The order in which the features of a disc are visited cannot change an accepted match: two candidates that tie for the
// best distance make best == second best and the ratio test fail (dense_stereo.cpp:262-283).
#include "relax_full.hpp"
#include "relax_functors.hpp"

#include <algorithm>
#include <cstring>
#include <numeric>

namespace oracle
{
namespace rx
{
namespace
{

constexpr double SEARCH_RADIUS_PIXELS = 150.0; // dense_stereo.cpp:51-55
constexpr double RATIO_THRESHOLD = 0.85;
constexpr int MAX_CANDIDATE_IMAGES = 10;
constexpr double MAX_ABSOLUTE_DESCRIPTOR_DISTANCE = 0.35;
constexpr double MAX_REPROJECTION_ERROR_PIXELS = 8.0;

uint32_t xy2d(int order, int x, int y) // types/hilbert.hpp:9-28
{
    uint32_t d = 0;
    for (int s = order / 2; s > 0; s /= 2)
    {
        const int rx = (x & s) > 0 ? 1 : 0, ry = (y & s) > 0 ? 1 : 0;
        d += s * s * ((3 * rx) ^ ry);
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

class UnionFind // types/union_find.hpp
{
  public:
    explicit UnionFind(size_t n) : _parent(n), _rank(n, 0)
    {
        std::iota(_parent.begin(), _parent.end(), 0);
    }
    size_t find(size_t x)
    {
        if (_parent[x] != x)
            _parent[x] = find(_parent[x]);
        return _parent[x];
    }
    void unite(size_t a, size_t b)
    {
        a = find(a);
        b = find(b);
        if (a == b)
            return;
        if (_rank[a] < _rank[b])
            std::swap(a, b);
        _parent[b] = a;
        if (_rank[a] == _rank[b])
            _rank[a]++;
    }
    bool is_singleton(size_t x) const
    {
        return _parent[x] == x && _rank[x] == 0;
    }

  private:
    std::vector<size_t> _parent, _rank;
};

Vec3 rotate(const Quat &q, const Vec3 &v)
{
    const double qq[4] = {q.x, q.y, q.z, q.w};
    const V3<double> r = quat_rotate<double>(qq, V3<double>{v.x, v.y, v.z});
    return Vec3{r.x, r.y, r.z};
}
Quat inverse(const Quat &q) // Eigen QuaternionBase::inverse(): conjugate / squaredNorm, zero when the norm is zero
{
    const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    if (n2 > 0)
        return Quat{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
    return Quat{0, 0, 0, 0};
}
// image_from_3d(point, model, camera_location, camera_orientation) (distort_keypoints.hpp:69-76)
Vec2 project(const Vec3 &point, const camera_model &model, const Vec3 &position, const Quat &orientation)
{
    return image_from_3d(rotate(inverse(orientation), point - position), model);
}
std::pair<Vec3, double> rayIntersection2(const Vec3 &d1, const Vec3 &o1, const Vec3 &d2, const Vec3 &o2) // intersection.cpp:116-143
{
    Vec3 res{NAN, NAN, NAN};
    double error = NAN;
    const double n1dn1 = dot(d1, d1), n1dn2 = dot(d1, d2), n2dn2 = dot(d2, d2);
    const double scale_denom = n1dn1 * n2dn2 - n1dn2 * n1dn2;
    if (std::abs(scale_denom) > 1e-9)
    {
        const Vec3 offset = o1 - o2;
        const double offsetdn1 = dot(offset, d1), offsetdn2 = dot(offset, d2);
        const double t = (n1dn2 * offsetdn2 - n2dn2 * offsetdn1) / scale_denom;
        const double s = (n1dn1 * offsetdn2 - n1dn2 * offsetdn1) / scale_denom;
        const Vec3 p1 = o1 + d1 * t, p2 = o2 + d2 * s;
        res = (p1 + p2) * 0.5;
        const Vec3 dd = p1 - p2;
        error = dot(dd, dd) * (t >= 0 && s >= 0 ? 1 : -1);
    }
    return {res, error};
}

struct dense_image
{
    Vec3 position;
    Quat orientation;
    camera_model model;
    const double *loc;      // all features, n x 2
    const uint64_t *desc;   // n x 8
    size_t n_features, num_sparse;
};

double descriptor_distance(const uint64_t *a, const uint64_t *b) // dense_stereo.cpp:57-60
{
    int c = 0;
    for (int w = 0; w < 8; w++)
        c += __builtin_popcountll(a[w] ^ b[w]);
    return c * (1.0 / 486);
}

} // namespace
} // namespace rx
} // namespace oracle

using namespace oracle;
using namespace oracle::rx;

extern "C"
{

// images: every graph node in order (node_ids = the ones with dense features, a model, finite pose: :78-88).
// model10: focal, pp x y, k1 k2 k3, p1 p2, cols, rows per image.  surface: an ocx_surface handle; its mesh is read, the
// merged cloud is appended to it.  Outputs (any may be NULL): matches as pairs of global measurement ids (in the order of
// the image loop, then the Hilbert walk, then the candidates), points n x 3.  counts: {matches, tracks with >= 2
// members, points}.
void ocx_densify(size_t n_images, const double *pos3, const double *ori4, const double *model10, const uint64_t *feat_off,
                 const double *loc, const uint64_t *desc8, const uint64_t *num_sparse, void *surface_handle, uint64_t *match_pairs,
                 size_t match_cap, double *points, size_t points_cap, uint64_t *counts)
{
    surface_model &surface = *(surface_model *)surface_handle;
    counts[0] = counts[1] = counts[2] = 0;
    std::vector<dense_image> images(n_images);
    for (size_t i = 0; i < n_images; i++)
    {
        dense_image &im = images[i];
        im.position = Vec3{pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2]};
        im.orientation = Quat{ori4[4 * i], ori4[4 * i + 1], ori4[4 * i + 2], ori4[4 * i + 3]};
        const double *m = model10 + 10 * i;
        im.model.focal_length_pixels = m[0];
        im.model.principle_point[0] = m[1], im.model.principle_point[1] = m[2];
        for (int k = 0; k < 3; k++)
            im.model.radial_distortion[k] = m[3 + k];
        im.model.tangential_distortion[0] = m[6], im.model.tangential_distortion[1] = m[7];
        im.model.pixels_cols = (size_t)m[8], im.model.pixels_rows = (size_t)m[9];
        im.loc = loc + 2 * feat_off[i];
        im.desc = desc8 + 8 * feat_off[i];
        im.n_features = feat_off[i + 1] - feat_off[i];
        im.num_sparse = num_sparse[i];
    }
    std::vector<size_t> node_ids;
    for (size_t i = 0; i < n_images; i++)
    {
        const dense_image &im = images[i];
        const bool pose_ok = !(std::isnan(im.position.x) || std::isnan(im.position.y) || std::isnan(im.position.z) ||
                               std::isnan(im.orientation.x) || std::isnan(im.orientation.y) || std::isnan(im.orientation.z) ||
                               std::isnan(im.orientation.w));
        if (im.n_features > im.num_sparse && pose_ok)
            node_ids.push_back(i);
    }
    if (node_ids.empty())
        return;

    const MeshGraph &mesh = surface.mesh;
    struct Measurement
    {
        size_t node_id, feat_idx;
    };
    std::vector<Measurement> id_to_measurement;
    std::vector<size_t> node_offset(n_images, 0);
    {
        size_t total = 0;
        for (size_t nid : node_ids)
        {
            node_offset[nid] = total;
            total += images[nid].n_features - images[nid].num_sparse;
        }
        id_to_measurement.resize(total);
        for (size_t nid : node_ids)
            for (size_t i = 0; i < images[nid].n_features - images[nid].num_sparse; i++)
                id_to_measurement[node_offset[nid] + i] = {nid, images[nid].num_sparse + i};
    }
    auto measurementId = [&](size_t nid, size_t feat_idx) { return node_offset[nid] + feat_idx - images[nid].num_sparse; };

    UnionFind uf(id_to_measurement.size());
    for (size_t src_nid : node_ids)
    {
        const dense_image &src = images[src_nid];
        MeshIntersectionSearcher searcher;
        if (!searcher.init(mesh))
            continue;
        // hilbertFeatureOrder (:24-49)
        std::vector<size_t> order;
        {
            const int w = (int)src.model.pixels_cols, h = (int)src.model.pixels_rows;
            int hil = 1;
            while (hil < std::max(w, h))
                hil *= 2;
            std::vector<std::pair<uint32_t, size_t>> indexed;
            for (size_t i = src.num_sparse; i < src.n_features; i++)
            {
                const int x = std::clamp((int)src.loc[2 * i], 0, w - 1), y = std::clamp((int)src.loc[2 * i + 1], 0, h - 1);
                indexed.push_back({xy2d(hil, x, y), i - src.num_sparse});
            }
            std::sort(indexed.begin(), indexed.end());
            for (auto &p : indexed)
                order.push_back(p.second);
        }
        for (size_t fi : order)
        {
            const size_t global_fi = src.num_sparse + fi;
            const Vec3 dir = rotate(src.orientation, image_to_3d(src.loc + 2 * global_fi, src.model));
            const auto &info = searcher.triangleIntersect(dir, src.position);
            if (info.type != MeshIntersectionSearcher::INTERSECTION)
                continue;
            const Vec3 pt3d = info.intersectionLocation;
            const size_t src_id = measurementId(src_nid, global_fi);
            // the MAX_CANDIDATE_IMAGES + 1 cameras nearest to the point, nearest first
            std::vector<std::pair<double, size_t>> cams;
            for (size_t nid : node_ids)
            {
                const Vec3 d = images[nid].position - pt3d;
                cams.push_back({dot(d, d), nid});
            }
            const size_t keep = std::min<size_t>(cams.size(), MAX_CANDIDATE_IMAGES + 1);
            std::partial_sort(cams.begin(), cams.begin() + keep, cams.end());
            for (size_t c = 0; c < keep; c++)
            {
                const size_t cand_nid = cams[c].second;
                if (cand_nid == src_nid)
                    continue;
                const dense_image &cand = images[cand_nid];
                const Vec2 predicted = project(pt3d, cand.model, cand.position, cand.orientation);
                if (predicted.x < 0 || predicted.x >= cand.model.pixels_cols || predicted.y < 0 ||
                    predicted.y >= cand.model.pixels_rows)
                    continue;
                double best_dist = INFINITY, second_best_dist = INFINITY;
                size_t best_feat_idx = 0, nearby = 0;
                for (size_t k = cand.num_sparse; k < cand.n_features; k++)
                {
                    const double dx = cand.loc[2 * k] - predicted.x, dy = cand.loc[2 * k + 1] - predicted.y;
                    if (!(dx * dx + dy * dy < SEARCH_RADIUS_PIXELS * SEARCH_RADIUS_PIXELS))
                        continue;
                    nearby++;
                    const double d = descriptor_distance(src.desc + 8 * global_fi, cand.desc + 8 * k);
                    if (d < second_best_dist)
                    {
                        if (d < best_dist)
                        {
                            second_best_dist = best_dist;
                            best_dist = d;
                            best_feat_idx = k;
                        }
                        else
                            second_best_dist = d;
                    }
                }
                if (nearby == 0)
                    continue;
                const bool good_match =
                    nearby >= 2 ? best_dist < RATIO_THRESHOLD * second_best_dist : best_dist < MAX_ABSOLUTE_DESCRIPTOR_DISTANCE;
                if (good_match)
                {
                    const size_t dst_id = measurementId(cand_nid, best_feat_idx);
                    if (match_pairs && counts[0] < match_cap)
                        match_pairs[2 * counts[0]] = src_id, match_pairs[2 * counts[0] + 1] = dst_id;
                    counts[0]++;
                    uf.unite(src_id, dst_id);
                }
            }
        }
    }

    // tracks in the order of their smallest member (an insertion-ordered map keyed by root, filled by ascending id)
    std::vector<std::vector<size_t>> multi_tracks;
    {
        std::vector<size_t> track_of_root(id_to_measurement.size(), NONE);
        for (size_t i = 0; i < id_to_measurement.size(); i++)
        {
            if (uf.is_singleton(i))
                continue;
            const size_t root = uf.find(i);
            if (track_of_root[root] == NONE)
            {
                track_of_root[root] = multi_tracks.size();
                multi_tracks.emplace_back();
            }
            multi_tracks[track_of_root[root]].push_back(i);
        }
        multi_tracks.erase(std::remove_if(multi_tracks.begin(), multi_tracks.end(), [](const std::vector<size_t> &t) { return t.size() < 2; }),
                           multi_tracks.end());
    }
    counts[1] = multi_tracks.size();

    const double max_reproj_err_sq = MAX_REPROJECTION_ERROR_PIXELS * MAX_REPROJECTION_ERROR_PIXELS;
    point_cloud merged_points;
    for (const auto &ids : multi_tracks)
    {
        struct RayMeasurement
        {
            Vec3 dir, offset;
            Vec2 pixel;
            const dense_image *img;
        };
        std::vector<RayMeasurement> measurements;
        for (size_t id : ids)
        {
            const Measurement &m = id_to_measurement[id];
            const dense_image &img = images[m.node_id];
            const double *px = img.loc + 2 * m.feat_idx;
            measurements.push_back({rotate(img.orientation, image_to_3d(px, img.model)), img.position, Vec2{px[0], px[1]}, &img});
        }
        // rayIntersection(std::vector<ray_d>) uses the first two rays only (intersection.cpp:145-161)
        auto triangulated = rayIntersection2(measurements[0].dir, measurements[0].offset, measurements[1].dir, measurements[1].offset);
        auto finite = [](const Vec3 &v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); };
        if (!finite(triangulated.first) || triangulated.second < 0)
            continue;
        std::vector<size_t> inlier_indices;
        for (size_t i = 0; i < measurements.size(); i++)
        {
            const RayMeasurement &rm = measurements[i];
            const Vec2 reproj = project(triangulated.first, rm.img->model, rm.img->position, rm.img->orientation);
            const double ex = reproj.x - rm.pixel.x, ey = reproj.y - rm.pixel.y;
            if (ex * ex + ey * ey <= max_reproj_err_sq)
                inlier_indices.push_back(i);
        }
        if (inlier_indices.size() < 2)
            continue;
        if (inlier_indices.size() < measurements.size())
        {
            const RayMeasurement &a = measurements[inlier_indices[0]], &b = measurements[inlier_indices[1]];
            triangulated = rayIntersection2(a.dir, a.offset, b.dir, b.offset);
            if (!finite(triangulated.first) || triangulated.second < 0)
                continue;
        }
        merged_points.push_back(triangulated.first);
    }
    counts[2] = merged_points.size();
    if (points)
        for (size_t i = 0; i < merged_points.size() && i < points_cap; i++)
            points[3 * i] = merged_points[i].x, points[3 * i + 1] = merged_points[i].y, points[3 * i + 2] = merged_points[i].z;
    if (!merged_points.empty())
        surface.cloud.push_back(std::move(merged_points));
}

uint32_t ocx_hilbert_xy2d(int order, int x, int y)
{
    return xy2d(order, x, y);
}

} // extern "C"
