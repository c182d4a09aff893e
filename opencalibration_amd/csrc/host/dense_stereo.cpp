#include "dense_stereo.hpp"

#include "../env.hpp"
#include "invert_distortion.hpp"
#include "relax_util.hpp"
#include "triangle_walker.hpp"

#include <algorithm>
#include <omp.h>
#include <chrono>
#include <cstring>
#include <memory>
#include <numeric>
#include <future>
#include <thread>

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
// (camera_tree.searcher().search(point, max, k), jk::KDTree 3-D - the k cameras nearest to a hit point, nearest first, ties
// to the lower index - is a scan over the cameras inside dense_predict_kernel, csrc/dense.hip)

double seconds_since(const std::chrono::steady_clock::time_point &t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

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

struct pinned_block // page-locked host memory from the context's pool
{
    ochip_ctx *ctx;
    void *p = nullptr;
    ~pinned_block()
    {
        if (p)
            ochip_host_free(ctx, p);
    }
};

inline void project(const v3 &point, const DenseImage &im, double pixel[2]) // image_from_3d(point, model, position, orientation)
{
    const v3 rel{point.x - im.img->position[0], point.y - im.img->position[1], point.z - im.img->position[2]};
    const v3 r = rotate(im.q_inv, rel);
    const double ray[3] = {r.x, r.y, r.z};
    image_from_3d(ray, *im.img->model, pixel);
}

} // namespace

namespace
{

// types/hilbert.hpp:9-28 as written: one level per trip, the quadrant's reflection and transposition applied to (x, y)
uint32_t hilbert_by_levels(int order, int x, int y)
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

// The same index four levels per table look-up.  What a level does to the bits below it is a transposition and / or a
// complement of both coordinates (s - 1 - x is ~x in the bits below s; the two commute), so the walk is a four-state machine
// over the ORIGINAL bits: state = transposed | complemented << 1, a level's digit (3 rx) ^ ry from the state's view of its
// two bits.  A level above the order sees two zero bits: digit 0 and one transposition - starting `extra` levels early
// needs the start state transposed when `extra` is odd.  (A survey's 9.2 M dense features took 94 ns each by levels.)
struct hilbert_table
{
    uint16_t step[4][256]; // [state][x nibble << 4 | y nibble] -> four digits | next state << 8
    hilbert_table()
    {
        for (int state = 0; state < 4; state++)
            for (int xy = 0; xy < 256; xy++)
            {
                int transposed = state & 1, complemented = state >> 1;
                unsigned digits = 0;
                for (int bit = 3; bit >= 0; bit--)
                {
                    const int a = (xy >> (4 + bit)) & 1, b = (xy >> bit) & 1;
                    const int rx = (transposed ? b : a) ^ complemented, ry = (transposed ? a : b) ^ complemented;
                    digits = digits << 2 | (unsigned)((3 * rx) ^ ry);
                    if (ry == 0)
                    {
                        complemented ^= rx;
                        transposed ^= 1;
                    }
                }
                step[state][xy] = (uint16_t)(digits | (unsigned)(transposed | complemented << 1) << 8);
            }
    }
};
const hilbert_table HILBERT;

inline uint32_t hilbert_by_table(int levels, uint32_t x, uint32_t y) // order = 1 << levels <= 65 536, x and y below it
{
    const int groups = (levels + 3) / 4;
    unsigned state = (unsigned)(4 * groups - levels) & 1u;
    uint32_t d = 0;
    for (int g = groups - 1; g >= 0; g--)
    {
        const unsigned e = HILBERT.step[state][((x >> (4 * g)) & 15u) << 4 | ((y >> (4 * g)) & 15u)];
        d = d << 8 | (e & 255u);
        state = e >> 8;
    }
    return d;
}

// (key, feature number) records in the order std::sort puts the reference's (index, feature number) pairs: the numbers
// are distinct and ascend in the input, so a stable sort by key is that order - three counting passes
struct hilbert_rec
{
    uint32_t key, k;
};
void sort_by_key(std::vector<hilbert_rec> &recs, std::vector<hilbert_rec> &other, int key_bits)
{
    const int bits = std::max(1, (key_bits + 2) / 3);
    const uint32_t mask = (1u << bits) - 1;
    const size_t n = recs.size();
    std::vector<uint32_t> hist((size_t)3 << bits, 0u);
    uint32_t *h0 = hist.data(), *h1 = h0 + ((size_t)1 << bits), *h2 = h1 + ((size_t)1 << bits);
    for (size_t i = 0; i < n; i++)
    {
        const uint32_t key = recs[i].key;
        h0[key & mask]++, h1[(key >> bits) & mask]++, h2[(key >> (2 * bits)) & mask]++;
    }
    for (uint32_t *h : {h0, h1, h2})
    {
        uint32_t run = 0;
        for (uint32_t b = 0; b <= mask; b++)
        {
            const uint32_t c = h[b];
            h[b] = run;
            run += c;
        }
    }
    hilbert_rec *src = recs.data(), *dst = other.data();
    int pass = 0;
    for (uint32_t *h : {h0, h1, h2})
    {
        for (size_t i = 0; i < n; i++)
            dst[h[(src[i].key >> (pass * bits)) & mask]++] = src[i];
        std::swap(src, dst);
        pass++;
    }
    recs.swap(other); // three passes: the result is in `other`'s storage
}

} // namespace

uint32_t hilbert_xy2d(int order, int x, int y)
{
    if (order >= 2 && order <= 65536 && (order & (order - 1)) == 0 && x >= 0 && y >= 0 && x < order && y < order)
        return hilbert_by_table(__builtin_ctz((unsigned)order), (uint32_t)x, (uint32_t)y);
    return hilbert_by_levels(order, x, y);
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
    // (descriptors and locations in cell order: 80 bytes per dense feature, written once and uploaded - page-locked staging
    // from the context's pool, no zero fill and no first-touch faults)
    pinned_block desc_stage{ctx}, loc_stage{ctx}, hits_stage{ctx};
    if (ochip_host_alloc(ctx, std::max<size_t>(total, 1) * 64, &desc_stage.p) != OCHIP_OK ||
        ochip_host_alloc(ctx, std::max<size_t>(total, 1) * 16, &loc_stage.p) != OCHIP_OK ||
        ochip_host_alloc(ctx, std::max<size_t>(total, 1) * 24, &hits_stage.p) != OCHIP_OK)
    {
        if (error)
            *error = std::string("ochip_host_alloc: ") + ochip_last_error(ctx);
        return finish(false);
    }
    uint64_t *const desc8 = static_cast<uint64_t *>(desc_stage.p);
    double *const loc2 = static_cast<double *>(loc_stage.p);
    // (every image's bounding box: a pass over the 88-byte records of all dense features - 0.8 GB for a 1 000-image survey -,
    // one image per task like the sort below)
#pragma omp parallel for schedule(dynamic, 4)
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
    // (the ids go up and the roots come back over PCIe: page-locked like the hits - 33 MB each for a 1 000-image survey)
    pinned_block ids_stage{ctx}, root_stage{ctx};
    if (ochip_host_alloc(ctx, total * 4, &ids_stage.p) != OCHIP_OK || ochip_host_alloc(ctx, total * 4, &root_stage.p) != OCHIP_OK)
    {
        if (error)
            *error = std::string("ochip_host_alloc: ") + ochip_last_error(ctx);
        return finish(false);
    }
    // the index goes to the device (0.74 GB for a 1 000-image survey: 15 ms of PCIe) while the rays are walked below: the
    // upload is a thread of its own that is the only one to use the context until it is joined
    ochip_dense_index *index = nullptr;
    int create_rc = OCHIP_OK;
    // (a future from std::async joins in its destructor: nothing that throws between here and the wait below can leave a
    // joinable thread behind; where no thread can be started the index is created right here)
    auto create_index = [&]() {
        create_rc = ochip_dense_index_create(ctx, (uint32_t)n_img, feat_off.data(), desc8, loc2, cell_off.data(), cell_start.data(),
                                             grid2.data(), origin2.data(), CELL_SIZE, &index);
    };
    std::future<void> uploader;
    try
    {
        uploader = std::async(std::launch::async, create_index);
    }
    catch (const std::system_error &)
    {
        create_index();
    }
    st.index_seconds = seconds_since(t0);

    const MeshGraph &mesh = surfaces[0].mesh;
    // ---- where every dense feature's ray meets the mesh (:174-207): the walker starts from the triangle of the previous
    //      feature of the image's Hilbert walk (hilbertFeatureOrder, :24-49), so an image is one sequential task
    auto t1 = std::chrono::steady_clock::now();
    double *const hits = static_cast<double *>(hits_stage.p);
    uint32_t *const id_of_pos = static_cast<uint32_t *>(ids_stage.p);
    std::vector<std::vector<uint32_t>> walk_order(matches_out ? n_img : 0); // dense feature numbers in Hilbert order (tests only)
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t si = 0; si < n_img; si++)
    {
        const DenseImage &src = images[si];
        const image &img = *src.img;
        for (size_t k = 0; k < src.n_dense; k++)
        {
            id_of_pos[src.feat_base + src.dense_to_sorted[k]] = (uint32_t)(src.offset + k);
            hits[3 * (src.feat_base + k)] = NAN;
        }
        TriangleWalker walker;
        if (!walker.init(mesh))
            continue;
        const int w = (int)img.model->pixels_cols, h = (int)img.model->pixels_rows;
        int order = 1;
        while (order < std::max(w, h))
            order *= 2;
        std::vector<hilbert_rec> indexed(src.n_dense), other(src.n_dense);
        for (size_t k = 0; k < src.n_dense; k++)
        {
            const double *l = img.features[img.num_sparse_features + k].location;
            indexed[k] = {hilbert_xy2d(order, std::clamp((int)l[0], 0, w - 1), std::clamp((int)l[1], 0, h - 1)), (uint32_t)k};
        }
        if (order <= 65536)
            sort_by_key(indexed, other, 2 * __builtin_ctz((unsigned)order));
        else
            std::sort(indexed.begin(), indexed.end(),
                      [](const hilbert_rec &a, const hilbert_rec &b) { return a.key != b.key ? a.key < b.key : a.k < b.k; });
        if (matches_out)
        {
            walk_order[si].resize(src.n_dense);
            for (size_t i = 0; i < src.n_dense; i++)
                walk_order[si][i] = indexed[i].k;
        }
        const v3 origin{img.position[0], img.position[1], img.position[2]};
        for (const hilbert_rec &entry : indexed)
        {
            const size_t k = entry.k;
            double ray[3];
            image_to_3d(img.features[img.num_sparse_features + k].location, *img.model, ray);
            const v3 dir = rotate(img.orientation, v3{ray[0], ray[1], ray[2]});
            if (walker.find(dir, origin) != TriangleWalker::INTERSECTION)
                continue;
            double *o = &hits[3 * (src.feat_base + src.dense_to_sorted[k])];
            o[0] = walker.hit.x, o[1] = walker.hit.y, o[2] = walker.hit.z;
            if (std::isnan(o[0])) // (cannot happen after INTERSECTION; NaN in x is the device's "no hit")
                o[0] = NAN;
        }
    }
    st.rays_seconds = seconds_since(t1);
    t1 = std::chrono::steady_clock::now();
    if (uploader.valid())
        uploader.get();
    st.index_seconds += seconds_since(t1); // (what is left of the upload)
    if (create_rc != OCHIP_OK)
    {
        if (error)
            *error = std::string("ochip_dense_index_create: ") + ochip_last_error(ctx);
        return finish(false);
    }
    struct index_guard // (the index outlives the link: ochip_dense_triangulate reads its locations, cameras and ids)
    {
        ochip_dense_index *ix;
        ~index_guard() { ochip_dense_index_destroy(ix); }
    } index_owner{index};

    // ---- nearest cameras, predictions, descriptor search, accept rule and the unions: the device (ochip_dense_link)
    t1 = std::chrono::steady_clock::now();
    std::vector<double> cams17(17 * n_img);
    for (size_t i = 0; i < n_img; i++)
    {
        const DenseImage &d = images[i];
        const CameraModel &m = *d.img->model;
        double *c = &cams17[17 * i];
        std::memcpy(c, d.img->position, 24);
        std::memcpy(c + 3, d.q_inv, 32);
        const double model10[10] = {m.focal_length_pixels,   m.principle_point[0],   m.principle_point[1],      m.radial_distortion[0],
                                    m.radial_distortion[1],  m.radial_distortion[2], m.tangential_distortion[0], m.tangential_distortion[1],
                                    (double)m.pixels_cols,   (double)m.pixels_rows};
        std::memcpy(c + 7, model10, 80);
    }
    uint32_t *const root = static_cast<uint32_t *>(root_stage.p);
    std::vector<uint32_t> slot_dst(matches_out ? total * (MAX_CANDIDATE_IMAGES + 1) : 0);
    uint64_t counts[2] = {0, 0};
    const int lrc = ochip_dense_link(index, cams17.data(), id_of_pos, hits, SEARCH_RADIUS_PIXELS, MAX_CANDIDATE_IMAGES,
                                     feature_2d::DESCRIPTOR_BITS, RATIO_THRESHOLD, MAX_ABSOLUTE_DESCRIPTOR_DISTANCE, root, counts,
                                     matches_out ? slot_dst.data() : nullptr);
    if (lrc != OCHIP_OK)
    {
        if (error)
            *error = std::string("ochip_dense_link: ") + ochip_last_error(ctx);
        return finish(false);
    }
    st.queries = counts[0];
    st.matches = counts[1];
    st.device_seconds = seconds_since(t1);
    if (matches_out) // the reference's order: source images in graph order, features along the Hilbert walk, candidates nearest first
        for (size_t si = 0; si < n_img; si++)
        {
            const DenseImage &src = images[si];
            for (uint32_t k : walk_order[si])
            {
                const uint32_t *sd = &slot_dst[(src.feat_base + src.dense_to_sorted[k]) * (MAX_CANDIDATE_IMAGES + 1)];
                for (int j = 0; j <= MAX_CANDIDATE_IMAGES; j++)
                    if (sd[j] != UINT32_MAX)
                        matches_out->emplace_back(src.offset + k, (size_t)sd[j]);
            }
        }

    // ---- tracks, in the order of their smallest member (:299-340), triangulated from their first two rays
    auto t2 = std::chrono::steady_clock::now();
    // a matched measurement's component = a track; tracks in the order of their smallest member (= their root), members
    // ascending: two counting passes into one flat array (800 k small vectors were a third of this phase)
    // (the members and the points cross the PCIe link for ochip_dense_triangulate: page-locked blocks of the context's pool)
    std::vector<uint32_t> track_start;
    pinned_block member_stage{ctx}, result_stage{ctx};
    uint32_t *track_member = nullptr;
    {
        // (the roots' ranks by per-thread counts)
        const std::unique_ptr<uint32_t[]> rank_of_root(new uint32_t[total]); // (only the roots' entries are written and read)
        const int nt = std::max(1, omp_get_max_threads());
        std::vector<size_t> chunk_roots((size_t)nt + 1, 0);
        const size_t per = (total + (size_t)nt - 1) / (size_t)nt;
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int t = 0; t < nt; t++)
        {
            size_t c = 0;
            for (size_t i = (size_t)t * per; i < std::min(total, (size_t)(t + 1) * per); i++)
                c += root[i] == (uint32_t)i;
            chunk_roots[(size_t)t + 1] = c;
        }
        for (int t = 0; t < nt; t++)
            chunk_roots[(size_t)t + 1] += chunk_roots[(size_t)t];
        const uint32_t n_tracks = (uint32_t)chunk_roots[(size_t)nt];
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int t = 0; t < nt; t++)
        {
            uint32_t r = (uint32_t)chunk_roots[(size_t)t];
            for (size_t i = (size_t)t * per; i < std::min(total, (size_t)(t + 1) * per); i++)
                if (root[i] == (uint32_t)i) // a root is the smallest member of its track: it comes first
                    rank_of_root[i] = r++;
        }
        // members to tracks without atomics: the measurements' (rank, id) records are dealt into buckets by the part of
        // the root range their track's root lies in (ranks ascend with the roots, so a bucket column owns a contiguous run
        // of tracks), one row of buckets per source thread; a column's task then counts and places its records row after
        // row - ids ascend inside a row and from row to row, so every track's members arrive in ascending order
        int shift = 0;
        while (((total - 1) >> shift) >= 128)
            shift++;
        const size_t n_owner = ((total - 1) >> shift) + 1;
        std::vector<std::vector<uint64_t>> bucket((size_t)nt * n_owner);
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int t = 0; t < nt; t++)
        {
            std::vector<uint64_t> *row = bucket.data() + (size_t)t * n_owner;
            const size_t i0 = (size_t)t * per, i1 = std::min(total, (size_t)(t + 1) * per);
            for (size_t o = 0; o < n_owner; o++)
                row[o].reserve((i1 > i0 ? i1 - i0 : 0) / n_owner + 64);
            for (size_t i = i0; i < i1; i++)
                if (root[i] != UINT32_MAX) // (UINT32_MAX: is_singleton)
                    row[root[i] >> shift].push_back((uint64_t)rank_of_root[root[i]] << 32 | (uint64_t)i);
        }
        track_start.assign((size_t)n_tracks + 1, 0);
#pragma omp parallel for schedule(dynamic, 1)
        for (size_t o = 0; o < n_owner; o++)
            for (int t = 0; t < nt; t++)
                for (const uint64_t e : bucket[(size_t)t * n_owner + o])
                    track_start[(size_t)(e >> 32) + 1]++; // (slot rank + 1 belongs to the track of that rank: this column's)
        for (size_t t = 0; t < n_tracks; t++)
            track_start[t + 1] += track_start[t];
        if (ochip_host_alloc(ctx, std::max<size_t>(track_start[n_tracks], 1) * 4, &member_stage.p) != OCHIP_OK)
        {
            if (error)
                *error = std::string("ochip_host_alloc: ") + ochip_last_error(ctx);
            return finish(false);
        }
        track_member = static_cast<uint32_t *>(member_stage.p);
        std::vector<uint32_t> fill(track_start.begin(), track_start.end() - 1);
#pragma omp parallel for schedule(dynamic, 1)
        for (size_t o = 0; o < n_owner; o++)
            for (int t = 0; t < nt; t++)
                for (const uint64_t e : bucket[(size_t)t * n_owner + o])
                    track_member[fill[(size_t)(e >> 32)]++] = (uint32_t)e;
    }
    const size_t n_tracks = track_start.size() - 1;
    st.tracks = n_tracks;
    const double grouping_seconds = seconds_since(t2);
    // measurement id -> image (offsets ascend): a table, filled image by image (a binary search per track member was 4 M searches)
    std::vector<uint32_t> image_of_id(total);
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t i = 0; i < n_img; i++)
        std::fill(image_of_id.begin() + (ptrdiff_t)images[i].offset, image_of_id.begin() + (ptrdiff_t)(images[i].offset + images[i].n_dense), (uint32_t)i);
    auto image_of = [&](size_t id) { return (size_t)image_of_id[id]; };
    const double max_err_sq = MAX_REPROJECTION_ERROR_PIXELS * MAX_REPROJECTION_ERROR_PIXELS;
    static_assert(sizeof(std::array<double, 3>) == 24, "points are three doubles");
    if (ochip_host_alloc(ctx, std::max<size_t>(n_tracks, 1) * 25, &result_stage.p) != OCHIP_OK) // points, then a flag per track
    {
        if (error)
            *error = std::string("ochip_host_alloc: ") + ochip_last_error(ctx);
        return finish(false);
    }
    std::array<double, 3> *const track_results = static_cast<std::array<double, 3> *>(result_stage.p);
    uint8_t *const track_valid = static_cast<uint8_t *>(result_stage.p) + n_tracks * 24;
    std::memset(track_valid, 0, n_tracks);
    // the tracks' points on the device (ochip_dense_triangulate: a thread per track, the operations of the loop below in its
    // order); OCHIP_TEST_HOOKS=host_triangulation keeps the loop - the tests hold the two to each other bit for bit
    const bool on_device = !ochip_test_hook("host_triangulation") && n_tracks > 0;
    if (on_device)
    {
        std::vector<double> cam_q(4 * n_img);
        for (size_t i = 0; i < n_img; i++)
            std::memcpy(&cam_q[4 * i], images[i].img->orientation, 32);
        if (ochip_dense_triangulate(index, cam_q.data(), (uint32_t)n_tracks, track_start.data(), track_member, MAX_REPROJECTION_ERROR_PIXELS,
                                    track_results[0].data(), track_valid) != OCHIP_OK)
        {
            if (error)
                *error = std::string("ochip_dense_triangulate: ") + ochip_last_error(ctx);
            return finish(false);
        }
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (size_t ti = 0; ti < (on_device ? 0 : n_tracks); ti++)
    {
        const uint32_t *ids_begin = track_member + track_start[ti], *ids_end = track_member + track_start[ti + 1];
        if (ids_end - ids_begin < 2)
            continue;
        struct RayMeasurement
        {
            v3 dir, origin;
            const double *pixel;
            const DenseImage *im;
        };
        // (a track has a handful of members - at most one per candidate image -: no heap allocation for the usual ones;
        // 650 k tracks made 1.3 M of them)
        const size_t n_ms = (size_t)(ids_end - ids_begin);
        RayMeasurement ms_small[16];
        std::vector<RayMeasurement> ms_large;
        RayMeasurement *ms_data = ms_small;
        if (n_ms > 16)
        {
            ms_large.resize(n_ms);
            ms_data = ms_large.data();
        }
        struct
        {
            RayMeasurement *p;
            size_t n;
            size_t size() const { return n; }
            RayMeasurement &operator[](size_t i) const { return p[i]; }
        } ms{ms_data, 0};
        for (const uint32_t *ip = ids_begin; ip != ids_end; ip++)
        {
            const size_t id = *ip;
            const DenseImage &im = images[image_of(id)];
            const image &img = *im.img;
            const double *px = img.features[img.num_sparse_features + (id - im.offset)].location;
            double ray[3];
            image_to_3d(px, *img.model, ray);
            ms.p[ms.n++] = RayMeasurement{rotate(img.orientation, v3{ray[0], ray[1], ray[2]}), v3{img.position[0], img.position[1], img.position[2]}, px, &im};
        }
        v3 point;
        double err;
        ray_intersection(ms[0].dir, ms[0].origin, ms[1].dir, ms[1].origin, &point, &err); // only the first two rays (:145-161)
        auto finite = [](const v3 &p) { return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z); };
        if (!finite(point) || err < 0)
            continue;
        // (only the number of inliers and the first two of them matter)
        size_t n_inliers = 0, first_two[2] = {0, 0};
        for (size_t i = 0; i < ms.size(); i++)
        {
            double reproj[2];
            project(point, *ms[i].im, reproj);
            const double ex = reproj[0] - ms[i].pixel[0], ey = reproj[1] - ms[i].pixel[1];
            if (ex * ex + ey * ey <= max_err_sq)
            {
                if (n_inliers < 2)
                    first_two[n_inliers] = i;
                n_inliers++;
            }
        }
        if (n_inliers < 2)
            continue;
        if (n_inliers < ms.size())
        {
            ray_intersection(ms[first_two[0]].dir, ms[first_two[0]].origin, ms[first_two[1]].dir, ms[first_two[1]].origin, &point, &err);
            if (!finite(point) || err < 0)
                continue;
        }
        track_results[ti] = {point.x, point.y, point.z};
        track_valid[ti] = 1;
    }
    const double triangulated_seconds = seconds_since(t2);
    point_cloud merged;
    for (size_t ti = 0; ti < n_tracks; ti++)
        if (track_valid[ti])
            merged.push_back(track_results[ti]);
    st.points = merged.size();
    if (!merged.empty())
        surfaces[0].cloud.push_back(std::move(merged));
    st.tracks_seconds += seconds_since(t2);
    if (ochip_verbose("dense"))
        fprintf(stderr, "[dense] index %.4f s, rays %.4f, device %.4f, tracks %.4f (grouping %.4f, triangulation %.4f, cloud %.4f)\n", st.index_seconds,
                st.rays_seconds, st.device_seconds, st.tracks_seconds, grouping_seconds, triangulated_seconds - grouping_seconds,
                st.tracks_seconds - triangulated_seconds);
    return finish(true);
}

} // namespace opencalibration_amd
