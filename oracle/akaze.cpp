// ORACLE — test infrastructure only.  See akaze.hpp: restated AKAZE + extract_features, PARITY UNPINNED.
//
// KNOWN DEPARTURES FROM OpenCV's AKAZE (features2d/src/kaze/AKAZEFeatures.cpp, nldiffusion_functions.cpp [3P: not under
// /root/reference, absent from this image - what OpenCV does is quoted from its published source, not checked here]).
// Each is a place where this file is deliberately NOT a transcription; scripts/akaze_pin.py --compare attributes what it
// finds to them (classes `suppression`, `angle`, `descriptor`).  STRUCTURE: OpenCV 4.x rewrote parts of the original A-KAZE
// / OpenCV 3.x code, and 4.x is what the reference is built and tested against: its CI runs on ubuntu-24.04 with the
// distribution's libopencv-dev, OpenCV 4.6 (/root/reference/.github/workflows/ci.yml:11, tools/install_dependencies.sh:5).
// Since round 6 this restatement follows the 4.x forms, as recalled (unverifiable in this image):
//   * Find_Scale_Space_Extrema: three passes over per-level keypoint masks (D1 below);
//   * Compute_Main_Orientation: the 109 samples of the radius-6 disc around the ROUNDED position (Sample_Derivative_Response_
//     Radius6: x0 = cvRound(pt.x / ratio), x = x0 + i * scale), their angles sorted into 42 slices of 2 pi / 42 by
//     quantized_counting_sort (slice = (int)(angle / step), out of range -> 0; it fills each slice from its end, so inside a
//     slice the later sample comes first), a window of 7 slices slid over the 42 starts, the last six wrapping; every window
//     summed in sorted order from zero, the first window with the largest norm wins.  (3.x, rounds 2 - 5: per-sample rounding
//     cvRound(xf + i * s), 42 windows stepped by 0.15f rad with strict comparisons, sums in sample order.)
//     The samples' weights are OpenCV's literal table gauss25 (orientation_weights: checkable - it is the Gaussian with pi =
//     3.14159 printed to eight decimals - and 10 float steps away from the exact Gaussian rounds 2 - 5 computed);
//   * the descriptor: the reference passes descriptor_size = 486 (extract_features.cpp:35), and AKAZEFeatures takes the FULL
//     path (MLDB_Full_Descriptor_Invoker: cell means, bits per grid, channel and cell pair) only for descriptor_size == 0; any
//     other size - the full length included - goes through MLDB_Descriptor_Subset_Invoker with the tables of
//     generateDescriptorSubsample: the 162 cell pairs in the order cv::RNG(1024) draws them (mldb_subset_tables), three
//     channel bits per pair, cell SUMS compared (no means), channels (Lt, rx co + ry si, -rx si + ry co), samples at
//     yf + ((l scale) co + (k scale) si).  Rounds 2 - 5 and the first half of round 6 restated the full path: the same 486
//     comparisons in another bit order (Hamming distances between two descriptors of one extractor do not see the order; a
//     descriptor stored by the reference and one from here would).
//   * the contrast factor: compute_kcontrast - every interior pixel's gradient magnitude goes to bin (int)(m * ((nbins - 1) /
//     hmax)), bin 0 is the background, k = hmax * bin / nbins at the first bin >= 1 whose lower bins hold the percentile (3.x's
//     compute_k_percentile: floor(nbins * (m / hmax)), zeros left out, counted from bin 0);
//   * the diffusion step sums its fluxes as the C expression xpos - xneg + ypos - yneg associates (left to right);
//   * a new octave's first image is cv::resize(.., INTER_AREA): the 2 x 2 mean only when BOTH dimensions halve exactly, the general
//     area path (overlap weights on both axes) as soon as one is odd - 1067 -> 533 rows of a 3:2 image (halfsample);
//   * both resizes' overlap tables are built from cv::resize's own scale_x = 1. / inv_scale_x: for the 8-bit working image
//     inv_scale_x is the fx extract_features passes - the FLOAT 1600 / max side as a double, 0.4000000059604645 for a 4000-pixel
//     side, whose reciprocal is not 2.5: the same taps with weights that differ in their last bits;
//   * the level table ends with the first octave below 80 pixels wide or 40 HIGH (Allocate_Memory_Evolution; rounds 2 - 5 had 80
//     for both, one octave less for images of 160 - 319 rows after the resize to 1600).
// A reference built against 3.x would differ in the first two places and in the orientation's details.
//
//  D1  scale-space suppression            (removed in round 6) this file: suppress_masks_4x, called by detect_and_describe, step 2
//      now:    OpenCV 4.x's three passes, in their order.  (1) Per level in raster order, a 3 x 3 maximum looks for the FIRST
//              keypoint already set in [y - r, y + r) x [x - r, x + r) within r = sigma_size (Euclidean; find_neighbor_point's
//              raster scan): none - it is set; it is stronger - that one is cleared and it is set; otherwise it is dropped.
//              (2) Levels upwards, a set keypoint projects into the level below (x, y times the octave step) and clears the
//              first set keypoint there within sigma_size * octave step if it is stronger ("else this pt may be pruned by the
//              upper scale").  (3) Levels downwards, the same against the level above (x, y divided by the octave step, radius
//              that level's sigma_size).  The device evaluates the same passes in dependency rounds instead of in sequence
//              (suppress_masks_4x_in_rounds is its schedule, restated here so that the CPU suite proves the two forms equal).
//      before: a symmetric, order-free rule (rounds 2 - 5; kept as suppress_order_free for the census): a candidate died if ANY
//              stronger maximum of its own or an adjacent level lay within its own size.  scripts/akaze_d1_census.py counts, on
//              rendered 1600 x 1200 views, 2.1 % of its survivors decided differently from the 4.x passes (25 768 against 25 908
//              survivors of 76 034 candidates) - and 43 % differently from the 3.x running list as recalled (suppress_list_3x:
//              36 408 survivors; a new point is compared with the first list entry of its own or the previous level within its
//              size only), so the two OpenCV generations differ from each other far more than the old rule did from 4.x.
//  D2  atan2 / sin / cos                  (removed in round 6) this file: cv_fast_atan2_deg, libm_sincosf; used in detect_and_describe, step 3
//      now:    the sample angles are cv::fastAtan2 (core/src/mathfuncs_core.simd.hpp, atan_f32: the odd polynomial with its
//              coefficients scaled to DEGREES, 90 / 180 / 360 folds) times (float)(CV_PI / 180) as hal::fastAtan32f(.., false)
//              returns them; the keypoint's angle is fastAtan2 of the best window's sums in degrees; the descriptor rotates by
//              cosf / sinf of angle * (float)(CV_PI / 180) - glibc's sinf / cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c,
//              sincosf.h: double-precision kernels, reduction by pi / 2) restated and PINNED: tests/test_oracle_akaze_properties.py
//              compares libm_sincosf with this image's libm over every float in [0, 6.3] and a sweep beyond.
//      before: one polynomial in radians and Taylor sin / cos (rounds 2 - 5): angles differed from OpenCV's by the
//              polynomials' accuracy, a sample on a rounding boundary could move by a pixel.
//  D3  sub-pixel refinement               (removed in round 6) this file: subpixel_solve
//      now:    cv::solve(Matx22f, Vec2f, DECOMP_LU) takes lapack.cpp's 2 x 2 fast path: Cramer's rule in DOUBLE - det2 =
//              (double)a00 * a11 - (double)a01 * a10, d = 1. / det2, x0 = (float)(((double)b0 * a11 - (double)b1 * a01) * d), x1
//              likewise - and a singular system leaves the offset at (0, 0) (AKAZE does not look at solve's result).
//      before: the same rule in float, a singular system dropped the keypoint.
//  D4  order of float sums in the separable filters   this file: gaussian_blur, scharr3, deriv_scale - taps in ascending order, no FMA
//      OpenCV: GaussianBlur, Scharr and sepFilter2D run filter.simd.hpp's symmetric row / column filters: the centre tap first,
//              then every pair of mirrored taps as k * (left + right) - f = k0 * c + k1 * (a + b) + ... - and the vector bodies
//              use v_muladd, which is a fused multiply-add where the build dispatches to AVX2 / FMA3 and a multiply and an add
//              where it does not; scalar tails differ again.  Which of these a pixel sees depends on the CPU the reference
//              runs on and on the pixel's column, so no single restatement is "OpenCV's"; this file keeps one fixed, written
//              order.  The same holds for hal::fastAtan32f's polynomial (v_fma in its vector body; restated without fusing, D2).
//              (The diffusion step, the contrast histogram, the resizes and the rest of detection and description are scalar
//              code in OpenCV and are restated operation for operation.)
//  D5  down-scaling the 8-bit image by exactly 2   this file: resize_area (half_up_cols)
//      OpenCV: cv::resize(INTER_AREA) with an exactly integer scale takes ResizeAreaFast.  For 4 and 8 its result equals the
//              general path's (every operation exact); for 2 (a 3200-pixel side) its vector body rounds (sum + 2) >> 2 and its
//              scalar tail rounds to even.  Restated for the AVX2 build's split - sixteen columns per vector step, the tail the
//              last dw % 16 columns -; a build with 8-lane vectors would round up to eight more columns upwards.
// Everything else (level table, FED step sizes and their reordering, k-contrast percentile, Scharr kernels and their
// normalisation, determinant scaling, the extremum test and its border margin, M-LDB grid and bit order) follows OpenCV's
// structure as recalled; the property tests (tests/test_oracle_akaze_properties.py) hold the restatement to what the
// algorithm must satisfy, nothing holds it to OpenCV's bits.
#include "akaze.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <numeric>

namespace oracle
{
namespace akaze
{

// ------------------------------------------------------------------------------------------ tables
std::vector<float> gaussian_kernel(float sigma)
{
    int ksize = (int)std::ceil(2.0f * (1.0f + (sigma - 0.8f) / 0.3f));
    if ((ksize % 2) == 0)
        ksize += 1;
    if (ksize < 1)
        ksize = 1;
    std::vector<double> k(ksize);
    double sum = 0;
    const int r = ksize / 2;
    for (int i = 0; i < ksize; i++)
    {
        const double x = i - r;
        k[i] = std::exp(-0.5 * x * x / ((double)sigma * sigma));
        sum += k[i];
    }
    std::vector<float> out(ksize);
    for (int i = 0; i < ksize; i++)
        out[i] = (float)(k[i] / sum);
    return out;
}

static bool is_prime(int n)
{
    if (n < 2)
        return false;
    for (int d = 2; d * d <= n; d++)
        if (n % d == 0)
            return false;
    return true;
}

// fed.cpp of AKAZE: fed_tau_by_process_time -> fed_tau_by_cycle_time -> fed_tau_internal [3P]
void fed_tau_by_process_time(float T, int M, float tau_max, bool reordering, std::vector<float> &tau)
{
    const double t = (double)T / M;
    const int n = (int)(std::ceil(std::sqrt(3.0 * t / tau_max + 0.25) - 0.5 - 1.0e-8) + 0.5);
    tau.clear();
    if (n <= 0)
        return;
    const double scale = 3.0 * t / (tau_max * (double)(n * (n + 1)));
    const double c = 1.0 / (4.0 * n + 2.0);
    const double d = scale * tau_max / 2.0;
    std::vector<double> tauh(n);
    for (int k = 0; k < n; k++)
    {
        const double h = std::cos(M_PI * (2.0 * k + 1.0) * c);
        tauh[k] = d / (h * h);
    }
    tau.resize(n);
    if (!reordering)
    {
        for (int k = 0; k < n; k++)
            tau[k] = (float)tauh[k];
        return;
    }
    const int kappa = n / 2;
    int prime = n + 1;
    while (!is_prime(prime))
        prime++;
    for (int k = 0, l = 0; l < n; ++k, ++l)
    {
        int index = 0;
        while ((index = ((k + 1) * kappa) % prime - 1) >= n)
            k++;
        tau[l] = (float)tauh[index];
    }
}

std::vector<Level> make_levels(int width, int height, const Options &o)
{
    std::vector<Level> levels;
    for (int i = 0; i < o.omax; i++)
    {
        const float rfactor = 1.0f / (float)(1 << i);
        const int lw = (int)(width * rfactor), lh = (int)(height * rfactor);
        if ((lw < 80 || lh < 40) && i != 0) // (Allocate_Memory_Evolution: "smallest possible octave" - 80 wide, 40 high)
            break;
        for (int j = 0; j < o.nsublevels; j++)
        {
            Level l;
            l.octave = i;
            l.sublevel = j;
            l.width = lw;
            l.height = lh;
            l.esigma = o.soffset * std::pow(2.0f, (float)j / (float)o.nsublevels + (float)i);
            l.sigma_size = (int)std::lrintf(l.esigma * o.derivative_factor / (float)(1 << i));
            l.etime = 0.5f * (l.esigma * l.esigma);
            levels.push_back(l);
        }
    }
    for (size_t i = 1; i < levels.size(); i++)
        fed_tau_by_process_time(levels[i].etime - levels[i - 1].etime, 1, 0.25f, true, levels[i].tsteps);
    return levels;
}

// Sample_Derivative_Response_Radius6's weights: OpenCV's literal table gauss25 (SURF's: the Gaussian of sigma 2.5 printed to eight
// decimals with pi = 3.14159 - every entry 7.5e-7 above the exact value, which is how the recalled table can be checked: it
// equals round(exp(-(i^2 + j^2) / 12.5) / (2 * 3.14159 * 6.25), 8) in all 49 places; scripts/check_gauss25.py), weight = [|i|][|j|]
std::vector<float> orientation_weights()
{
    static const float gauss25[7][7] = {{0.02546481f, 0.02350698f, 0.01849125f, 0.01239505f, 0.00708017f, 0.00344629f, 0.00142946f},
                                        {0.02350698f, 0.02169968f, 0.01706957f, 0.01144208f, 0.00653582f, 0.00318132f, 0.00131956f},
                                        {0.01849125f, 0.01706957f, 0.01342740f, 0.00900066f, 0.00514126f, 0.00250252f, 0.00103800f},
                                        {0.01239505f, 0.01144208f, 0.00900066f, 0.00603332f, 0.00344629f, 0.00167749f, 0.00069579f},
                                        {0.00708017f, 0.00653582f, 0.00514126f, 0.00344629f, 0.00196855f, 0.00095820f, 0.00039744f},
                                        {0.00344629f, 0.00318132f, 0.00250252f, 0.00167749f, 0.00095820f, 0.00046640f, 0.00019346f},
                                        {0.00142946f, 0.00131956f, 0.00103800f, 0.00069579f, 0.00039744f, 0.00019346f, 0.00008024f}};
    std::vector<float> w(13 * 13);
    for (int i = -6; i <= 6; i++)
        for (int j = -6; j <= 6; j++)
            w[(i + 6) * 13 + (j + 6)] = gauss25[std::abs(i)][std::abs(j)];
    return w;
}

// ---------------------------------------------------------------------------- image building blocks
void bgr_to_gray(const uint8_t *bgr, int w, int h, uint8_t *gray) // cv::cvtColor BGR2GRAY fixed point [3P]
{
    for (size_t i = 0; i < (size_t)w * h; i++)
        gray[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + (1 << 13)) >> 14);
}

// cv::resize INTER_AREA, down-scaling, 8-bit single channel: overlap-weighted box average [3P]
struct area_tab
{
    std::vector<int> si, di;
    std::vector<float> alpha;
};
// computeResizeAreaTab.  `scale` is cv::resize's scale_x = 1. / inv_scale_x: inv_scale_x is the fx it was called with
// (extract_features.cpp:26-27 passes the FLOAT 1600 / max side as a double - 1 / 0.4000000059604645 is not 2.5) or dsize / ssize
// when it was given a size (the octaves' half-sampling).
static area_tab area_table(int ssize, int dsize, double scale)
{
    area_tab t;
    for (int dx = 0; dx < dsize; dx++)
    {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3)
        {
            t.si.push_back(sx1 - 1);
            t.di.push_back(dx);
            t.alpha.push_back((float)((sx1 - fsx1) / cell));
        }
        for (int sx = sx1; sx < sx2; sx++)
        {
            t.si.push_back(sx);
            t.di.push_back(dx);
            t.alpha.push_back((float)(1.0 / cell));
        }
        if (fsx2 - sx2 > 1e-3)
        {
            t.si.push_back(sx2);
            t.di.push_back(dx);
            t.alpha.push_back((float)(std::min(std::min(fsx2 - sx2, 1.0), cell) / cell));
        }
    }
    return t;
}
void resize_area(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, double inv_scale)
{
    if (sw == dw && sh == dh)
    {
        std::memcpy(dst, src, (size_t)sw * sh);
        return;
    }
    const area_tab tx = area_table(sw, dw, 1.0 / inv_scale), ty = area_table(sh, dh, 1.0 / inv_scale);
    std::vector<float> acc((size_t)dw * dh, 0.0f), row(dw);
    for (size_t e = 0; e < ty.si.size(); e++)
    {
        std::fill(row.begin(), row.end(), 0.0f);
        const uint8_t *s = src + (size_t)ty.si[e] * sw;
        for (size_t k = 0; k < tx.si.size(); k++)
            row[tx.di[k]] += (float)s[tx.si[k]] * tx.alpha[k];
        float *a = acc.data() + (size_t)ty.di[e] * dw;
        for (int x = 0; x < dw; x++)
            a[x] += row[x] * ty.alpha[e];
    }
    // a scale of exactly 2 is cv::resize's integer path (ResizeAreaFast): its vector body rounds (a + b + c + d + 2) >> 2 - acc is
    // exact there -, sixteen columns at a time on an AVX2 build, and its scalar tail rounds to even like the general path (D5)
    const int half_up_cols = inv_scale == 0.5 ? dw - dw % 16 : 0;
    for (size_t i = 0; i < acc.size(); i++)
    {
        const long v = (int)(i % (size_t)dw) < half_up_cols ? (long)std::floor(acc[i] + 0.5f) : std::lrintf(acc[i]);
        dst[i] = (uint8_t)std::min(255L, std::max(0L, v));
    }
}

static inline int clampi(int v, int lo, int hi)
{
    return v < lo ? lo : (v > hi ? hi : v);
}
static inline int reflect101(int v, int n)
{
    if (n == 1)
        return 0;
    while (v < 0 || v >= n)
        v = v < 0 ? -v : 2 * (n - 1) - v;
    return v;
}

// separable Gaussian, rows then columns, BORDER_REPLICATE, taps accumulated in ascending order
static void gaussian_blur(const std::vector<float> &in, std::vector<float> &out, int w, int h, const std::vector<float> &k)
{
    const int n = (int)k.size(), r = n / 2;
    std::vector<float> tmp((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            float acc = 0.0f;
            for (int t = 0; t < n; t++)
                acc = acc + k[t] * in[(size_t)y * w + clampi(x + t - r, 0, w - 1)];
            tmp[(size_t)y * w + x] = acc;
        }
    out.resize((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            float acc = 0.0f;
            for (int t = 0; t < n; t++)
                acc = acc + k[t] * tmp[(size_t)clampi(y + t - r, 0, h - 1) * w + x];
            out[(size_t)y * w + x] = acc;
        }
}

// unnormalised 3x3 Scharr (cv::Scharr scale 1), BORDER_REFLECT_101
static inline void scharr3(const float *I, int w, int h, int x, int y, float *lx, float *ly)
{
    const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w), ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
    const float a = I[(size_t)ym * w + xp] - I[(size_t)ym * w + xm];
    const float b = I[(size_t)y * w + xp] - I[(size_t)y * w + xm];
    const float c = I[(size_t)yp * w + xp] - I[(size_t)yp * w + xm];
    *lx = (3.0f * a + 10.0f * b) + 3.0f * c;
    const float d = I[(size_t)yp * w + xm] - I[(size_t)ym * w + xm];
    const float e = I[(size_t)yp * w + x] - I[(size_t)ym * w + x];
    const float f = I[(size_t)yp * w + xp] - I[(size_t)ym * w + xp];
    *ly = (3.0f * d + 10.0f * e) + 3.0f * f;
}

float compute_k_percentile(const std::vector<float> &img, int w, int h, const Options &o)
{
    std::vector<float> sm;
    gaussian_blur(img, sm, w, h, gaussian_kernel(1.0f));
    float hmax = 0.0f;
    std::vector<float> modg((size_t)w * h, 0.0f);
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++)
        {
            float lx, ly;
            scharr3(sm.data(), w, h, x, y, &lx, &ly);
            const float m = std::sqrt(lx * lx + ly * ly);
            modg[(size_t)y * w + x] = m;
            if (m > hmax)
                hmax = m;
        }
    // OpenCV 4.x's compute_kcontrast (rounds 2 - 5 had 3.x's compute_k_percentile: nbins * (m / hmax) floored, zeros left out,
    // bins counted from 0, hmax * (k / nbins)): every interior pixel goes to bin (int)(m * ((nbins - 1) / hmax)), bin 0 is the
    // background, and the contrast is hmax * k / nbins at the first k >= 1 whose lower bins 1 .. k - 1 hold the percentile
    if (hmax == 0.0f)
        return 0.03f; // (a blank image)
    const int nbins = o.kcontrast_nbins;
    std::vector<int> hist(nbins, 0);
    const float to_bin = (float)(nbins - 1) / hmax;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++)
            hist[(int)(modg[(size_t)y * w + x] * to_bin)]++;
    const int total = (w - 2) * (h - 2);
    const int nthreshold = (int)((float)(total - hist[0]) * o.kcontrast_percentile);
    int nelements = 0;
    for (int k = 1; k < nbins; k++)
    {
        if (nelements >= nthreshold)
            return hmax * (float)k / (float)nbins;
        nelements = nelements + hist[k];
    }
    return 0.03f;
}

static void pm_g2_flow(const std::vector<float> &sm, std::vector<float> &flow, int w, int h, float k)
{
    flow.resize((size_t)w * h);
    const float inv = 1.0f / (k * k);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            float lx, ly;
            scharr3(sm.data(), w, h, x, y, &lx, &ly);
            flow[(size_t)y * w + x] = 1.0f / (1.0f + inv * (lx * lx + ly * ly));
        }
}

static void nld_step(const std::vector<float> &L, const std::vector<float> &c, std::vector<float> &out, int w, int h, float tau)
{
    out.resize((size_t)w * h);
    const float half = 0.5f * tau;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const size_t i = (size_t)y * w + x;
            const float xpos = x + 1 < w ? (c[i] + c[i + 1]) * (L[i + 1] - L[i]) : 0.0f;
            const float xneg = x > 0 ? (c[i - 1] + c[i]) * (L[i] - L[i - 1]) : 0.0f;
            const float ypos = y + 1 < h ? (c[i] + c[i + w]) * (L[i + w] - L[i]) : 0.0f;
            const float yneg = y > 0 ? (c[i - w] + c[i]) * (L[i] - L[i - w]) : 0.0f;
            out[i] = L[i] + half * (((xpos - xneg) + ypos) - yneg); // (NonLinearScalarDiffusionStep writes xpos - xneg + ypos - yneg)
        }
}

// The next octave's first image: cv::resize(Lt, size / 2, INTER_AREA) on a float plane.  With both dimensions exactly halved
// OpenCV takes its integer-scale path (ResizeAreaFast: the 2 x 2 mean, ((a + b) + (c + d)) * 0.25f in its vector body); as soon as
// one dimension is odd - 1067 -> 533 rows of a 3:2 image, 333 -> 166 - neither scale is an integer to it and the GENERAL
// area path runs on both axes: computeResizeAreaTab's overlap weights (area_table above, the same as for the 8-bit resize), per
// source row buf[dx] += S[sx] * alpha in table order, then sum = beta * buf for a destination row's first source row and
// sum += beta * buf for the others.  (Rounds 2 - 5 clamped the 2 x 2 mean at the odd edge instead.)
static void halfsample(const std::vector<float> &in, int w, int h, std::vector<float> &out, int ow, int oh)
{
    out.resize((size_t)ow * oh);
    if (w == 2 * ow && h == 2 * oh)
    {
        for (int y = 0; y < oh; y++)
            for (int x = 0; x < ow; x++)
                out[(size_t)y * ow + x] = ((in[(size_t)(2 * y) * w + 2 * x] + in[(size_t)(2 * y) * w + 2 * x + 1]) +
                                           (in[(size_t)(2 * y + 1) * w + 2 * x] + in[(size_t)(2 * y + 1) * w + 2 * x + 1])) *
                                          0.25f;
        return;
    }
    const area_tab tx = area_table(w, ow, 1.0 / ((double)ow / w)), ty = area_table(h, oh, 1.0 / ((double)oh / h));
    std::vector<float> buf(ow);
    int prev_dy = -1;
    for (size_t e = 0; e < ty.si.size(); e++)
    {
        std::fill(buf.begin(), buf.end(), 0.0f);
        const float *S = in.data() + (size_t)ty.si[e] * w;
        for (size_t k = 0; k < tx.si.size(); k++)
            buf[tx.di[k]] = buf[tx.di[k]] + S[tx.si[k]] * tx.alpha[k];
        float *sum = out.data() + (size_t)ty.di[e] * ow;
        const float beta = ty.alpha[e];
        if (ty.di[e] != prev_dy)
            for (int x = 0; x < ow; x++)
                sum[x] = beta * buf[x];
        else
            for (int x = 0; x < ow; x++)
                sum[x] = sum[x] + beta * buf[x];
        prev_dy = ty.di[e];
    }
}

// first derivatives at integer scale s: Scharr-like 3-tap kernels spread to distance s, reflect101
static void deriv_scale(const std::vector<float> &I, std::vector<float> &Dx, std::vector<float> &Dy, int w, int h, int s)
{
    const float wgt = 10.0f / 3.0f;
    const float nrm = 1.0f / (2.0f * (float)s * (wgt + 2.0f));
    const float wn = wgt * nrm;
    Dx.resize((size_t)w * h);
    Dy.resize((size_t)w * h);
    for (int y = 0; y < h; y++)
    {
        const int ym = reflect101(y - s, h), yp = reflect101(y + s, h);
        for (int x = 0; x < w; x++)
        {
            const int xm = reflect101(x - s, w), xp = reflect101(x + s, w);
            const float a = I[(size_t)ym * w + xp] - I[(size_t)ym * w + xm];
            const float b = I[(size_t)y * w + xp] - I[(size_t)y * w + xm];
            const float c = I[(size_t)yp * w + xp] - I[(size_t)yp * w + xm];
            Dx[(size_t)y * w + x] = (nrm * a + wn * b) + nrm * c;
            const float d = I[(size_t)yp * w + xm] - I[(size_t)ym * w + xm];
            const float e = I[(size_t)yp * w + x] - I[(size_t)ym * w + x];
            const float f = I[(size_t)yp * w + xp] - I[(size_t)ym * w + xp];
            Dy[(size_t)y * w + x] = (nrm * d + wn * e) + nrm * f;
        }
    }
}

ScaleSpace build_scale_space(const std::vector<float> &img, int w, int h, const Options &o)
{
    ScaleSpace ss;
    ss.levels = make_levels(w, h, o);
    const size_t N = ss.levels.size();
    ss.Lt.resize(N);
    ss.Lx.resize(N);
    ss.Ly.resize(N);
    ss.Ldet.resize(N);
    gaussian_blur(img, ss.Lt[0], w, h, gaussian_kernel(o.soffset));
    float kcontrast = compute_k_percentile(img, w, h, o);
    ss.kcontrast = kcontrast;
    const std::vector<float> g1 = gaussian_kernel(o.sderivatives);
    // Lsmooth of every level: what the conductivity AND the detector's multiscale derivatives are computed on
    // (AKAZE's Compute_Multiscale_Derivatives works on evolution[i].Lsmooth).  It is the Gaussian(1) smoothing of the
    // image the level STARTS from - the previous level's result, half-sampled at a new octave - taken before the
    // level's own diffusion steps; level 0's Lsmooth is its Lt.
    std::vector<std::vector<float>> Lsmooth(N);
    Lsmooth[0] = ss.Lt[0];
    std::vector<float> flow, nxt;
    for (size_t i = 1; i < N; i++)
    {
        const Level &l = ss.levels[i], &p = ss.levels[i - 1];
        if (l.octave > p.octave)
        {
            halfsample(ss.Lt[i - 1], p.width, p.height, ss.Lt[i], l.width, l.height);
            kcontrast = kcontrast * 0.75f;
        }
        else
            ss.Lt[i] = ss.Lt[i - 1];
        gaussian_blur(ss.Lt[i], Lsmooth[i], l.width, l.height, g1);
        pm_g2_flow(Lsmooth[i], flow, l.width, l.height, kcontrast);
        for (float tau : l.tsteps)
        {
            nld_step(ss.Lt[i], flow, nxt, l.width, l.height, tau);
            ss.Lt[i].swap(nxt);
        }
    }
    std::vector<float> lxx, lxy, lyy, tmp;
    for (size_t i = 0; i < N; i++)
    {
        const Level &l = ss.levels[i];
        deriv_scale(Lsmooth[i], ss.Lx[i], ss.Ly[i], l.width, l.height, l.sigma_size);
        deriv_scale(ss.Lx[i], lxx, lxy, l.width, l.height, l.sigma_size);
        deriv_scale(ss.Ly[i], tmp, lyy, l.width, l.height, l.sigma_size);
        const float s4 = (float)(l.sigma_size * l.sigma_size * l.sigma_size * l.sigma_size);
        ss.Ldet[i].resize(lxx.size());
        for (size_t k = 0; k < lxx.size(); k++)
            ss.Ldet[i][k] = (lxx[k] * lyy[k] - lxy[k] * lxy[k]) * s4;
    }
    return ss;
}

// ------------------------------------------------------------------------ float-only math helpers
// (deterministic across CPU and GPU: only + - * / and comparisons)
static const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;

// cv::fastAtan2 (modules/core/src/mathfuncs_core.simd.hpp, atan_f32): degrees in [0, 360]
float cv_fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale, p5 = 0.1555786518463281f * scale,
                p7 = -0.04432655554792128f * scale;
    const float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay)
    {
        c = ay / (ax + 2.220446e-16f); // (float)DBL_EPSILON
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    else
    {
        c = ax / (ay + 2.220446e-16f);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0)
        a = 180.f - a;
    if (y < 0)
        a = 360.f - a;
    return a;
}
static const float DEG2RAD_F = (float)(3.1415926535897932384626433832795 / 180.0);
float fast_atan2(float y, float x) // hal::fastAtan32f(y, x, angle, n, false): the degrees above times (float)(CV_PI / 180)
{
    return cv_fast_atan2_deg(y, x) * DEG2RAD_F;
}

// glibc's sinf and cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c with sincosf.h, the non-FMA C code): for |x| < 120 the
// argument goes to double, is reduced by multiples of pi / 2 (reduce_fast) and one of two double polynomials gives the value.
// Arguments of 120 and more (and NaN / inf) do not occur here: an angle is within [0, 2 pi].
namespace
{
struct sincos_tab
{
    double sign[4], hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3;
};
const sincos_tab SINCOSF_TABLE[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
     -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
     0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
inline float sinf_poly(double x, double x2, const sincos_tab *p, int n)
{
    if ((n & 1) == 0)
    {
        const double x3 = x * x2;
        const double s1 = p->s2 + x2 * p->s3;
        const double x7 = x3 * x2;
        const double s = x + x3 * p->s1;
        return (float)(s + x7 * s1);
    }
    const double x4 = x2 * x2;
    const double c2 = p->c3 + x2 * p->c4;
    const double c1 = p->c1 + x2 * p->c2;
    const double x6 = x4 * x2;
    const double c = p->c0 + x2 * c1;
    return (float)(c + x6 * c2);
}
inline uint32_t abstop12(float x)
{
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return (u >> 20) & 0x7ff;
}
inline double reduce_fast(double x, const sincos_tab *p, int *np)
{
    const double r = x * p->hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * p->hpi;
}
} // namespace
float libm_sinf(float y)
{
    double x = y;
    const sincos_tab *p = &SINCOSF_TABLE[0];
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) // |y| < pi / 4
    {
        const double s = x * x;
        if (abstop12(y) < abstop12(0x1p-12f))
            return y;
        return sinf_poly(x, s, p, 0);
    }
    int n;
    x = reduce_fast(x, p, &n);
    const double s = p->sign[n & 3];
    if (n & 2)
        p = &SINCOSF_TABLE[1];
    return sinf_poly(x * s, x * x, p, n);
}
float libm_cosf(float y)
{
    double x = y;
    const sincos_tab *p = &SINCOSF_TABLE[0];
    if (abstop12(y) < abstop12(0x1.921FB6p-1f))
    {
        const double x2 = x * x;
        if (abstop12(y) < abstop12(0x1p-12f))
            return 1.0f;
        return sinf_poly(x, x2, p, 1);
    }
    int n;
    x = reduce_fast(x, p, &n);
    const double s = p->sign[n & 3];
    if (n & 2)
        p = &SINCOSF_TABLE[1];
    return sinf_poly(x * s, x * x, p, n ^ 1);
}
void libm_sincosf(float a, float *s, float *c)
{
    *s = libm_sinf(a);
    *c = libm_cosf(a);
}

// cv::solve(A, b, dst, DECOMP_LU) for AKAZE's 2 x 2 system [Dxx Dxy; Dxy Dyy] d = -[Dx Dy] (core/src/lapack.cpp, the 2 x 2
// fast path for CV_32FC1): Cramer's rule in double; a singular system leaves dst = (0, 0)
void subpixel_solve(float Dxx, float Dxy, float Dyy, float Dx, float Dy, float *dx, float *dy)
{
    const float b0 = -Dx, b1 = -Dy;
    double d = (double)Dxx * Dyy - (double)Dxy * Dxy;
    *dx = 0.0f;
    *dy = 0.0f;
    if (d != 0.)
    {
        d = 1. / d;
        *dx = (float)(((double)b0 * Dyy - (double)b1 * Dxy) * d);
        *dy = (float)(((double)b1 * Dxx - (double)b0 * Dxy) * d);
    }
}

// --------------------------------------------------------------------------------- detect + describe
struct cand
{
    int level, x, y;
    float response;
};

// step 1 of detect_and_describe: the strict 3 x 3 maxima of every level above the threshold whose descriptor window stays inside
// the level, in detection order (level, row, column)
static std::vector<cand> find_candidates(const ScaleSpace &ss, const Options &o)
{
    const size_t N = ss.levels.size();
    (void)N;
    // per-level 3x3 maxima above the threshold
    std::vector<cand> cands;
    for (size_t i = 0; i < N; i++)
    {
        const Level &l = ss.levels[i];
        const float *D = ss.Ldet[i].data();
        const int w = l.width, h = l.height;
        // AKAZE's Find_Scale_Space_Extrema drops points whose descriptor window would leave the level image:
        // [round(x - smax s) - 1, round(x + smax s) + 1] must lie inside, smax = 10 sqrt(2) for M-LDB, s = sigma_size
        const float margin = (10.0f * std::sqrt(2.0f)) * (float)l.sigma_size;
        for (int y = 1; y < h - 1; y++)
            for (int x = 1; x < w - 1; x++)
            {
                const float v = D[(size_t)y * w + x];
                if (!(v > o.dthreshold))
                    continue;
                if ((int)std::rint((float)x - margin) - 1 < 0 || (int)std::rint((float)x + margin) + 1 >= w ||
                    (int)std::rint((float)y - margin) - 1 < 0 || (int)std::rint((float)y + margin) + 1 >= h)
                    continue;
                bool mx = true;
                for (int dy = -1; dy <= 1 && mx; dy++)
                    for (int dx = -1; dx <= 1; dx++)
                        if ((dx || dy) && !(v > D[(size_t)(y + dy) * w + x + dx]))
                        {
                            mx = false;
                            break;
                        }
                if (mx)
                    cands.push_back(cand{(int)i, x, y, v});
            }
    }
    return cands;
}

// the rule of rounds 2 - 5 (header, D1 'before'): symmetric and order-free.  Census only.
static std::vector<char> suppress_order_free(const ScaleSpace &ss, const Options &o, const std::vector<cand> &cands)
{
    // scale-space suppression: a candidate dies if a stronger one (ties: lower (level,y,x) wins) of
    //    an adjacent level lies within its own size (esigma * derivative_factor, base-image pixels)
    std::vector<char> dead(cands.size(), 0);
    {
        std::map<std::pair<int, int>, std::vector<int>> grid; // 32 px buckets in base coordinates
        auto bx = [&](const cand &c) { return (float)c.x * (float)(1 << ss.levels[c.level].octave); };
        auto by = [&](const cand &c) { return (float)c.y * (float)(1 << ss.levels[c.level].octave); };
        for (size_t k = 0; k < cands.size(); k++)
            grid[{(int)(bx(cands[k]) / 32.0f), (int)(by(cands[k]) / 32.0f)}].push_back((int)k);
        for (size_t k = 0; k < cands.size(); k++)
        {
            const cand &c = cands[k];
            const float rad = ss.levels[c.level].esigma * o.derivative_factor, r2 = rad * rad;
            const float cx = bx(c), cy = by(c);
            const int gx0 = (int)((cx - rad) / 32.0f) - 1, gx1 = (int)((cx + rad) / 32.0f) + 1;
            const int gy0 = (int)((cy - rad) / 32.0f) - 1, gy1 = (int)((cy + rad) / 32.0f) + 1;
            for (int gy = gy0; gy <= gy1 && !dead[k]; gy++)
                for (int gx = gx0; gx <= gx1 && !dead[k]; gx++)
                {
                    auto it = grid.find({gx, gy});
                    if (it == grid.end())
                        continue;
                    for (int m : it->second)
                    {
                        if ((size_t)m == k)
                            continue;
                        const cand &d = cands[m];
                        if (std::abs(d.level - c.level) > 1)
                            continue;
                        const float ex = bx(d) - cx, ey = by(d) - cy;
                        if (!(ex * ex + ey * ey <= r2))
                            continue;
                        const bool stronger = d.response > c.response ||
                                              (d.response == c.response &&
                                               std::make_tuple(d.level, d.y, d.x) < std::make_tuple(c.level, c.y, c.x));
                        if (stronger)
                        {
                            dead[k] = 1;
                            break;
                        }
                    }
                }
        }
    }
    return dead;
}

// ---- OpenCV's two sequential rules on the same candidates, as recalled (header of this file).  detect_and_describe runs the
// 4.x one; the 3.x one is here so that the difference between the generations can be COUNTED (scripts/akaze_d1_census.py).
// OpenCV 3.x (the original A-KAZE loop): a running list in detection order; a new point is compared with the FIRST list entry of
// its own or the previous level within its size - it takes that entry's place when stronger, is dropped otherwise -, then
// every list entry is dropped that has a stronger entry of the next level behind it in the list within its size.
static std::vector<char> suppress_list_3x(const ScaleSpace &ss, const Options &o, const std::vector<cand> &cands)
{
    struct entry
    {
        int cand, class_id;
        float x, y, size, response;
    };
    std::vector<entry> list;
    for (size_t k = 0; k < cands.size(); k++)
    {
        const cand &c = cands[k];
        const Level &l = ss.levels[c.level];
        const float ratio = (float)(1 << l.octave), size = l.esigma * o.derivative_factor;
        bool is_extremum = true, repeated = false;
        size_t id_repeated = 0;
        for (size_t ik = 0; ik < list.size(); ik++)
            if (list[ik].class_id == c.level - 1 || list[ik].class_id == c.level)
            {
                const float dx = (float)c.x * ratio - list[ik].x, dy = (float)c.y * ratio - list[ik].y;
                if (dx * dx + dy * dy <= size * size)
                {
                    if (c.response > list[ik].response)
                        id_repeated = ik, repeated = true;
                    else
                        is_extremum = false;
                    break;
                }
            }
        if (!is_extremum)
            continue;
        const entry e{(int)k, c.level, (float)((float)c.x * ratio + .5 * (ratio - 1.0)), (float)((float)c.y * ratio + .5 * (ratio - 1.0)), size,
                      c.response};
        if (repeated)
            list[id_repeated] = e;
        else
            list.push_back(e);
    }
    std::vector<char> dead(cands.size(), 1);
    for (size_t i = 0; i < list.size(); i++)
    {
        bool repeated = false;
        for (size_t j = i + 1; j < list.size() && !repeated; j++)
            if (list[i].class_id + 1 == list[j].class_id)
            {
                const float dx = list[i].x - list[j].x, dy = list[i].y - list[j].y;
                repeated = dx * dx + dy * dy <= list[i].size * list[i].size && list[i].response < list[j].response;
            }
        if (!repeated)
            dead[(size_t)list[i].cand] = 0;
    }
    return dead;
}
// OpenCV 4.x: per-level masks (header, D1 'now').  Same level (raster order): the first keypoint already set within sigma_size of
// the new one is cleared when the new one is stronger, else the new one is dropped.  Then every level against the level below (in
// raster order, the first set keypoint of the lower level within sigma_size * octave step of the projected position is cleared
// when the upper one is stronger - the upper one stays either way), then every level against the level above, from the top down.
static std::vector<char> suppress_masks_4x(const ScaleSpace &ss, const Options &o, const std::vector<cand> &cands)
{
    const size_t N = ss.levels.size();
    std::vector<std::vector<int>> mask(N); // candidate index + 1 at its pixel, 0 elsewhere
    for (size_t i = 0; i < N; i++)
        mask[i].assign((size_t)ss.levels[i].width * ss.levels[i].height, 0);
    auto find_neighbor = [&](size_t lvl, int x, int y, int r, int *idx) {
        const int w = ss.levels[lvl].width, h = ss.levels[lvl].height;
        for (int i = std::max(y - r, 0); i < std::min(y + r, h); i++)
            for (int j = std::max(x - r, 0); j < std::min(x + r, w); j++)
                if (mask[lvl][(size_t)i * w + j] != 0 && (j - x) * (j - x) + (i - y) * (i - y) <= r * r)
                {
                    *idx = i * w + j;
                    return true;
                }
        return false;
    };
    for (size_t k = 0; k < cands.size(); k++) // (detection order = level by level, raster inside a level)
    {
        const cand &c = cands[k];
        const int w = ss.levels[c.level].width;
        int idx = 0;
        if (find_neighbor((size_t)c.level, c.x, c.y, ss.levels[c.level].sigma_size, &idx))
        {
            if (c.response > cands[(size_t)mask[c.level][idx] - 1].response)
                mask[c.level][idx] = 0;
            else
                continue;
        }
        mask[c.level][(size_t)c.y * w + c.x] = (int)k + 1;
    }
    for (size_t i = 1; i < N; i++) // against the level below
    {
        const int w = ss.levels[i].width, h = ss.levels[i].height;
        const int diff = (1 << ss.levels[i].octave) / (1 << ss.levels[i - 1].octave), r = ss.levels[i].sigma_size * diff;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
            {
                const int me = mask[i][(size_t)y * w + x];
                int idx = 0;
                if (me != 0 && find_neighbor(i - 1, x * diff, y * diff, r, &idx))
                {
                    if (cands[(size_t)me - 1].response > cands[(size_t)mask[i - 1][idx] - 1].response)
                        mask[i - 1][idx] = 0; // (else: the pass from the top down may prune this point)
                }
            }
    }
    for (int i = (int)N - 2; i >= 0; i--) // against the level above
    {
        const int w = ss.levels[i].width, h = ss.levels[i].height;
        const int diff = (1 << ss.levels[i + 1].octave) / (1 << ss.levels[i].octave), r = ss.levels[i + 1].sigma_size;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
            {
                const int me = mask[i][(size_t)y * w + x];
                int idx = 0;
                if (me != 0 && find_neighbor((size_t)i + 1, x / diff, y / diff, r, &idx))
                {
                    if (cands[(size_t)me - 1].response > cands[(size_t)mask[i + 1][idx] - 1].response)
                        mask[i + 1][idx] = 0;
                }
            }
    }
    std::vector<char> dead(cands.size(), 1);
    for (size_t i = 0; i < N; i++)
        for (int v : mask[i])
            if (v != 0)
                dead[(size_t)v - 1] = 0;
    return dead;
}
// The 4.x rule as the DEVICE evaluates it (csrc/akaze.hip, suppress_rounds_kernel): not in sequence but in rounds.  Pass 1
// (inside a level): a maximum takes its turn once every maximum in front of it in raster order within 2 sigma_size (Chebyshev)
// has had its own - two maxima further apart read and write disjoint neighbourhoods.  Passes 2 and 3 only ever clear keypoints
// of the OTHER level and read their own level as the previous pass left it, so the levels of a pass are independent given that
// snapshot, and inside a level a keypoint waits for the keypoints in front of it whose search windows can overlap its own.
// Must give suppress_masks_4x's masks exactly; rounds[p] = rounds pass p needed.
static std::vector<char> suppress_masks_4x_in_rounds(const ScaleSpace &ss, const Options &o, const std::vector<cand> &cands, int rounds[3])
{
    (void)o;
    const size_t N = ss.levels.size();
    std::vector<std::vector<int>> own(N), W(N); // candidate index + 1 at its pixel: who takes turns in this pass / the masks searched and cleared
    std::vector<std::vector<char>> D(N);        // per pixel: the candidate there has had its turn in the current pass
    for (size_t i = 0; i < N; i++)
    {
        own[i].assign((size_t)ss.levels[i].width * ss.levels[i].height, 0);
        W[i] = own[i];
        D[i].assign(own[i].size(), 0);
    }
    for (size_t k = 0; k < cands.size(); k++)
        own[cands[k].level][(size_t)cands[k].y * ss.levels[cands[k].level].width + cands[k].x] = (int)k + 1;
    auto first_set = [&](size_t lvl, int x, int y, int r, int *idx) {
        const int w = ss.levels[lvl].width, h = ss.levels[lvl].height;
        for (int i = std::max(y - r, 0); i < std::min(y + r, h); i++)
            for (int j = std::max(x - r, 0); j < std::min(x + r, w); j++)
                if (W[lvl][(size_t)i * w + j] != 0 && (j - x) * (j - x) + (i - y) * (i - y) <= r * r)
                {
                    *idx = i * w + j;
                    return true;
                }
        return false;
    };
    // a point of this pass in front of (x, y) in raster order, inside the box of half-width r, that has not had its turn
    auto pending = [&](size_t lvl, int x, int y, int r) {
        const int w = ss.levels[lvl].width;
        for (int i = std::max(y - r, 0); i <= y; i++)
            for (int j = std::max(x - r, 0); j <= std::min(x + r, w - 1); j++)
            {
                if (i == y && j >= x)
                    break;
                if (own[lvl][(size_t)i * w + j] != 0 && !D[lvl][(size_t)i * w + j])
                    return true;
            }
        return false;
    };
    auto run_pass = [&](int pass) {
        for (size_t i = 0; i < N; i++)
            std::fill(D[i].begin(), D[i].end(), 0);
        std::vector<size_t> todo;
        for (size_t k = 0; k < cands.size(); k++)
        {
            const cand &c = cands[k];
            const size_t i = (size_t)c.level;
            bool has_turn = own[i][(size_t)c.y * ss.levels[i].width + c.x] == (int)k + 1 && !(pass == 2 && i == 0) && !(pass == 3 && i == N - 1);
            if (has_turn && pass != 1)
            {
                // a keypoint whose window holds no keypoint of the other level when the pass starts never finds one (the pass
                // only clears) and clears nothing: it has no turn to take and nobody waits for it
                int idx = 0;
                if (pass == 2)
                {
                    const int diff = (1 << ss.levels[i].octave) / (1 << ss.levels[i - 1].octave);
                    has_turn = first_set(i - 1, c.x * diff, c.y * diff, ss.levels[i].sigma_size * diff, &idx);
                }
                else
                {
                    const int diff = (1 << ss.levels[i + 1].octave) / (1 << ss.levels[i].octave);
                    has_turn = first_set(i + 1, c.x / diff, c.y / diff, ss.levels[i + 1].sigma_size, &idx);
                }
            }
            if (has_turn)
                todo.push_back(k);
        }
        // (who waits for whom is asked of the points WITH a turn only)
        for (size_t i = 0; i < N; i++)
            std::fill(own[i].begin(), own[i].end(), 0);
        for (size_t k : todo)
            own[cands[k].level][(size_t)cands[k].y * ss.levels[cands[k].level].width + cands[k].x] = (int)k + 1;
        int n_rounds = 0;
        while (!todo.empty())
        {
            std::vector<size_t> ready, later;
            for (size_t k : todo)
            {
                const cand &c = cands[k];
                const size_t i = (size_t)c.level;
                int box; // (two windows [p - r, p + r) overlap iff the centres are at most 2 r - 1 apart)
                if (pass == 1 || pass == 2)
                    box = 2 * ss.levels[i].sigma_size - 1;
                else
                    box = 2 * ss.levels[i + 1].sigma_size * ((1 << ss.levels[i + 1].octave) / (1 << ss.levels[i].octave)) - 1;
                (pending(i, c.x, c.y, box) ? later : ready).push_back(k);
            }
            n_rounds++;
            // the round's turns: every ready point reads the masks as the round found them (ready points do not influence each other)
            struct change
            {
                size_t lvl;
                int idx, value;
            };
            std::vector<change> changes;
            for (size_t k : ready)
            {
                const cand &c = cands[k];
                const size_t i = (size_t)c.level;
                int idx = 0;
                if (pass == 1)
                {
                    bool keep = true;
                    if (first_set(i, c.x, c.y, ss.levels[i].sigma_size, &idx))
                    {
                        if (c.response > cands[(size_t)W[i][idx] - 1].response)
                            changes.push_back({i, idx, 0});
                        else
                            keep = false;
                    }
                    if (keep)
                        changes.push_back({i, c.y * ss.levels[i].width + c.x, (int)k + 1});
                }
                else
                {
                    const size_t j = pass == 2 ? i - 1 : i + 1;
                    int px, py, r;
                    if (pass == 2)
                    {
                        const int diff = (1 << ss.levels[i].octave) / (1 << ss.levels[i - 1].octave);
                        px = c.x * diff, py = c.y * diff, r = ss.levels[i].sigma_size * diff;
                    }
                    else
                    {
                        const int diff = (1 << ss.levels[i + 1].octave) / (1 << ss.levels[i].octave);
                        px = c.x / diff, py = c.y / diff, r = ss.levels[i + 1].sigma_size;
                    }
                    if (first_set(j, px, py, r, &idx) && c.response > cands[(size_t)W[j][idx] - 1].response)
                        changes.push_back({j, idx, 0});
                }
            }
            for (const change &ch : changes)
                W[ch.lvl][(size_t)ch.idx] = ch.value;
            for (size_t k : ready)
                D[cands[k].level][(size_t)cands[k].y * ss.levels[cands[k].level].width + cands[k].x] = 1;
            todo.swap(later);
        }
        return n_rounds;
    };
    rounds[0] = run_pass(1); // (own = every maximum, W starts empty)
    own = W;
    rounds[1] = run_pass(2);
    own = W;
    rounds[2] = run_pass(3);
    std::vector<char> dead(cands.size(), 1);
    for (size_t i = 0; i < N; i++)
        for (int v : W[i])
            if (v != 0)
                dead[(size_t)v - 1] = 0;
    return dead;
}
// counts[0] = candidates, [1..3] = survivors of the order-free rule / the 3.x list / the 4.x masks, [4] = candidates on which
// the order-free rule and the 3.x list disagree, [5] = ... and the 4.x masks, [6] = the 3.x list and the 4.x masks, [7] = the 4.x
// masks evaluated in rounds against the sequential evaluation (0), [8..10] = rounds the three passes needed
void suppression_census(const ScaleSpace &ss, const Options &o, uint64_t counts[11])
{
    const std::vector<cand> cands = find_candidates(ss, o);
    const std::vector<char> a = suppress_order_free(ss, o, cands), b = suppress_list_3x(ss, o, cands), c = suppress_masks_4x(ss, o, cands);
    int rounds[3];
    const std::vector<char> d = suppress_masks_4x_in_rounds(ss, o, cands, rounds);
    for (int i = 0; i < 11; i++)
        counts[i] = 0;
    for (size_t k = 0; k < cands.size(); k++)
        counts[7] += c[k] != d[k]; // (the rounds form against the sequential one: must be 0)
    for (int p = 0; p < 3; p++)
        counts[8 + p] = (uint64_t)rounds[p];
    counts[0] = cands.size();
    for (size_t k = 0; k < cands.size(); k++)
    {
        counts[1] += !a[k];
        counts[2] += !b[k];
        counts[3] += !c[k];
        counts[4] += a[k] != b[k];
        counts[5] += a[k] != c[k];
        counts[6] += b[k] != c[k];
    }
}

// generateDescriptorSubsample (AKAZEFeatures.cpp) for nbits = 486, pattern_size = 10, 3 channels, with cv::RNG(1024) restated
// (core/operations.hpp: state = (uint64)(unsigned)state * 4164903690U + (unsigned)(state >> 32), the draw is the low word; rng(N) =
// next() % N).  fullM lists the 162 cell pairs (grid i, cell j < cell k; cell j covers [psz (j % g) - 10, +psz) in k and
// [psz (j / g) - 10, +psz) in l); every pick takes one of the rows not yet taken (the first six picks are forced to rows 0 .. 5,
// after the draw), enters its two cells in the sample list if they are new, and the row picked is overwritten by the last live
// one.  comps[3 i + c] = {3 first + c, 3 second + c}: indices into the values, three channels per cell.
struct mldb_subset
{
    int n_samples = 0;
    int samples[29][3];
    int comps[486][2];
};
static const mldb_subset &mldb_subset_tables()
{
    static const mldb_subset T = []() {
        mldb_subset t;
        const int P = 10, nch = 3, nbits = 486;
        int full[162][5], c = 0;
        for (int i = 0; i < 3; i++)
        {
            const int g = i + 2, gsz = g * g, psz = (2 * P + g - 1) / g;
            for (int j = 0; j < gsz; j++)
                for (int k = j + 1; k < gsz; k++, c++)
                {
                    full[c][0] = i;
                    full[c][1] = psz * (j % g) - P;
                    full[c][2] = psz * (j / g) - P;
                    full[c][3] = psz * (k % g) - P;
                    full[c][4] = psz * (k / g) - P;
                }
        }
        uint64_t state = 1024;
        auto next = [&]() {
            state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
            return (uint32_t)state;
        };
        const int npicks = (nbits + nch - 1) / nch;
        int count = 0;
        for (int i = 0; i < npicks; i++)
        {
            int k = (int)(next() % (uint32_t)(162 - i));
            if (i < 6)
                k = i; // "Force use of the coarser grid values and comparisons"
            for (int side = 0; side < 2; side++)
            {
                const int c0 = full[k][0], c1 = full[k][side ? 3 : 1], c2 = full[k][side ? 4 : 2];
                int at = -1;
                for (int j = 0; j < count && at < 0; j++)
                    if (t.samples[j][0] == c0 && t.samples[j][1] == c1 && t.samples[j][2] == c2)
                        at = j;
                if (at < 0)
                {
                    if (count >= 29)
                        std::abort(); // (4 + 9 + 16 cells: cannot happen)
                    at = count++;
                    t.samples[at][0] = c0;
                    t.samples[at][1] = c1;
                    t.samples[at][2] = c2;
                }
                for (int ch = 0; ch < nch; ch++)
                    t.comps[i * nch + ch][side] = nch * at + ch;
            }
            for (int q = 0; q < 5; q++)
                full[k][q] = full[162 - i - 1][q];
        }
        t.n_samples = count;
        return t;
    }();
    return T;
}

std::vector<Keypoint> detect_and_describe(const ScaleSpace &ss, const Options &o)
{
    // 1. per-level 3x3 maxima above the threshold
    const std::vector<cand> cands = find_candidates(ss, o);
    // 2. scale-space suppression: a candidate dies if a stronger one (ties: lower (level,y,x) wins) of
    //    an adjacent level lies within its own size (esigma * derivative_factor, base-image pixels)
    const std::vector<char> dead = suppress_masks_4x(ss, o, cands);
    // 3. sub-pixel fit, orientation, descriptor
    const std::vector<float> gw = orientation_weights();
    std::vector<Keypoint> out;
    for (size_t k = 0; k < cands.size(); k++)
    {
        if (dead[k])
            continue;
        const cand &c = cands[k];
        const Level &l = ss.levels[c.level];
        const int w = l.width, h = l.height;
        const float *D = ss.Ldet[c.level].data();
        auto at = [&](int x, int y) { return D[(size_t)y * w + x]; };
        const float Dx = 0.5f * (at(c.x + 1, c.y) - at(c.x - 1, c.y));
        const float Dy = 0.5f * (at(c.x, c.y + 1) - at(c.x, c.y - 1));
        const float Dxx = (at(c.x + 1, c.y) + at(c.x - 1, c.y)) - 2.0f * at(c.x, c.y);
        const float Dyy = (at(c.x, c.y + 1) + at(c.x, c.y - 1)) - 2.0f * at(c.x, c.y);
        const float Dxy = 0.25f * ((at(c.x + 1, c.y + 1) + at(c.x - 1, c.y - 1)) - (at(c.x - 1, c.y + 1) + at(c.x + 1, c.y - 1)));
        float dx, dy;
        subpixel_solve(Dxx, Dxy, Dyy, Dx, Dy, &dx, &dy);
        if (!(std::fabs(dx) <= 1.0f && std::fabs(dy) <= 1.0f))
            continue;
        const float ratio = (float)(1 << l.octave);
        Keypoint kp;
        kp.x = ((float)c.x + dx) * ratio + 0.5f * (ratio - 1.0f);
        kp.y = ((float)c.y + dy) * ratio + 0.5f * (ratio - 1.0f);
        kp.size = 2.0f * (l.esigma * o.derivative_factor);
        kp.response = c.response;
        kp.level = c.level;
        kp.octave = l.octave;
        const float *Lt = ss.Lt[c.level].data(), *Lx = ss.Lx[c.level].data(), *Ly = ss.Ly[c.level].data();
        const float xf = kp.x / ratio, yf = kp.y / ratio;
        const int s = (int)std::lrintf(0.5f * kp.size / ratio);
        // dominant orientation (OpenCV 4.x's Compute_Main_Orientation, header "STRUCTURE"): 109 samples of a radius-6 disc around the
        // ROUNDED position (Sample_Derivative_Response_Radius6: x0 = cvRound(pt.x / ratio), offsets i * scale), their angles sorted
        // into 42 slices of 2 pi / 42 (quantized_counting_sort: slice (int)(angle / step), out of range -> 0; inside a slice the
        // LATER sample first), and a window of 7 slices (pi / 3) slid over the slices - sums in sorted order, the last six windows
        // wrapping from the last slice to the first
        float resX[109], resY[109], Ang[109];
        {
            const int x0 = (int)std::lrintf(xf), y0 = (int)std::lrintf(yf);
            int idx = 0;
            for (int i = -6; i <= 6; i++)
                for (int j = -6; j <= 6; j++)
                    if (i * i + j * j < 36)
                    {
                        const int iy = clampi(y0 + j * s, 0, h - 1), ix = clampi(x0 + i * s, 0, w - 1);
                        const float g = gw[(i + 6) * 13 + (j + 6)];
                        resX[idx] = g * Lx[(size_t)iy * w + ix];
                        resY[idx] = g * Ly[(size_t)iy * w + ix];
                        Ang[idx] = fast_atan2(resY[idx], resX[idx]);
                        idx++;
                    }
        }
        constexpr int slices = 42, win = 7;
        const float ang_step = (float)(2.0 * 3.14159265358979323846 / slices);
        int slice[slices + 1], sorted_idx[109];
        {
            for (int i = 0; i <= slices; i++)
                slice[i] = 0;
            auto key = [&](int q) {
                const int k = (int)(Ang[q] / ang_step);
                return (k < 0 || k >= slices) ? 0 : k;
            };
            for (int q = 0; q < 109; q++)
                slice[key(q)]++;
            for (int i = 1; i <= slices; i++) // inclusive prefix sums: the slices' ends
                slice[i] += slice[i - 1];
            for (int q = 0; q < 109; q++) // filled from each slice's end downwards; slice[] becomes the slices' starts
                sorted_idx[--slice[key(q)]] = q;
        }
        float maxX = 0.0f, maxY = 0.0f;
        for (int i = slice[0]; i < slice[win]; i++)
        {
            maxX = maxX + resX[sorted_idx[i]];
            maxY = maxY + resY[sorted_idx[i]];
        }
        float best = maxX * maxX + maxY * maxY;
        for (int sn = 1; sn < slices; sn++)
        {
            // (OpenCV skips a window whose contents did not change; it would not be strictly larger than itself either)
            float sumX = 0.0f, sumY = 0.0f;
            const int last = std::min(sn + win, slices), remain = sn + win - slices;
            for (int i = slice[sn]; i < slice[last]; i++)
            {
                sumX = sumX + resX[sorted_idx[i]];
                sumY = sumY + resY[sorted_idx[i]];
            }
            for (int i = slice[0]; remain > 0 && i < slice[remain]; i++)
            {
                sumX = sumX + resX[sorted_idx[i]];
                sumY = sumY + resY[sorted_idx[i]];
            }
            const float m = sumX * sumX + sumY * sumY;
            if (m > best)
                best = m, maxX = sumX, maxY = sumY;
        }
        float angle = cv_fast_atan2_deg(maxY, maxX); // KeyPoint::angle is in degrees
        angle = angle * DEG2RAD_F; // what the descriptor rotates by (and what this interface reports: radians)
        kp.angle = angle;
        // M-LDB, 3 channels, grids 2x2 / 3x3 / 4x4 over [-10, 10) * scale, rotated by the orientation.  The reference asks for
        // descriptor_size = 486 (extract_features.cpp:35), and any descriptor_size other than 0 takes OpenCV's SUBSET path
        // (AKAZEFeatures.cpp: MLDB_Descriptor_Subset_Invoker with generateDescriptorSubsample's tables - header "STRUCTURE"):
        // a cell's value is the SUM over its samples (no mean), the channels are Lt, rx co + ry si, -rx si + ry co, a sample sits
        // at yf + ((l scale) co + (k scale) si), xf + ((-l scale) si + (k scale) co), and bit 3 i + c is comparison picks[i] in
        // channel c - all 162 comparisons, in the order cv::RNG(1024) draws them
        float si, co;
        libm_sincosf(angle, &si, &co);
        std::memset(kp.desc, 0, sizeof kp.desc);
        const mldb_subset &T = mldb_subset_tables();
        float values[29 * 3];
        for (int i = 0; i < T.n_samples; i++)
        {
            const int step = T.samples[i][0] == 0 ? 10 : (T.samples[i][0] == 1 ? 7 : 5);
            float di = 0.0f, dx = 0.0f, dy = 0.0f;
            for (int k = T.samples[i][1]; k < T.samples[i][1] + step; k++)
                for (int l = T.samples[i][2]; l < T.samples[i][2] + step; l++)
                {
                    const float sy = yf + ((float)(l * s) * co + (float)(k * s) * si);
                    const float sx = xf + ((float)(-l * s) * si + (float)(k * s) * co);
                    const int y1 = (int)std::lrintf(sy), x1 = (int)std::lrintf(sx);
                    if (x1 < 0 || y1 < 0 || x1 >= w || y1 >= h)
                        continue;
                    const float rx = Lx[(size_t)y1 * w + x1], ry = Ly[(size_t)y1 * w + x1];
                    di = di + Lt[(size_t)y1 * w + x1];
                    dx = dx + (rx * co + ry * si);
                    dy = dy + (-rx * si + ry * co);
                }
            values[3 * i] = di;
            values[3 * i + 1] = dx;
            values[3 * i + 2] = dy;
        }
        for (int i = 0; i < 486; i++)
            if (values[T.comps[i][0]] > values[T.comps[i][1]])
                kp.desc[i >> 6] |= (uint64_t)1 << (i & 63);
        out.push_back(kp);
    }
    return out;
}

// The tail of extract_features (src/extract/extract_features.cpp:38-87): rescale to full-resolution pixels, the
// unstable std::sort by strength from AKAZE's keypoint order, the 8 px NMS, [sparse..., dense...].
void extract_tail(const std::vector<Keypoint> &kps, double scale, Extracted &ex)
{
    struct feat
    {
        double x, y;
        float strength;
        uint64_t d[8];
    };
    std::vector<feat> f(kps.size());
    for (size_t i = 0; i < kps.size(); i++)
    {
        f[i].x = kps[i].x / scale;
        f[i].y = kps[i].y / scale;
        f[i].strength = kps[i].response;
        std::memcpy(f[i].d, kps[i].desc, 64);
    }
    std::sort(f.begin(), f.end(), [](const feat &a, const feat &b) -> bool { return a.strength > b.strength; });
    // NMS, radius 8 px in the scaled image; note the first keypoint is seeded into the tree AND visited by
    // the loop, so it re-appears at the head of the dense list (faithful to :63-83)
    std::vector<feat> results, dense;
    std::vector<std::pair<double, double>> kept;
    auto nearest2 = [&](double x, double y) {
        double best = std::numeric_limits<double>::infinity();
        for (auto &p : kept)
        {
            const double dx = x - p.first, dy = y - p.second;
            double d = 0;
            d += dx * dx;
            d += dy * dy;
            best = std::min(best, d);
        }
        return best;
    };
    if (!f.empty())
    {
        kept.emplace_back(f[0].x, f[0].y);
        results.push_back(f[0]);
    }
    for (const feat &p : f)
    {
        if (nearest2(p.x, p.y) * (scale * scale) > 64.0)
        {
            kept.emplace_back(p.x, p.y);
            results.push_back(p);
        }
        else
            dense.push_back(p);
    }
    ex.num_sparse = results.size();
    results.insert(results.end(), dense.begin(), dense.end());
    for (const feat &p : results)
    {
        ex.loc.push_back(p.x);
        ex.loc.push_back(p.y);
        ex.strength.push_back(p.strength);
        ex.desc.insert(ex.desc.end(), p.d, p.d + 8);
    }
}

Extracted extract_features(const uint8_t *bgr, int w, int h) // src/extract/extract_features.cpp:11-88
{
    Extracted ex;
    if (w <= 0 || h <= 0)
        return ex;
    std::vector<uint8_t> gray((size_t)w * h);
    bgr_to_gray(bgr, w, h, gray.data());
    const double scale = std::min(1.f, float(1600) / (float)std::max(w, h));
    const int sw = (int)std::lrint(w * scale), sh = (int)std::lrint(h * scale);
    std::vector<uint8_t> small((size_t)sw * sh);
    resize_area(gray.data(), w, h, small.data(), sw, sh, scale);
    std::vector<float> img((size_t)sw * sh);
    for (size_t i = 0; i < img.size(); i++)
        img[i] = (float)small[i] * (1.0f / 255.0f);
    Options o;
    const ScaleSpace ss = build_scale_space(img, sw, sh, o);
    const std::vector<Keypoint> kps = detect_and_describe(ss, o);

    extract_tail(kps, scale, ex);
    return ex;
}

} // namespace akaze
} // namespace oracle

using namespace oracle::akaze;

extern "C"
{

// the host tail alone, on keypoints given as kp6 rows {x, y, size, angle, response, level} in detection order
size_t oc_extract_tail(const float *kp6, const uint64_t *desc, size_t n, double scale, double *loc, float *strength, uint64_t *desc_out,
                       uint64_t *num_sparse)
{
    std::vector<Keypoint> kps(n);
    for (size_t i = 0; i < n; i++)
    {
        kps[i].x = kp6[6 * i];
        kps[i].y = kp6[6 * i + 1];
        kps[i].response = kp6[6 * i + 4];
        std::memcpy(kps[i].desc, desc + 8 * i, 64);
    }
    Extracted ex;
    extract_tail(kps, scale, ex);
    const size_t m = ex.strength.size();
    std::memcpy(loc, ex.loc.data(), m * 16);
    std::memcpy(strength, ex.strength.data(), m * 4);
    std::memcpy(desc_out, ex.desc.data(), m * 64);
    *num_sparse = ex.num_sparse;
    return m;
}

// keypoints: n x {x, y, size, angle, response, level}; desc n x 8; returns n (capped at max_kp)
size_t oc_akaze(const uint8_t *gray, int w, int h, size_t max_kp, float *kp6, uint64_t *desc, float *kcontrast)
{
    std::vector<float> img((size_t)w * h);
    for (size_t i = 0; i < img.size(); i++)
        img[i] = (float)gray[i] * (1.0f / 255.0f);
    Options o;
    const ScaleSpace ss = build_scale_space(img, w, h, o);
    if (kcontrast)
        *kcontrast = ss.kcontrast;
    const std::vector<Keypoint> kps = detect_and_describe(ss, o);
    const size_t n = std::min(kps.size(), max_kp);
    for (size_t i = 0; i < n; i++)
    {
        kp6[6 * i] = kps[i].x;
        kp6[6 * i + 1] = kps[i].y;
        kp6[6 * i + 2] = kps[i].size;
        kp6[6 * i + 3] = kps[i].angle;
        kp6[6 * i + 4] = kps[i].response;
        kp6[6 * i + 5] = (float)kps[i].level;
        std::memcpy(desc + 8 * i, kps[i].desc, 64);
    }
    return kps.size();
}

// census of the suppression rules on one grey image (suppression_census): counts11 as documented there
void oc_akaze_suppression_census(const uint8_t *gray, int w, int h, uint64_t *counts11)
{
    std::vector<float> img((size_t)w * h);
    for (size_t i = 0; i < img.size(); i++)
        img[i] = (float)gray[i] * (1.0f / 255.0f);
    Options o;
    const ScaleSpace ss = build_scale_space(img, w, h, o);
    suppression_census(ss, o, counts11);
}

// level images for stage-by-stage parity: which = 0 Lt, 1 Lx, 2 Ly, 3 Ldet; returns w*h of the level
size_t oc_akaze_level(const uint8_t *gray, int w, int h, int level, int which, float *out, int *lw, int *lh)
{
    std::vector<float> img((size_t)w * h);
    for (size_t i = 0; i < img.size(); i++)
        img[i] = (float)gray[i] * (1.0f / 255.0f);
    Options o;
    const ScaleSpace ss = build_scale_space(img, w, h, o);
    if (level < 0 || level >= (int)ss.levels.size())
        return 0;
    const std::vector<float> &src = which == 0 ? ss.Lt[level] : which == 1 ? ss.Lx[level] : which == 2 ? ss.Ly[level] : ss.Ldet[level];
    std::memcpy(out, src.data(), src.size() * 4);
    *lw = ss.levels[level].width;
    *lh = ss.levels[level].height;
    return src.size();
}

size_t oc_extract_features(const uint8_t *bgr, int w, int h, size_t max_n, double *loc, float *strength, uint64_t *desc,
                           uint64_t *num_sparse)
{
    const Extracted ex = extract_features(bgr, w, h);
    const size_t n = std::min(ex.strength.size(), max_n);
    std::memcpy(loc, ex.loc.data(), n * 16);
    std::memcpy(strength, ex.strength.data(), n * 4);
    std::memcpy(desc, ex.desc.data(), n * 64);
    *num_sparse = ex.num_sparse;
    return ex.strength.size();
}

void oc_gray_resize(const uint8_t *bgr, int w, int h, uint8_t *out, int ow, int oh)
{
    std::vector<uint8_t> gray((size_t)w * h);
    bgr_to_gray(bgr, w, h, gray.data());
    // (the working image of extract_features: cv::resize is called with fx = fy = the float scale, ow x oh is what it derives)
    const double scale = std::min(1.f, float(1600) / (float)std::max(w, h));
    resize_area(gray.data(), w, h, out, ow, oh, scale);
}

} // extern "C"
