// ORACLE — test infrastructure only (see oracle.hpp).
//
// Restatement of the extract step: src/extract/extract_features.cpp:11-88 and, behind its one call
// `cv::AKAZE::create(DESCRIPTOR_MLDB, 486, 3, 0.00005f)->detectAndCompute` (:35-36), the AKAZE algorithm
// [3P: OpenCV features2d, not under /root/reference and absent from this image].  AKAZE is restated
// from its publication (P. F. Alcantarilla, J. Nuevo, A. Bartoli, "Fast Explicit Diffusion for
// Accelerated Features in Nonlinear Scale Spaces", BMVC 2013) following the structure of OpenCV's
// implementation: FED nonlinear scale space with Perona-Malik g2 conductivity, 4 octaves x 4
// sublevels, Scharr derivatives at the level's integer scale, scale-normalised Hessian determinant,
// scale-space extrema + sub-pixel fit, dominant orientation, 3-channel M-LDB over 2x2 + 3x3 + 4x4
// grids = (6 + 36 + 120) * 3 = 486 bits.
//
// PARITY UNPINNED: there is no OpenCV here to compare with and the reference's extract tests need the
// absent test images and only assert counts (test/test_extract_features.cpp:8-75).  Everything below
// is therefore a self-consistent definition; the GPU implementation is checked against *this* code.
// Arithmetic is float32 with plain mul/add in the written order (no FMA) so that a device
// implementation can be bit-identical; every transcendental (Gaussian taps, FED step sizes, the
// orientation weights) is tabulated on the host in double and handed to both sides.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace oracle
{
namespace akaze
{

struct Options // AKAZEOptions defaults + the arguments of extract_features.cpp:35
{
    int omax = 4, nsublevels = 4;
    float soffset = 1.6f, derivative_factor = 1.5f, sderivatives = 1.0f;
    float dthreshold = 0.00005f;
    float kcontrast_percentile = 0.7f;
    int kcontrast_nbins = 300;
    int descriptor_pattern_size = 10;
};

struct Level
{
    int octave, sublevel, width, height, sigma_size;
    float esigma, etime;
    std::vector<float> tsteps; // FED step sizes that take the previous level to this one
};

struct Keypoint
{
    float x, y;     // pixels of the working image
    float size;     // diameter
    float angle;    // radians in [0, 2 pi)
    float response; // Hessian determinant at the detection level
    int level, octave;
    uint64_t desc[8]; // 486 bits, bit j = word j>>6, bit j&63 (the packing of extract_features.cpp:47-51)
};

// tables shared with the device implementation
std::vector<float> gaussian_kernel(float sigma);                 // odd length, normalised, cv::GaussianBlur sizing
std::vector<Level> make_levels(int width, int height, const Options &o);
void fed_tau_by_process_time(float T, int M, float tau_max, bool reordering, std::vector<float> &tau);
// the float functions of the orientation and the descriptor (akaze.cpp: D2, D3): cv::fastAtan2 in degrees, glibc's sinf / cosf
// restated, cv::solve's 2 x 2 fast path
float cv_fast_atan2_deg(float y, float x);
float libm_sinf(float x);
float libm_cosf(float x);
void subpixel_solve(float Dxx, float Dxy, float Dyy, float Dx, float Dy, float *dx, float *dy);
std::vector<float> orientation_weights();                        // 13 x 13 Gaussian (sigma 2.5), SURF's gauss25

// building blocks (float images, row-major)
void bgr_to_gray(const uint8_t *bgr, int w, int h, uint8_t *gray);
void resize_area(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, double inv_scale); // inv_scale: cv::resize's fx = fy
float compute_k_percentile(const std::vector<float> &img, int w, int h, const Options &o);

struct ScaleSpace
{
    std::vector<Level> levels;
    std::vector<std::vector<float>> Lt, Lx, Ly, Ldet;
    float kcontrast = 0;
};
ScaleSpace build_scale_space(const std::vector<float> &img, int w, int h, const Options &o);
std::vector<Keypoint> detect_and_describe(const ScaleSpace &ss, const Options &o);
// census of the suppression rules (akaze.cpp): OpenCV 4.x's passes (what detect_and_describe runs), their evaluation in rounds,
// the order-free rule of rounds 2 - 5 and the 3.x running list as recalled
void suppression_census(const ScaleSpace &ss, const Options &o, uint64_t counts[11]);

// extract_features(cv::Mat) restated: gray, INTER_AREA downscale to max side 1600, AKAZE, strength sort,
// 8 px NMS, [sparse..., dense...].  Outputs feature arrays (locations in full-resolution pixels).
struct Extracted
{
    std::vector<double> loc;      // n x 2
    std::vector<float> strength;  // n
    std::vector<uint64_t> desc;   // n x 8
    size_t num_sparse = 0;
};
Extracted extract_features(const uint8_t *bgr, int w, int h);

} // namespace akaze
} // namespace oracle
