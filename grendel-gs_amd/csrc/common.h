// common.h -- shared device/host helpers of the gfx950 rasterizer library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsraster.h"

#define GSR_WAVE 64

#define GSR_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

#define GSR_HIP(expr)                              \
    do {                                           \
        hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

static inline int gsr_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// tile rect [min,max) of (pixel centre, radius): C truncation, clamped to the grid
// (restates SURVEY.md A.2 step 7; identical in preprocess, K2 and K3 so that the three agree).
__device__ __forceinline__ void gsr_get_rect(float px, float py, int radius, int gx, int gy, int &minx, int &miny,
                                             int &maxx, int &maxy) {
    const float r = (float)radius;
    minx = min(gx, max(0, (int)((px - r) / GSR_BLOCK_X)));
    miny = min(gy, max(0, (int)((py - r) / GSR_BLOCK_Y)));
    maxx = min(gx, max(0, (int)((px + r + (GSR_BLOCK_X - 1)) / GSR_BLOCK_X)));
    maxy = min(gy, max(0, (int)((py + r + (GSR_BLOCK_Y - 1)) / GSR_BLOCK_Y)));
}

// Half extents (in pixels) of the axis-aligned box around the region where a Gaussian's
// alpha = min(0.99, o * exp(power)) can reach 1/255, i.e. power >= -ln(255 o): an ellipse with half
// extents sqrt(2 tau cov_xx), sqrt(2 tau cov_yy), cov = conic^-1.  Returns false when alpha can
// never reach 1/255 (o < 1/255; exact, the blend kernels compare against the same constant).  The
// extents are CONSERVATIVE: +1 % and +0.5 px cover the rounding of det (cancellation for very
// anisotropic splats); an unusable det gives "unbounded".  Shared by K3 (tile rect) and K8/K10
// (quadrant test) so that the two can only ever skip work that contributes exactly nothing.
__device__ __forceinline__ bool gsr_alpha_extent(const float4 co, float &ex, float &ey) {
    ex = ey = 1e30f;
    if (!(co.w >= 1.0f / 255.0f)) return false;
    const float tau = __logf(255.0f * co.w);
    const float det = co.x * co.z - co.y * co.y;
    if (det > 0.f && tau >= 0.f) {
        const float s = 2.0f * tau / det;
        ex = sqrtf(s * co.z) * 1.01f + 0.5f;
        ey = sqrtf(s * co.x) * 1.01f + 0.5f;
    }
    return true;
}

// ---- internal launchers (defined in the .hip files, called from api.hip) --------------------
int gsr_launch_preprocess_forward(int P, int D, int M, const float *means3D, const float *scales, float scale_modifier,
                                  const float *rotations, const float *shs, const float *opacities,
                                  const float *viewmatrix, const float *projmatrix, const float *campos, int W, int H,
                                  float tanfovx, float tanfovy, float *means2D, float *depths, int32_t *radii,
                                  float *cov3D, float *conic_opacity, float *rgb, uint8_t *clamped,
                                  hipStream_t stream);
int gsr_launch_preprocess_backward(int P, int D, int M, const float *means3D, const float *scales,
                                   float scale_modifier, const float *rotations, const float *shs,
                                   const float *viewmatrix, const float *projmatrix, const float *campos, int W,
                                   int H, float tanfovx, float tanfovy, const int32_t *radii, const float *cov3D,
                                   const uint8_t *clamped, const float *dL_dmeans2D, const float *dL_dconic_opacity,
                                   const float *dL_drgb, float *dL_dmeans3D, float *dL_dscales, float *dL_drotations,
                                   float *dL_dshs, float *dL_dopacities, hipStream_t stream);
int gsr_launch_local2j(int P, int W, int H, int ws, const float *means2D, const int32_t *radii, const int32_t *div,
                       uint8_t *out, hipStream_t stream);

// device-wide primitives (binning.hip)
size_t gsr_scan_temp_bytes(long long n);
int gsr_exclusive_scan_u32(const uint32_t *in, uint32_t *out, long long n, void *temp, hipStream_t stream);
size_t gsr_radix_temp_bytes(long long n);
// stable LSD radix sort of (key,value) u32 pairs on key bits [bit_lo, bit_hi); ping-pongs between
// (k0,v0) and (k1,v1); *result_in_first tells where the sorted data ended up.
int gsr_radix_sort_pairs(uint32_t *k0, uint32_t *v0, uint32_t *k1, uint32_t *v1, long long n, int bit_lo, int bit_hi,
                         void *temp, int *result_in_first, hipStream_t stream);

int gsr_launch_composite_forward(int P, int W, int H, const int32_t *ranges, const uint32_t *point_list,
                                 const float *means2D, const float *conic_opacity, const float *rgb,
                                 const uint8_t *compute_locally, const float *bg, float *out_color, float *final_T,
                                 int32_t *n_contrib, hipStream_t stream);
int gsr_launch_composite_backward(int P, int W, int H, const int32_t *ranges, const uint32_t *point_list,
                                  const float *means2D, const float *conic_opacity, const float *rgb,
                                  const uint8_t *compute_locally, const float *bg, const float *final_T,
                                  const int32_t *n_contrib, const float *dL_dpixels, float *dL_dmeans2D,
                                  float *dL_dconic_opacity, float *dL_drgb, hipStream_t stream);
