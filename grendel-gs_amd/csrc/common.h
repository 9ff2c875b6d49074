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

// Workgroup b is dispatched to XCD b % 8 (each XCD has its own 4 MiB L2).  Neighbouring tiles share most of their
// inputs (the Gaussians of a tile list, the halo of a loss tile), so hand every XCD a CONTIGUOUS span of tile ids
// instead of every 8th tile: the re-reads of one span then hit one L2.  Bijective for any tile count
// (cdna_hip_programming.md section 5).
__device__ __forceinline__ int gsr_xcd_span_of_block(int b, int nwg) {
#ifdef GSR_NO_XCD_MAP
    return b;
#else
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
#endif
}

// The same for a launch in which only the tile rows [lo, hi) (a Grendel row band: the hull of the mask) carry work and
// the other tiles exit at once: block -> tile such that EVERY XCD gets a contiguous 1/8 of the BAND (and a contiguous
// 1/8 of the rest).  With the plain span map a band that is 1/8 of the image is exactly one XCD's span: 32 of the 256
// CUs would composite the whole band (measured on one rank of an 8-rank partition: K8 / K10 as slow as for the whole
// image).  hull = number of tiles in the band, first = first tile of the band, nwg = all tiles.  Bijective; equal
// to gsr_xcd_span_of_block when the band is the whole grid.
__device__ __forceinline__ int gsr_xcd_span_of_block_band(int b, int nwg, int first, int hull) {
#ifdef GSR_NO_XCD_MAP
    return b;
#else
    const int xcd = b & 7, j = b >> 3;
    const int hl = (hull - xcd + 7) >> 3;                       // band tiles of this XCD ...
    const int hs = xcd * (hull >> 3) + min(xcd, hull & 7);      // ... starting at this band-local index
    if (j < hl) return first + hs + j;
    const int ts = xcd * (nwg >> 3) + min(xcd, nwg & 7);        // blocks of the XCDs before this one
    const int n = (ts - hs) + (j - hl);                         // index among the tiles outside the band
    return n < first ? n : n + hull;
#endif
}

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

// Exact (up to a conservative tolerance) test "can this Gaussian reach alpha >= 1/255 on some pixel
// centre inside the box [x0,x1] x [y0,y1]?": alpha >= 1/255 <=> power >= -tau with tau = ln(255 o) and
// power(d) = -1/2 (A dx^2 + 2 B dx dy + C dy^2) concave, so the question is whether the MINIMUM of the
// quadratic form over the box is <= 2 tau.  The centre inside the box gives 0; otherwise the minimum
// lies on the boundary and is the smallest of the four edges' 1-D minima (vertex clamped to the edge).
// Tolerance: 2 tau is inflated by 0.2 % + 0.01 (alpha error < 1 % of 1/255) -- only ever keeps more.
__device__ __forceinline__ float gsr_edge_min_q(float a, float b, float c, float rc, float u, float v0, float v1) {
    // min over v in [v0,v1] of a u^2 + 2 b u v + c v^2   (c > 0, rc ~ 1 / c: the vertex only has to be NEAR the
    // minimiser -- the form is flat there -- so a 1-ulp reciprocal replaces four IEEE divisions per entry)
    const float vs = fminf(v1, fmaxf(v0, -b * u * rc));
    return a * u * u + (2.0f * b * u + c * vs) * vs;
}
__device__ __forceinline__ bool gsr_can_touch_box(const float2 xy, const float4 co, float x0, float y0, float x1,
                                                  float y1) {
    if (!(co.w >= 1.0f / 255.0f)) return false;
    // 2 ln(255 o) via the hardware log2 (v_log_f32; 255 o >= 1, no denormal path needed)
    const float lim = (2.0f * 0.6931471805599453f) * __builtin_amdgcn_logf(255.0f * co.w) * 1.002f + 0.01f;
    // box in coordinates relative to the centre: d = pixel - centre (sign irrelevant for the form)
    const float l = x0 - xy.x, r = x1 - xy.x, t = y0 - xy.y, bt = y1 - xy.y;
    if (l <= 0.f && r >= 0.f && t <= 0.f && bt >= 0.f) return true;
    if (!(co.x > 0.f && co.z > 0.f)) return true;  // not a proper conic: do not cull
    const float rcz = __builtin_amdgcn_rcpf(co.z), rcx = __builtin_amdgcn_rcpf(co.x);
    float q = gsr_edge_min_q(co.x, co.y, co.z, rcz, l, t, bt);           // edge x = x0
    q = fminf(q, gsr_edge_min_q(co.x, co.y, co.z, rcz, r, t, bt));       // edge x = x1
    q = fminf(q, gsr_edge_min_q(co.z, co.y, co.x, rcx, t, l, r));        // edge y = y0
    q = fminf(q, gsr_edge_min_q(co.z, co.y, co.x, rcx, bt, l, r));       // edge y = y1
    return !(q > lim);  // NaN -> keep
}

// One element of the dense Adam update (torch.optim.Adam as the reference configures it, scene/gaussian_model.py:292:
// no weight decay, no amsgrad).  Spelled with explicit roundings so that optim.hip (contraction on) and the fused
// K11 + Adam kernel of preprocess.hip (contraction off) produce the SAME bits: the fused step is tested for equality
// with K11 followed by gsr_adam_step_multi.  omb = 1 - beta rounded from double, as the stock optimizer's scalars are.
__device__ __forceinline__ void gsr_adam1(float &p, float g, float &m, float &v, float lr_c, float b1, float b2,
                                          float omb1, float omb2, float inv_sqrt_bc2, float eps) {
    m = __builtin_fmaf(b1, m, __fmul_rn(omb1, g));
    v = __builtin_fmaf(b2, v, __fmul_rn(__fmul_rn(omb2, g), g));
    const float denom = __builtin_fmaf(__fsqrt_rn(v), inv_sqrt_bc2, eps);
    p = __builtin_fmaf(-lr_c, __fdiv_rn(m, denom), p);
}

// ---- internal launchers (defined in the .hip files, called from api.hip) --------------------
int gsr_launch_preprocess_forward(int P, int D, int M, const float *means3D, const float *scales, float scale_modifier,
                                  const float *rotations, const float *shs, const float *shs_rest,
                                  const float *opacities,
                                  const float *viewmatrix, const float *projmatrix, const float *campos, int W, int H,
                                  float tanfovx, float tanfovy, float *means2D, float *depths, int32_t *radii,
                                  float *cov3D, float *conic_opacity, float *rgb, uint8_t *clamped,
                                  hipStream_t stream);
int gsr_launch_preprocess_backward(int P, int D, int M, const float *means3D, const float *scales,
                                   float scale_modifier, const float *rotations, const float *shs,
                                   const float *shs_rest, const float *opacities_raw, const float *viewmatrix, const float *projmatrix, const float *campos, int W,
                                   int H, float tanfovx, float tanfovy, const int32_t *radii, const float *cov3D,
                                   const uint8_t *clamped, const float *dL_dmeans2D, const float *dL_dconic_opacity,
                                   const float *dL_drgb, int grad_row_stride, float *dL_dmeans3D, float *dL_dscales, float *dL_drotations,
                                   float *dL_dshs, float *dL_dshs_rest, float *dL_dopacities, hipStream_t stream);
int gsr_launch_local2j(int P, int W, int H, int ws, const float *means2D, const int32_t *radii, const int32_t *div,
                       uint8_t *out, hipStream_t stream);

int gsr_launch_composite_forward(int P, int W, int H, const int32_t *ranges, const uint32_t *point_list,
                                 const float *means2D, const float *conic_opacity, const float *rgb,
                                 const uint8_t *compute_locally, const float *bg, float *out_color, float *final_T,
                                 int32_t *n_contrib, void *seg_ws, size_t seg_bytes, int row_lo, int row_hi,
                                 void *zero_ptr, size_t zero_bytes, hipStream_t stream);
size_t gsr_composite_seg_bytes(int W, int H);
int gsr_launch_composite_backward(int P, int W, int H, const int32_t *ranges, const uint32_t *point_list,
                                  const float *means2D, const float *conic_opacity, const float *rgb,
                                  const uint8_t *compute_locally, const float *bg, const float *final_T,
                                  const int32_t *n_contrib, const float *dL_dpixels, float *dL_record,
                                  const float *out_color, void *seg_ws, size_t seg_bytes, int row_lo, int row_hi,
                                  int record_is_zero, hipStream_t stream);
