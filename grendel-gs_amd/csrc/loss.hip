// loss.hip -- fused band-local L1 + SSIM loss (forward and backward) for gfx950.
//
// Replaces the stock-PyTorch sequence behind final_system_loss_computation
// (gaussian_renderer/loss_distribution.py:2536-2585 -> utils/loss_utils.py:88-132 of the reference):
// 5 depthwise 11x11 conv2d per image pair + ~15 elementwise kernels forward, the same again backward
// (MIOpen runs them at ~6 ms each at 1080p on this chip), by two kernels that stream the band once.
//
// Math (identical to the reference): window = outer product of the fp32-normalised 11-tap Gaussian
// (sigma 1.5), zero padding at the band edges (no halo rows from neighbouring bands),
//   mu1 = w*x, mu2 = w*y, s1 = w*x^2 - mu1^2, s2 = w*y^2 - mu2^2, s12 = w*xy - mu1 mu2,
//   ssim = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)),  C1 = 0.01^2, C2 = 0.03^2,
// y = uint8 ground truth / 255.  The forward also stores the three partial-derivative maps
// d ssim/d mu1, d ssim/d(w*x^2), d ssim/d(w*xy); the backward convolves them with the same window:
//   d/dx_j sum_i ssim_i = (w * M1)_j + 2 x_j (w * M2)_j + y_j (w * M3)_j .
// A 32x32 output tile per 512-thread workgroup, 42x42 halo tile staged in LDS, separable passes.  The halo is
// fetched as ALIGNED 16-byte vectors (12 per row cover [ox - 8, ox + 40): one load instruction per thread and map
// instead of ~3.5 scalar ones with their div / mod address arithmetic) whenever the width is a multiple of 4 and
// the base pointers allow it, and the workgroup -> tile map hands each XCD a contiguous span of tiles so that
// the halo re-reads of neighbouring tiles hit that XCD's L2.
#include "common.h"

namespace {

__constant__ const float WIN[11] = {1.0283801239e-03f, 7.5987582095e-03f, 3.6000773311e-02f, 1.0936068743e-01f,
                                    2.1300552785e-01f, 2.6601171494e-01f, 2.1300552785e-01f, 1.0936068743e-01f,
                                    3.6000773311e-02f, 7.5987582095e-03f, 1.0283801239e-03f};
constexpr int LT = 512;                  // threads per workgroup (8 waves at the same LDS: measured best of 256/512/1024)
constexpr int VO = 1024 / LT;            // vertical outputs per thread (4 at 256 threads, 2 at 512)
constexpr int TW = 32, TH = 32;          // output tile of one workgroup
constexpr int HW = TW + 10, HH = TH + 10;  // halo tile
constexpr int HSTR = TW + 8;             // row stride of the horizontal-pass results (4 rows = 32 banks apart)
constexpr int TS = TW;                   // (partials are per workgroup: see gsr_l1_ssim_num_partials)
constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;

__device__ __forceinline__ float block_sum(float v, float *smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < LT / 64; w++) r += smem[w];
    __syncthreads();
    return r;
}

// Both kernels are separable 11-tap convolutions out of LDS with REGISTER sliding windows: a thread produces
// 4 adjacent outputs from 14 loaded values (horizontal: 4 columns of a row, vertical: VO rows of a column), i.e.
// 3.5 LDS reads per output and map instead of 11.  The forward is VALU-bound (measured: SQ_ACTIVE_INST_VALU 0.86 of
// the SIMD cycles, 30 M instructions per 1080p launch), so its five maps are carried as PACKED pairs -- (x, y) and
// (x^2, y^2) in v_pk_mul_f32 / v_pk_fma_f32, the products formed once per loaded element instead of once per tap:
// 3 instructions per tap and output instead of 7.
typedef float v2f __attribute__((ext_vector_type(2)));

// workgroup -> (channel, tile row, tile column); tile id = the index of the workgroup's partial sums
__device__ __forceinline__ int loss_tile(int gxT, int gyT, int &c, int &ox, int &oy, int nwg) {
    const int lin = gsr_xcd_span_of_block(blockIdx.x, nwg);
    c = lin / (gxT * gyT);
    const int rem = lin - c * (gxT * gyT);
    const int by = rem / gxT;
    ox = (rem - by * gxT) * TW;
    oy = by * TH;
    return lin;
}
// A launch whose band is DEVICE data (gsr_l1_ssim_*_band: one captured launch serves every band of a camera): `rows` is
// then the capacity the buffers were sized for -- ground truth and maps keep the capacity's channel stride -- and
// band = { first pixel row, end row } of the image.  The workgroups are mapped over the band's own tiles, so tile ids and
// the order in which the finalize adds the partial sums are those of a launch sized for the band; the workgroups
// above that count publish zero sums.  -> false: nothing to do for this workgroup
__device__ __forceinline__ bool loss_band(const int *__restrict__ band, int rows_cap, int W, int gxT, int &rows, int &gyT,
                                          int &nwg, long long &row_off) {
    nwg = gridDim.x;
    row_off = 0;
    if (!band) return true;
    const int channels = nwg / (gxT * gyT);
    const int y0 = band[0], y1 = band[1];
    rows = max(0, min(y1 - y0, rows_cap));
    gyT = (rows + TH - 1) / TH;
    nwg = channels * gxT * gyT;
    row_off = (long long)y0 * W;
    return (int)blockIdx.x < nwg;
}
constexpr int HV = (TW + 16) / 4;  // aligned 4-element vectors per halo row: columns [ox - 8, ox + TW + 8)
static_assert(HH * HV <= LT, "one halo vector per thread");

template <bool VEC>
__global__ void __launch_bounds__(LT)
l1_ssim_forward_kernel(int rows, int W, int gxT, int gyT, const float *__restrict__ image, long long img_cstride,
                       const uint8_t *__restrict__ gt, float *__restrict__ partials, float *__restrict__ M1,
                       float *__restrict__ M2, float *__restrict__ M3, const int *__restrict__ band) {
    __shared__ v2f sXY[HH][HW + 1];     // (x, y): rendered band / ground truth
    __shared__ v2f hAB[HH][HSTR];       // horizontal pass of (x, y)
    __shared__ v2f hCD[HH][HSTR];       // ... of (x^2, y^2)
    __shared__ float hE[HH][HSTR];      // ... of x y
    __shared__ float red[LT / 64];
    int c, ox, oy, nwg;
    long long row_off;
    const int rows_cap = rows;
    if (!loss_band(band, rows_cap, W, gxT, rows, gyT, nwg, row_off)) {
        if (threadIdx.x < 2) partials[2 * (size_t)blockIdx.x + threadIdx.x] = 0.f;
        return;
    }
    const int tile_id = loss_tile(gxT, gyT, c, ox, oy, nwg);
    const int tid = threadIdx.x;
    const float *img_c = image + (long long)c * img_cstride + row_off;
    const uint8_t *gt_c = gt + (size_t)c * rows_cap * W;
    if (VEC) {
        if (tid < HH * HV) {
            const int ly = tid / HV, q = tid - ly * HV;
            const int gy = oy + ly - 5, gx0 = ox - 8 + 4 * q;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            uchar4 y = make_uchar4(0, 0, 0, 0);
            if (gy >= 0 && gy < rows && gx0 >= 0 && gx0 < W) {  // W % 4 == 0: a vector is inside or outside as a whole
                x = *reinterpret_cast<const float4 *>(img_c + (size_t)gy * W + gx0);
                y = *reinterpret_cast<const uchar4 *>(gt_c + (size_t)gy * W + gx0);
            }
            const float xs[4] = {x.x, x.y, x.z, x.w};
            const float ys[4] = {(float)y.x, (float)y.y, (float)y.z, (float)y.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int lx = 4 * q - 3 + i;
                if (lx >= 0 && lx < HW) sXY[ly][lx] = v2f{xs[i], ys[i] * (1.0f / 255.0f)};
            }
        }
    } else {
        for (int idx = tid; idx < HH * HW; idx += LT) {
            const int ly = idx / HW, lx = idx % HW;
            const int gy = oy + ly - 5, gx = ox + lx - 5;
            v2f v = {0.f, 0.f};
            if (gy >= 0 && gy < rows && gx >= 0 && gx < W) {
                v.x = img_c[(size_t)gy * W + gx];
                v.y = (float)gt_c[(size_t)gy * W + gx] * (1.0f / 255.0f);
            }
            sXY[ly][lx] = v;
        }
    }
    __syncthreads();
    for (int task = tid; task < HH * (TW / 4); task += LT) {
        const int r = task / (TW / 4), cx0 = (task % (TW / 4)) * 4;
        v2f xy[14], sq[14];
        float pr[14];
#pragma unroll
        for (int i = 0; i < 14; i++) {
            xy[i] = sXY[r][cx0 + i];
            sq[i] = xy[i] * xy[i];
            pr[i] = xy[i].x * xy[i].y;
        }
#pragma unroll
        for (int o = 0; o < 4; o++) {
            v2f a = {0.f, 0.f}, b = {0.f, 0.f};
            float e = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = WIN[k];
                a += xy[o + k] * w;
                b += sq[o + k] * w;
                e += pr[o + k] * w;
            }
            hAB[r][cx0 + o] = a;
            hCD[r][cx0 + o] = b;
            hE[r][cx0 + o] = e;
        }
    }
    __syncthreads();
    const int tx = tid % TW, ty0 = (tid / TW) * VO;
    v2f vab[VO], vcd[VO];
    float ve[VO];
    {
        v2f p[10 + VO], q[10 + VO];
        float t[10 + VO];
#pragma unroll
        for (int i = 0; i < 10 + VO; i++) {
            p[i] = hAB[ty0 + i][tx];
            q[i] = hCD[ty0 + i][tx];
            t[i] = hE[ty0 + i][tx];
        }
#pragma unroll
        for (int o = 0; o < VO; o++) {
            v2f a = {0.f, 0.f}, b = {0.f, 0.f};
            float e = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = WIN[k];
                a += p[o + k] * w;
                b += q[o + k] * w;
                e += t[o + k] * w;
            }
            vab[o] = a;
            vcd[o] = b;
            ve[o] = e;
        }
    }
    const int gx = ox + tx;
    float l1 = 0.f, ssim_sum = 0.f;
#pragma unroll
    for (int o = 0; o < VO; o++) {
        const int ty = ty0 + o, gy = oy + ty;
        if (gy < rows && gx < W) {
            const float mu1 = vab[o].x, mu2 = vab[o].y, e11 = vcd[o].x, e22 = vcd[o].y, e12 = ve[o];
            const v2f cxy = sXY[ty + 5][tx + 5];
            const float x = cxy.x, y = cxy.y;
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
            const float A = 2.f * mu12 + SSIM_C1, B = 2.f * s12 + SSIM_C2;
            const float Cd = mu1_sq + mu2_sq + SSIM_C1, Dd = s1 + s2 + SSIM_C2;
            const float inv_d = __builtin_amdgcn_rcpf(Dd);   // v_rcp_f32 (1 ulp) instead of two IEEE divisions per
            const float inv = __builtin_amdgcn_rcpf(Cd) * inv_d;  // output: ~20 VALU instructions each; Cd, Dd >= C1, C2 > 0
            const float ssim = A * B * inv;
            ssim_sum += ssim;
            l1 += fabsf(x - y);
            if (M1) {
                const size_t off = ((size_t)c * rows_cap + gy) * W + gx;
                M1[off] = 2.f * mu2 * (B - A) * inv - ssim * 2.f * mu1 * (Dd - Cd) * inv;
                M2[off] = -ssim * inv_d;
                M3[off] = 2.f * A * inv;
            }
        }
    }
    const float sl1 = block_sum(l1, red);
    const float sss = block_sum(ssim_sum, red);
    if (tid == 0) {  // indexed by tile, not by workgroup: the finalize adds them in the same fixed order either way
        partials[2 * (size_t)tile_id] = sl1;
        partials[2 * (size_t)tile_id + 1] = sss;
    }
}

template <bool VEC>
__global__ void __launch_bounds__(LT)
l1_ssim_backward_kernel(int rows, int W, int gxT, int gyT, const float *__restrict__ image, long long img_cstride,
                        const uint8_t *__restrict__ gt, const float *__restrict__ M1, const float *__restrict__ M2,
                        const float *__restrict__ M3, const float *__restrict__ grad_l1_sum,
                        const float *__restrict__ grad_ssim_sum, float scale_l1, float scale_ssim,
                        float *__restrict__ grad_image, long long grad_cstride, const int *__restrict__ band) {
    __shared__ float sM[3][HH][HW + 1];
    __shared__ float hor[3][HH][HSTR];
    int c, ox, oy, nwg;
    long long row_off;
    const int rows_cap = rows;
    if (!loss_band(band, rows_cap, W, gxT, rows, gyT, nwg, row_off)) return;
    loss_tile(gxT, gyT, c, ox, oy, nwg);
    const int tid = threadIdx.x;
    const size_t cbase = (size_t)c * rows_cap * W;
    if (VEC) {
        if (tid < HH * HV) {
            const int ly = tid / HV, q = tid - ly * HV;
            const int gy = oy + ly - 5, gx0 = ox - 8 + 4 * q;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, d = a;
            if (gy >= 0 && gy < rows && gx0 >= 0 && gx0 < W) {
                const size_t o = cbase + (size_t)gy * W + gx0;
                a = *reinterpret_cast<const float4 *>(M1 + o);
                b = *reinterpret_cast<const float4 *>(M2 + o);
                d = *reinterpret_cast<const float4 *>(M3 + o);
            }
            const float as[4] = {a.x, a.y, a.z, a.w}, bs[4] = {b.x, b.y, b.z, b.w}, ds[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int lx = 4 * q - 3 + i;
                if (lx >= 0 && lx < HW) {
                    sM[0][ly][lx] = as[i];
                    sM[1][ly][lx] = bs[i];
                    sM[2][ly][lx] = ds[i];
                }
            }
        }
    } else {
        for (int idx = tid; idx < HH * HW; idx += LT) {
            const int ly = idx / HW, lx = idx % HW;
            const int gy = oy + ly - 5, gx = ox + lx - 5;
            float a = 0.f, b = 0.f, d = 0.f;
            if (gy >= 0 && gy < rows && gx >= 0 && gx < W) {
                const size_t o = cbase + (size_t)gy * W + gx;
                a = M1[o];
                b = M2[o];
                d = M3[o];
            }
            sM[0][ly][lx] = a;
            sM[1][ly][lx] = b;
            sM[2][ly][lx] = d;
        }
    }
    __syncthreads();
    for (int task = tid; task < HH * (TW / 4); task += LT) {
        const int r = task / (TW / 4), cx0 = (task % (TW / 4)) * 4;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            float v[14];
#pragma unroll
            for (int i = 0; i < 14; i++) v[i] = sM[m][r][cx0 + i];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) a += WIN[k] * v[o + k];
                hor[m][r][cx0 + o] = a;
            }
        }
    }
    __syncthreads();
    const int tx = tid % TW, ty0 = (tid / TW) * VO;
    float cv[3][VO];
#pragma unroll
    for (int m = 0; m < 3; m++) {
        float v[10 + VO];
#pragma unroll
        for (int i = 0; i < 10 + VO; i++) v[i] = hor[m][ty0 + i][tx];
#pragma unroll
        for (int o = 0; o < VO; o++) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) a += WIN[k] * v[o + k];
            cv[m][o] = a;
        }
    }
    const int gx = ox + tx;
    if (gx >= W) return;
    const float gl1 = grad_l1_sum[0] * scale_l1, gss = grad_ssim_sum[0] * scale_ssim;
#pragma unroll
    for (int o = 0; o < VO; o++) {
        const int gy = oy + ty0 + o;
        if (gy >= rows) break;
        const float x = image[(long long)c * img_cstride + row_off + (size_t)gy * W + gx];
        const float y = (float)gt[cbase + (size_t)gy * W + gx] * (1.0f / 255.0f);
        const float d = x - y;
        const float sgn = (d > 0.f ? 1.f : 0.f) - (d < 0.f ? 1.f : 0.f);
        grad_image[(long long)c * grad_cstride + row_off + (size_t)gy * W + gx] =
            gl1 * sgn + gss * (cv[0][o] + 2.f * x * cv[1][o] + y * cv[2][o]);
    }
}

// one workgroup: deterministic (fixed-order) reduction of the per-workgroup partial sums, then
//   out[0] = loss = c_l1 * S_l1 + c_ssim * S_ssim + bias,  out[1] = S_l1 * inv_n (Ll1),  out[2] = S_ssim * inv_n (ssim)
__global__ void __launch_bounds__(256) l1_ssim_finalize_kernel(int nb, const float *__restrict__ partials, float c_l1,
                                                                float c_ssim, float bias, float inv_n,
                                                                float *__restrict__ out) {
    __shared__ double red[2][4];
    double a = 0.0, b = 0.0;
    // eight independent 8-byte loads per trip (clamped index, masked value): one workgroup is a pure latency chain --
    // 24 dependent load / add rounds for the 6120 partials of a 1080p image took 8 us
    const float2 *p2 = reinterpret_cast<const float2 *>(partials);
    for (int i0 = threadIdx.x; i0 < nb; i0 += 256 * 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p2[min(i0 + 256 * u, nb - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + 256 * u < nb) {
                a += (double)v[u].x;
                b += (double)v[u].y;
            }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d, 64);
        b += __shfl_xor(b, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s1 = (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        const float s2 = (float)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        out[0] = c_l1 * s1 + c_ssim * s2 + bias;
        out[1] = s1 * inv_n;
        out[2] = s2 * inv_n;
    }
}

}  // namespace

static inline bool aligned_to(const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

extern "C" int gsr_l1_ssim_num_partials(int channels, int rows, int width) {
    if (channels <= 0 || rows <= 0 || width <= 0) return 0;
    return channels * gsr_div_up(rows, TS) * gsr_div_up(width, TS);
}

static int l1_ssim_forward_impl(int channels, int rows, int width, const float *image, int64_t image_channel_stride,
                                const uint8_t *gt, float *partials, float *dm_dmu1, float *dm_dE11, float *dm_dE12,
                                const int *band, gsr_stream_t stream) {
    if (channels <= 0 || rows < 0 || width <= 0) return GSR_EINVAL;
    if (rows == 0) return 0;
    if (!image || !gt || !partials) return GSR_EINVAL;
    if ((dm_dmu1 || dm_dE11 || dm_dE12) && !(dm_dmu1 && dm_dE11 && dm_dE12)) return GSR_EINVAL;
    const int gxT = gsr_div_up(width, TS), gyT = gsr_div_up(rows, TS);
    const dim3 grid(gxT * gyT * channels);
    const bool vec = width % 4 == 0 && image_channel_stride % 4 == 0 && aligned_to(image, 16) && aligned_to(gt, 4);
    if (vec)
        hipLaunchKernelGGL(l1_ssim_forward_kernel<true>, grid, dim3(LT), 0, reinterpret_cast<hipStream_t>(stream), rows,
                           width, gxT, gyT, image, (long long)image_channel_stride, gt, partials, dm_dmu1, dm_dE11,
                           dm_dE12, band);
    else
        hipLaunchKernelGGL(l1_ssim_forward_kernel<false>, grid, dim3(LT), 0, reinterpret_cast<hipStream_t>(stream), rows,
                           width, gxT, gyT, image, (long long)image_channel_stride, gt, partials, dm_dmu1, dm_dE11,
                           dm_dE12, band);
    GSR_LAUNCH_CHECK();
    return 0;
}

extern "C" int gsr_l1_ssim_forward(int channels, int rows, int width, const float *image, int64_t image_channel_stride,
                                   const uint8_t *gt, float *partials, float *dm_dmu1, float *dm_dE11,
                                   float *dm_dE12, gsr_stream_t stream) {
    return l1_ssim_forward_impl(channels, rows, width, image, image_channel_stride, gt, partials, dm_dmu1, dm_dE11,
                                dm_dE12, nullptr, stream);
}

extern "C" int gsr_l1_ssim_forward_band(int channels, int rows_capacity, int width, const float *image,
                                        int64_t image_channel_stride, const uint8_t *gt, float *partials,
                                        float *dm_dmu1, float *dm_dE11, float *dm_dE12, const int32_t *band_rows,
                                        gsr_stream_t stream) {
    if (!band_rows || rows_capacity <= 0) return GSR_EINVAL;
    return l1_ssim_forward_impl(channels, rows_capacity, width, image, image_channel_stride, gt, partials, dm_dmu1,
                                dm_dE11, dm_dE12, band_rows, stream);
}

static int l1_ssim_backward_impl(int channels, int rows, int width, const float *image,
                                 int64_t image_channel_stride, const uint8_t *gt, const float *dm_dmu1,
                                 const float *dm_dE11, const float *dm_dE12, const float *grad_l1_sum,
                                 const float *grad_ssim_sum, float scale_l1, float scale_ssim, float *grad_image,
                                 int64_t grad_channel_stride, const int *band, gsr_stream_t stream) {
    if (channels <= 0 || rows < 0 || width <= 0) return GSR_EINVAL;
    if (rows == 0) return 0;
    if (!image || !gt || !dm_dmu1 || !dm_dE11 || !dm_dE12 || !grad_l1_sum || !grad_ssim_sum || !grad_image)
        return GSR_EINVAL;
    const int gxT = gsr_div_up(width, TS), gyT = gsr_div_up(rows, TS);
    const dim3 grid(gxT * gyT * channels);
    const bool vec = width % 4 == 0 && aligned_to(dm_dmu1, 16) && aligned_to(dm_dE11, 16) && aligned_to(dm_dE12, 16);
    if (vec)
        hipLaunchKernelGGL(l1_ssim_backward_kernel<true>, grid, dim3(LT), 0, reinterpret_cast<hipStream_t>(stream), rows,
                           width, gxT, gyT, image, (long long)image_channel_stride, gt, dm_dmu1, dm_dE11, dm_dE12,
                           grad_l1_sum, grad_ssim_sum, scale_l1, scale_ssim, grad_image, (long long)grad_channel_stride,
                           band);
    else
        hipLaunchKernelGGL(l1_ssim_backward_kernel<false>, grid, dim3(LT), 0, reinterpret_cast<hipStream_t>(stream), rows,
                           width, gxT, gyT, image, (long long)image_channel_stride, gt, dm_dmu1, dm_dE11, dm_dE12,
                           grad_l1_sum, grad_ssim_sum, scale_l1, scale_ssim, grad_image, (long long)grad_channel_stride,
                           band);
    GSR_LAUNCH_CHECK();
    return 0;
}

extern "C" int gsr_l1_ssim_backward(int channels, int rows, int width, const float *image,
                                    int64_t image_channel_stride, const uint8_t *gt, const float *dm_dmu1,
                                    const float *dm_dE11, const float *dm_dE12, const float *grad_l1_sum,
                                    const float *grad_ssim_sum, float scale_l1, float scale_ssim, float *grad_image,
                                    int64_t grad_channel_stride, gsr_stream_t stream) {
    return l1_ssim_backward_impl(channels, rows, width, image, image_channel_stride, gt, dm_dmu1, dm_dE11, dm_dE12,
                                 grad_l1_sum, grad_ssim_sum, scale_l1, scale_ssim, grad_image, grad_channel_stride,
                                 nullptr, stream);
}

extern "C" int gsr_l1_ssim_backward_band(int channels, int rows_capacity, int width, const float *image,
                                         int64_t image_channel_stride, const uint8_t *gt, const float *dm_dmu1,
                                         const float *dm_dE11, const float *dm_dE12, const float *grad_l1_sum,
                                         const float *grad_ssim_sum, float scale_l1, float scale_ssim,
                                         float *grad_image, int64_t grad_channel_stride, const int32_t *band_rows,
                                         gsr_stream_t stream) {
    if (!band_rows || rows_capacity <= 0) return GSR_EINVAL;
    return l1_ssim_backward_impl(channels, rows_capacity, width, image, image_channel_stride, gt, dm_dmu1, dm_dE11,
                                 dm_dE12, grad_l1_sum, grad_ssim_sum, scale_l1, scale_ssim, grad_image,
                                 grad_channel_stride, band_rows, stream);
}

extern "C" int gsr_l1_ssim_finalize(int num_partials, const float *partials, float c_l1, float c_ssim, float bias,
                                    float inv_n, float *out3, gsr_stream_t stream) {
    if (num_partials < 0 || !out3 || (num_partials > 0 && !partials) || !aligned_to(partials, 8)) return GSR_EINVAL;
    hipLaunchKernelGGL(l1_ssim_finalize_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       num_partials, partials, c_l1, c_ssim, bias, inv_n, out3);
    GSR_LAUNCH_CHECK();
    return 0;
}
