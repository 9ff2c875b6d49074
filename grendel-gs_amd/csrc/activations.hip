// activations.hip -- fused parameter activations for gfx950.
//
// GaussianModel.get_scaling / get_rotation / get_opacity / get_features of the reference
// (scene/gaussian_model.py:109-129) are four stock-torch kernels forward (exp, normalize = norm + clamp +
// div, sigmoid, cat) and about ten backward; they touch every Gaussian's 59 floats several times per
// iteration outside the rasterizer op.  Here: one streaming kernel each way, one lane per Gaussian.
//   scales    = exp(_scaling)                       d_scaling  = g * scales
//   rotations = _rotation / max(|_rotation|, 1e-12) d_rotation = (g - y (y . g)) / max(|x|, 1e-12)
//   opacities = sigmoid(_opacity)                   d_opacity  = g * s (1 - s)
//   shs       = cat(_features_dc, _features_rest)   d_dc, d_rest = split(g)
// (means3D is the raw parameter itself.)  Pure HBM streaming: 224 B read + 224 B written per Gaussian.
#include "common.h"

namespace {

// thread t = (Gaussian i, quarter k4 of its 16x3 SH block when rest == 15): every lane stores one float4 of
// `shs` (fully coalesced) gathered from 4 consecutive source floats; the lanes with k4 == 0 also do the
// Gaussian's three small activations.  Generic `rest` falls back to one float per lane.
__device__ __forceinline__ float sh_src(const float *__restrict__ f_dc, const float *__restrict__ f_rest, size_t i,
                                        int c, int rest) {
    return c < 3 ? f_dc[3 * i + c] : f_rest[i * rest * 3 + (c - 3)];
}

__global__ void __launch_bounds__(256)
activate_forward_kernel(int N, int rest, const float *__restrict__ scaling, const float4 *__restrict__ rotation,
                        const float *__restrict__ opacity, const float *__restrict__ f_dc,
                        const float *__restrict__ f_rest, float *__restrict__ scales, float4 *__restrict__ rotations,
                        float *__restrict__ opacities, float *__restrict__ shs) {
    const int per = ((1 + rest) * 3 + 3) / 4;  // float4 groups per Gaussian (12 for degree 3)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * per) return;
    const size_t i = (size_t)(t / per);
    const int k4 = (int)(t % per), width = (1 + rest) * 3;
    float *o = shs + i * width;
    if (width % 4 == 0) {
        *reinterpret_cast<float4 *>(o + 4 * k4) =
            make_float4(sh_src(f_dc, f_rest, i, 4 * k4, rest), sh_src(f_dc, f_rest, i, 4 * k4 + 1, rest),
                        sh_src(f_dc, f_rest, i, 4 * k4 + 2, rest), sh_src(f_dc, f_rest, i, 4 * k4 + 3, rest));
    } else {
        for (int c = 4 * k4; c < min(4 * k4 + 4, width); c++) o[c] = sh_src(f_dc, f_rest, i, c, rest);
    }
    if (k4 == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) scales[3 * i + k] = expf(scaling[3 * i + k]);
        const float4 q = rotation[i];
        const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        rotations[i] = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
        opacities[i] = 1.0f / (1.0f + expf(-opacity[i]));
    }
}

__global__ void __launch_bounds__(256)
activate_backward_kernel(int N, int rest, const float4 *__restrict__ rotation, const float *__restrict__ scales,
                         const float *__restrict__ opacities, const float *__restrict__ g_scales,
                         const float4 *__restrict__ g_rotations, const float *__restrict__ g_opacities,
                         const float *__restrict__ g_shs, float *__restrict__ d_scaling,
                         float4 *__restrict__ d_rotation, float *__restrict__ d_opacity, float *__restrict__ d_dc,
                         float *__restrict__ d_rest) {
    const int per = ((1 + rest) * 3 + 3) / 4;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * per) return;
    const size_t i = (size_t)(t / per);
    const int k4 = (int)(t % per), width = (1 + rest) * 3;
    const float *gs = g_shs + i * width;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (width % 4 == 0) {
        const float4 g4 = *reinterpret_cast<const float4 *>(gs + 4 * k4);
        v[0] = g4.x; v[1] = g4.y; v[2] = g4.z; v[3] = g4.w;
    } else {
        for (int c = 0; c < 4 && 4 * k4 + c < width; c++) v[c] = gs[4 * k4 + c];
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int col = 4 * k4 + c;
        if (col < 3) d_dc[3 * i + col] = v[c];
        else if (col < width) d_rest[i * rest * 3 + (col - 3)] = v[c];
    }
    if (k4 == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) d_scaling[3 * i + k] = g_scales[3 * i + k] * scales[3 * i + k];
        const float4 x = rotation[i], g = g_rotations[i];
        const float nr = sqrtf(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
        const float n = fmaxf(nr, 1e-12f);
        const float4 y = make_float4(x.x / n, x.y / n, x.z / n, x.w / n);
        // below the clamp the norm is a constant: plain scaling
        const float dot = nr > 1e-12f ? (y.x * g.x + y.y * g.y + y.z * g.z + y.w * g.w) : 0.f;
        d_rotation[i] = make_float4((g.x - y.x * dot) / n, (g.y - y.y * dot) / n, (g.z - y.z * dot) / n,
                                    (g.w - y.w * dot) / n);
        const float s = opacities[i];
        d_opacity[i] = g_opacities[i] * s * (1.0f - s);
    }
}

}  // namespace

extern "C" int gsr_activate_forward(int N, int sh_rest, const float *scaling, const float *rotation,
                                    const float *opacity, const float *features_dc, const float *features_rest,
                                    float *scales, float *rotations, float *opacities, float *shs,
                                    gsr_stream_t stream) {
    if (N < 0 || sh_rest < 0) return GSR_EINVAL;
    if (N == 0) return 0;
    if (!scaling || !rotation || !opacity || !features_dc || (sh_rest > 0 && !features_rest) || !scales ||
        !rotations || !opacities || !shs)
        return GSR_EINVAL;
    const long long threads = (long long)N * (((1 + sh_rest) * 3 + 3) / 4);
    hipLaunchKernelGGL(activate_forward_kernel, dim3(gsr_div_up(threads, 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), N, sh_rest, scaling,
                       reinterpret_cast<const float4 *>(rotation), opacity, features_dc, features_rest, scales,
                       reinterpret_cast<float4 *>(rotations), opacities, shs);
    GSR_LAUNCH_CHECK();
    return 0;
}

extern "C" int gsr_activate_backward(int N, int sh_rest, const float *rotation, const float *scales,
                                     const float *opacities, const float *g_scales, const float *g_rotations,
                                     const float *g_opacities, const float *g_shs, float *d_scaling,
                                     float *d_rotation, float *d_opacity, float *d_features_dc,
                                     float *d_features_rest, gsr_stream_t stream) {
    if (N < 0 || sh_rest < 0) return GSR_EINVAL;
    if (N == 0) return 0;
    if (!rotation || !scales || !opacities || !g_scales || !g_rotations || !g_opacities || !g_shs || !d_scaling ||
        !d_rotation || !d_opacity || !d_features_dc || (sh_rest > 0 && !d_features_rest))
        return GSR_EINVAL;
    const long long threads = (long long)N * (((1 + sh_rest) * 3 + 3) / 4);
    hipLaunchKernelGGL(activate_backward_kernel, dim3(gsr_div_up(threads, 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), N, sh_rest, reinterpret_cast<const float4 *>(rotation),
                       scales, opacities, g_scales, reinterpret_cast<const float4 *>(g_rotations), g_opacities, g_shs,
                       d_scaling, reinterpret_cast<float4 *>(d_rotation), d_opacity, d_features_dc, d_features_rest);
    GSR_LAUNCH_CHECK();
    return 0;
}
