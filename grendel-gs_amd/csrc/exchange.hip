// exchange.hip -- device side of the sparse pixel-partition exchange.
//
// The reference runs K2 once per camera, then `nonzero()` once per (camera, destination band) -- W*B host
// syncs -- and gathers / concatenates the 9 + 2 per-Gaussian floats with ~10 torch ops per camera
// (gaussian_renderer/workload_division.py:721-744, gaussian_renderer/__init__.py:542-698).  With the
// camera-batched K1 the screen-space state of the whole batch is ONE set of [B, P, .] arrays, so:
//   gsr_exchange_need : ONE launch marks, for every (destination rank, camera, Gaussian), whether the rank
//                       renders a row band of that camera the Gaussian's tile rect touches, already in the
//                       (destination, camera, Gaussian) order the all-to-all-v needs, and counts per
//                       (destination, camera);
//   the 11-float records are then packed / unpacked by gsr_gather_rows (compact.hip), one launch each.
#include "common.h"

namespace {

__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
exchange_need_kernel(int P, int B, int W, int gx, int gy, const float2 *__restrict__ means2D,
                     const int32_t *__restrict__ radii, const int32_t *__restrict__ bands,
                     uint8_t *__restrict__ need, int32_t *__restrict__ counts) {
    __shared__ int32_t s_cnt[256];
    const int k = blockIdx.y;
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int miny = 0, maxy = 0;
    bool nonempty = false;
    if (i < P) {
        const size_t r = (size_t)k * P + i;
        const int rad = radii[r];
        if (rad > 0) {
            const float2 xy = means2D[r];
            int minx, maxx;
            gsr_get_rect(xy.x, xy.y, rad, gx, gy, minx, miny, maxx, maxy);
            nonempty = maxx > minx && maxy > miny;
        }
    }
    const int lane = threadIdx.x & 63;
    for (int g = 0; g < W; g++) {
        const int lo = bands[((size_t)k * W + g) * 2], hi = bands[((size_t)k * W + g) * 2 + 1];  // uniform loads
        const bool hit = nonempty && hi > lo && max(lo, miny) < min(hi, maxy);
        if (i < P) need[((size_t)g * B + k) * P + i] = hit ? 1 : 0;
        const unsigned long long m = __ballot(hit);
        if (lane == 0 && m) atomicAdd(&s_cnt[g], (int32_t)__popcll(m));
    }
    __syncthreads();
    if ((int)threadIdx.x < W && s_cnt[threadIdx.x]) atomicAdd(&counts[(size_t)threadIdx.x * B + k], s_cnt[threadIdx.x]);
}
}  // namespace

extern "C" int gsr_exchange_need(int P, int B, int W, int width, int height, const float *means2D,
                                 const int32_t *radii, const int32_t *bands, uint8_t *need, int32_t *counts,
                                 gsr_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || B < 1 || B > 65535 || W < 1 || W > 256 || width <= 0 || height <= 0 || !counts) return GSR_EINVAL;
    GSR_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)W * B, stream));
    if (P == 0) return 0;
    if (!means2D || !radii || !bands || !need) return GSR_EINVAL;
    const int gx = (width + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    hipLaunchKernelGGL(exchange_need_kernel, dim3(gsr_div_up(P, GSR_ONE_DIM_BLOCK), B), dim3(GSR_ONE_DIM_BLOCK), 0,
                       stream, P, B, W, gx, gy, reinterpret_cast<const float2 *>(means2D), radii, bands, need, counts);
    GSR_LAUNCH_CHECK();
    return 0;
}
