// knn.hip -- mean squared distance to the 3 nearest neighbours of every point: the operator behind
// `simple_knn._C.distCUDA2(points)` that the reference uses once, at scene creation, to size the initial
// Gaussians (scene/gaussian_model.py:20,163-166 of the reference; the submodule itself is absent from the
// reference tree -- https://gitlab.inria.fr/bkerbl/simple-knn, `.gitmodules:1-3`).
//
// Published algorithm of that submodule, restated for gfx950: Morton-order the points, cut the order into boxes
// of 1024 points with an axis-aligned bound, and for every point visit only the boxes whose bound is not
// farther than its current third-best distance.  The result is the EXACT 3-NN mean (only the point itself is
// excluded, by index; coincident points count with distance 0), so the oracle is a brute-force scan.
//
// MI355X mapping: the Morton sort is the one-sweep radix sort of radix.h (4 passes over 30 bits, histograms
// fused into the key-producing kernel); the search runs one workgroup per 256 Morton-consecutive points, which
// share almost the same set of candidate boxes: a candidate box is staged ONCE per workgroup in LDS (16 KB) and
// walked with broadcast reads; the three best distances live in registers and are updated branch-free.
#include "common.h"

#include "radix.h"

#pragma clang fp contract(off)  // d = dx*dx + dy*dy + dz*dz evaluated exactly like the oracle (no FMA)

namespace {

constexpr int KNN_BOX = 1024;
constexpr int KNN_THREADS = 256;

__device__ __forceinline__ uint32_t float_order(float f) {  // monotone float -> uint
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float order_float(uint32_t o) {
    const uint32_t u = (o >> 31) ? (o ^ 0x80000000u) : ~o;
    return __uint_as_float(u);
}

// bb[0..2] = max of order(x,y,z); bb[3..5] = max of ~order (i.e. the minimum), both start at 0 (memset)
__global__ void __launch_bounds__(KNN_THREADS) bbox_kernel(int P, const float *__restrict__ pts, uint32_t *__restrict__ bb) {
    __shared__ uint32_t s[6];
    if (threadIdx.x < 6) s[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mx[3] = {0, 0, 0}, mn[3] = {0, 0, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = pts[3 * i + c];
            if (!(fabsf(v) <= 3.0e38f)) continue;  // NaN / inf do not shape the bound
            const uint32_t o = float_order(v);
            mx[c] = max(mx[c], o);
            mn[c] = max(mn[c], ~o);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        atomicMax(&s[c], mx[c]);
        atomicMax(&s[3 + c], mn[c]);
    }
    __syncthreads();
    if (threadIdx.x < 6 && s[threadIdx.x]) atomicMax(&bb[threadIdx.x], s[threadIdx.x]);
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {  // 10 bits -> every third bit
    x &= 0x3FFu;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(KNN_THREADS)
morton_kernel(int P, const float *__restrict__ pts, const uint32_t *__restrict__ bb, RadixPlan plan,
              uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t *__restrict__ ghist) {
    __shared__ uint32_t mh[RADIX_MAX_PASSES][RADIX_DIGITS];
    for (int p = 0; p < RADIX_MAX_PASSES; p++) mh[p][threadIdx.x] = 0;
    __syncthreads();
    float lo[3], inv[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float hi = bb[c] ? order_float(bb[c]) : 0.f;
        lo[c] = bb[3 + c] ? order_float(~bb[3 + c]) : 0.f;
        inv[c] = hi > lo[c] ? 1023.0f / (hi - lo[c]) : 0.f;
    }
    for (long long base = (long long)blockIdx.x * blockDim.x; base < P; base += (long long)gridDim.x * blockDim.x) {
        const long long i = base + threadIdx.x;
        const bool valid = i < P;
        uint32_t key = 0;
        if (valid) {
            uint32_t q[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float t = (pts[3 * i + c] - lo[c]) * inv[c];
                q[c] = (uint32_t)fminf(fmaxf(t, 0.f), 1023.f);  // NaN -> 0
            }
            key = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
            keys[i] = key;
            vals[i] = (uint32_t)i;
        }
        multihist_add(mh, plan, key, valid);
    }
    __syncthreads();
    multihist_flush(mh, plan, ghist);
}

// one workgroup per box of KNN_BOX Morton-consecutive points: sorted coordinates (float4, w unused) + bound
__global__ void __launch_bounds__(KNN_THREADS)
box_kernel(int P, const float *__restrict__ pts, const uint32_t *__restrict__ sorted_ids, float4 *__restrict__ spts,
           float *__restrict__ boxes) {
    __shared__ float red[6][KNN_THREADS / 64];
    const long long b0 = (long long)blockIdx.x * KNN_BOX;
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int k = threadIdx.x; k < KNN_BOX; k += KNN_THREADS) {
        const long long j = b0 + k;
        if (j >= P) break;
        const uint32_t g = sorted_ids[j];
        const float x = pts[3 * (size_t)g], y = pts[3 * (size_t)g + 1], z = pts[3 * (size_t)g + 2];
        spts[j] = make_float4(x, y, z, 0.f);
        mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
        mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
        mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = mn[c], b = mx[c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            a = fminf(a, __shfl_xor(a, d, 64));
            b = fmaxf(b, __shfl_xor(b, d, 64));
        }
        if (lane == 0) { red[c][wave] = a; red[3 + c][wave] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int w = 1; w < KNN_THREADS / 64; w++)
            v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
        boxes[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

__device__ __forceinline__ void best3_update(float &b0, float &b1, float &b2, float d) {
    const float t0 = fmaxf(b0, d);
    b0 = fminf(b0, d);
    const float t1 = fmaxf(b1, t0);
    b1 = fminf(b1, t0);
    b2 = fminf(b2, t1);
}

// workgroup = 256 Morton-consecutive points (all inside one box); candidate boxes are staged in LDS
__global__ void __launch_bounds__(KNN_THREADS)
search_kernel(int P, const float4 *__restrict__ spts, const uint32_t *__restrict__ sorted_ids,
              const float *__restrict__ boxes, int nboxes, float *__restrict__ out) {
    __shared__ float4 sbox[KNN_BOX];
    const long long j = (long long)blockIdx.x * KNN_THREADS + threadIdx.x;
    const bool valid = j < P;
    const float4 p = valid ? spts[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    float b0 = 3.4028235e38f, b1 = 3.4028235e38f, b2 = 3.4028235e38f;
    const int own = (int)(((long long)blockIdx.x * KNN_THREADS) / KNN_BOX);
    for (int it = 0; it < nboxes; it++) {
        // own box first: it tightens the third-best distance before the other bounds are tested
        const int b = it == 0 ? own : (it <= own ? it - 1 : it);
        bool need = false;
        if (valid) {
            const float *bx = boxes + (size_t)b * 6;
            const float dx = fmaxf(fmaxf(bx[0] - p.x, p.x - bx[3]), 0.f);
            const float dy = fmaxf(fmaxf(bx[1] - p.y, p.y - bx[4]), 0.f);
            const float dz = fmaxf(fmaxf(bx[2] - p.z, p.z - bx[5]), 0.f);
            need = !(dx * dx + dy * dy + dz * dz > b2);  // NaN -> visit
        }
        if (!__syncthreads_or(need)) continue;
        const long long s0 = (long long)b * KNN_BOX;
        const int cnt = (int)min((long long)KNN_BOX, (long long)P - s0);
        for (int k = threadIdx.x; k < cnt; k += KNN_THREADS) sbox[k] = spts[s0 + k];
        __syncthreads();
        if (need) {
            const int self = (int)(j - s0);  // position of this point inside the box, if it is there
            for (int k = 0; k < cnt; k++) {
                const float4 q = sbox[k];
                const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
                float d = dx * dx + dy * dy + dz * dz;
                if (k == self || !(d == d)) d = 3.4028235e38f;  // the point itself; NaN candidates
                best3_update(b0, b1, b2, d);
            }
        }
        __syncthreads();
    }
    if (valid) {
        // fewer than three other points (P < 4): average what exists, 0 for a single point
        float s = 0.f;
        int n = 0;
        if (b0 < 3.4028235e38f) { s += b0; n++; }
        if (b1 < 3.4028235e38f) { s += b1; n++; }
        if (b2 < 3.4028235e38f) { s += b2; n++; }
        out[sorted_ids[j]] = n == 3 ? s / 3.0f : (n ? s / (float)n : 0.f);
    }
}

struct KnnLayout {
    size_t bb, ctrl, kA, vA, kB, vB, spts, boxes, total;
    CtrlLayout C;
};
KnnLayout knn_layout(int P) {
    KnnLayout L;
    size_t o = 0;
    L.bb = o; o += 256;
    L.ctrl = o;
    L.C = ctrl_layout(P, 4, false);
    o += L.C.total;
    const size_t np = align_up((size_t)(P + 1) * 4);
    L.kA = o; o += np;
    L.vA = o; o += np;
    L.kB = o; o += np;
    L.vB = o; o += np;
    L.spts = o; o += align_up((size_t)(P + 1) * 16);
    L.boxes = o; o += align_up((size_t)((P + KNN_BOX - 1) / KNN_BOX + 1) * 6 * 4);
    L.total = o;
    return L;
}
}  // namespace

extern "C" size_t gsr_knn_workspace_bytes(int P) {
    if (P < 0) return 0;
    return knn_layout(P).total;
}

extern "C" int gsr_knn_mean_dist2(int P, const float *points, float *mean_dist2, void *workspace,
                                  size_t workspace_bytes, gsr_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0) return GSR_EINVAL;
    if (P == 0) return 0;
    if (!points || !mean_dist2 || !workspace) return GSR_EINVAL;
    if (P > RADIX_MAX_N) return GSR_EINVAL;
    const KnnLayout L = knn_layout(P);
    if (workspace_bytes < L.total) return GSR_ENOSPACE;
    char *base = reinterpret_cast<char *>(workspace);
    uint32_t *bb = reinterpret_cast<uint32_t *>(base + L.bb);
    char *ctrl = base + L.ctrl;
    uint32_t *kA = reinterpret_cast<uint32_t *>(base + L.kA), *vA = reinterpret_cast<uint32_t *>(base + L.vA);
    uint32_t *kB = reinterpret_cast<uint32_t *>(base + L.kB), *vB = reinterpret_cast<uint32_t *>(base + L.vB);
    float4 *spts = reinterpret_cast<float4 *>(base + L.spts);
    float *boxes = reinterpret_cast<float *>(base + L.boxes);

    GSR_HIP(hipMemsetAsync(base, 0, L.ctrl + L.C.total, stream));  // bound + sort control block
    const int grid = gsr_div_up(P, KNN_THREADS) < 512 ? gsr_div_up(P, KNN_THREADS) : 512;
    hipLaunchKernelGGL(bbox_kernel, dim3(grid), dim3(KNN_THREADS), 0, stream, P, points, bb);
    const RadixPlan plan = radix_plan(0, 30);
    hipLaunchKernelGGL(morton_kernel, dim3(grid), dim3(KNN_THREADS), 0, stream, P, points, bb, plan, kA, vA,
                       reinterpret_cast<uint32_t *>(ctrl + L.C.ghist));
    int in_first = 1;
    int rc = radix_sort_pairs(kA, vA, kB, vB, P, plan, ctrl, L.C, &in_first, stream, nullptr);
    if (rc) return rc;
    const uint32_t *sorted_ids = in_first ? vA : vB;
    const int nboxes = gsr_div_up(P, KNN_BOX);
    hipLaunchKernelGGL(box_kernel, dim3(nboxes), dim3(KNN_THREADS), 0, stream, P, points, sorted_ids, spts, boxes);
    hipLaunchKernelGGL(search_kernel, dim3(gsr_div_up(P, KNN_THREADS)), dim3(KNN_THREADS), 0, stream, P, spts,
                       sorted_ids, boxes, nboxes, mean_dist2);
    GSR_LAUNCH_CHECK();
    return 0;
}
