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
// The fused path used by the mirror does not materialise the need mask at all:
//   gsr_exchange_count : the same K2 test, counted per (destination, camera) and per 1024-Gaussian chunk;
//   gsr_exchange_pack  : (after the ONE host read-back that sizes the message) every wave recomputes the test for its
//                        chunk, ranks the hits with ballots -- the order is (destination, camera, local index), the
//                        order the reference produces -- and writes the 11-float records (means2D 2, rgb 3,
//                        conic_opacity 4, radius bits, depth) straight into the all-to-all send buffer, plus the row
//                        index list the backward needs;
//   gsr_scatter_add_rows: the mirror step of the backward: gradient rows coming back from the peers are added into
//                        the owners' rows (a Gaussian needed by two bands receives two contributions), 9 adjacent
//                        lanes per row like K10's flush.
// These replace nonzero_static + two index kernels + a gather launch in the forward and three zero fills + three
// index_add_ launches in the backward.
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
#ifndef GSR_XCHUNK
#define GSR_XCHUNK 1024
#endif
constexpr int XCHUNK = GSR_XCHUNK;  // Gaussians per wave-chunk: XCHUNK / 64 rounds of 64 lanes

// rows [miny, maxy) of the 3-sigma tile rect of Gaussian r (the K2 rule) packed as miny | maxy << 16, or 0 when it
// touches nothing (0 = the empty interval: no band test passes).  Both loads are unconditional, so that the loads of
// several rounds can be in flight together (the radius no longer gates the load of the position).
__device__ __forceinline__ uint32_t rect_rows_packed(int rad, float2 xy, int gx, int gy) {
    int minx, miny, maxx, maxy;
    gsr_get_rect(xy.x, xy.y, rad > 0 ? rad : 0, gx, gy, minx, miny, maxx, maxy);
    const bool ok = rad > 0 && maxx > minx && maxy > miny;
    return ok ? ((uint32_t)miny | ((uint32_t)maxy << 16)) : 0u;
}
__device__ __forceinline__ bool rows_hit(uint32_t rows, int lo, int hi) {
    const int miny = (int)(rows & 0xffffu), maxy = (int)(rows >> 16);
    return hi > lo && max(lo, miny) < min(hi, maxy);
}

// grid (chunks, cameras), 256 threads = 4 waves, one chunk per wave.  One wave is a latency chain (a chunk is 16
// rounds of 64 Gaussians and there is less than one wave per SIMD at 10^6 Gaussians), so ALL loads of the chunk are
// issued first and the band tests run on registers; the counts stay in scalar registers (round 3: the per-round
// dependent loads and an LDS counter per destination made this 33 us for 0.75 M Gaussians and 8 destinations).
__global__ void __launch_bounds__(256)
exchange_count_kernel(int P, int B, int W, int k0, int gx, int gy, int nchunk, const float2 *__restrict__ means2D,
                      const int32_t *__restrict__ radii, const int32_t *__restrict__ bands,
                      int32_t *__restrict__ chunkcnt, int32_t *__restrict__ counts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = blockIdx.y, k = k0 + kk;  // kk: camera within the launch, k: camera of the batch
    const int chunk = blockIdx.x * 4 + wave;
    if (chunk >= nchunk) return;
    constexpr int R = XCHUNK / 64;
    int32_t rad[R];
    float2 xy[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = chunk * XCHUNK + r * 64 + lane;
        const size_t row = (size_t)k * P + min(i, P - 1);  // clamped, not branched around: the loads stay together
        rad[r] = radii[row];
        xy[r] = means2D[row];
        if (i >= P) rad[r] = 0;
    }
    uint32_t rows[R];
#pragma unroll
    for (int r = 0; r < R; r++) rows[r] = rect_rows_packed(rad[r], xy[r], gx, gy);
    for (int g = 0; g < W; g++) {
        const int lo = bands[((size_t)k * W + g) * 2], hi = bands[((size_t)k * W + g) * 2 + 1];  // uniform
        int32_t c = 0;
#pragma unroll
        for (int r = 0; r < R; r++) c += (int32_t)__popcll(__ballot(rows_hit(rows[r], lo, hi)));
        if (lane == 0) chunkcnt[((size_t)g * B + kk) * nchunk + chunk] = c;
    }
}

// counts[segment] = sum of the segment's per-chunk counts; one workgroup per (destination, camera) segment.  (The count
// kernel used to add every chunk's count to its segment's total with a global atomic: W x B words of ONE cache line
// hammered by every wave of the launch -- 5 900 same-line atomics for 0.75 M Gaussians and 8 destinations, which an L2
// channel retires at ~100 per microsecond.  Once the loads no longer hid it that WAS the kernel: 71 us.)
__global__ void __launch_bounds__(256)
exchange_sum_kernel(int nchunk, const int32_t *__restrict__ chunkcnt, int32_t *__restrict__ counts) {
    __shared__ int32_t red[4];
    const int32_t *row = chunkcnt + (size_t)blockIdx.x * nchunk;
    int32_t acc = 0;
    for (int c0 = threadIdx.x; c0 < nchunk; c0 += 256 * 4) {
        int32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = row[min(c0 + 256 * u, nchunk - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (c0 + 256 * u < nchunk) acc += v[u];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

struct SegOffsets {
    int32_t off[513];  // first message row of segment (destination g, camera kk) = off[g * B + kk]; SLAB: off[s + 1] is
                       // also the END of segment s (its capacity is off[s + 1] - off[s])
};

// SLAB = false: the segments are exactly as long as the counts say (the caller has read them back).
// SLAB = true : the segments are capacity slabs chosen BEFORE the counts were known (no read-back): a record whose
//               position falls past its segment's end is not written (overflow -- the caller learns it from the counts
//               later and repeats the exchange with the sized layout).
// Same latency discipline as the count kernel: the records of FOUR rounds are loaded together (unconditionally: a
// Gaussian that goes nowhere costs 44 bytes of reads), the running position of destination g lives in lane g & 63 of a
// register (read with a wave-uniform lane index), not in LDS.
template <bool SLAB>
__global__ void __launch_bounds__(256)
exchange_pack_kernel(int P, int B, int W, int k0, int cntB, int cnt0, int gx, int gy, int nchunk,
                     const float2 *__restrict__ means2D,
                     const float *__restrict__ rgb, const float4 *__restrict__ conic_opacity,
                     const int32_t *__restrict__ radii, const float *__restrict__ depths,
                     const int32_t *__restrict__ bands, const int32_t *__restrict__ chunkcnt, SegOffsets seg,
                     float *__restrict__ msg, int32_t *__restrict__ send_idx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = blockIdx.y, k = k0 + kk;
    const int chunk = blockIdx.x * 4 + wave;
    if (chunk >= nchunk) return;  // wave-uniform; no workgroup barrier below
    // first row of this chunk in each segment: segment start + the counts of the chunks before it.  Eight
    // destinations at a time, so that their loads are in flight together; base[g >> 6] of lane g & 63 = destination g's.
    int32_t base[4] = {0, 0, 0, 0};
    for (int g0 = 0; g0 < W; g0 += 8) {
        int32_t part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int32_t *cc[8];
#pragma unroll
        for (int j = 0; j < 8; j++)  // (rows past the last destination re-read the last one: no branch around a load)
            cc[j] = chunkcnt + ((size_t)min(g0 + j, W - 1) * cntB + (cnt0 + kk)) * nchunk;
#pragma unroll 2
        for (int c = lane; c < chunk; c += 64) {
#pragma unroll
            for (int j = 0; j < 8; j++) part[j] += cc[j][c];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int g = g0 + j;
            if (g < W) {  // uniform
                int32_t v = part[j];
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
                v += seg.off[g * B + kk];
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if ((g >> 6) == q && lane == (g & 63)) base[q] = v;
            }
        }
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    constexpr int RB = 4;  // rounds whose records are in registers together
    for (int rb = 0; rb < XCHUNK / 64; rb += RB) {
        float rec[RB][11];
        uint32_t rows[RB];
#pragma unroll
        for (int q = 0; q < RB; q++) {
            const int i = chunk * XCHUNK + (rb + q) * 64 + lane;
            const size_t row = (size_t)k * P + min(i, P - 1);  // clamped, not branched around (see the count kernel)
            const int32_t rad = radii[row];
            const float2 xy = means2D[row];
            const float4 co = conic_opacity[row];
            rec[q][0] = xy.x; rec[q][1] = xy.y;
            rec[q][2] = rgb[3 * row]; rec[q][3] = rgb[3 * row + 1]; rec[q][4] = rgb[3 * row + 2];
            rec[q][5] = co.x; rec[q][6] = co.y; rec[q][7] = co.z; rec[q][8] = co.w;
            rec[q][9] = __int_as_float(rad);
            rec[q][10] = depths[row];
        }
        __builtin_amdgcn_sched_barrier(0);  // all twenty loads are issued before the first result is looked at
#pragma unroll
        for (int q = 0; q < RB; q++) {
            const int i = chunk * XCHUNK + (rb + q) * 64 + lane;
            rows[q] = rect_rows_packed(i < P ? __float_as_int(rec[q][9]) : 0, make_float2(rec[q][0], rec[q][1]), gx, gy);
        }
#pragma unroll
        for (int q = 0; q < RB; q++) {
            if (__ballot(rows[q] != 0u) == 0ull) continue;
            const int i = chunk * XCHUNK + (rb + q) * 64 + lane;
#pragma unroll
            for (int gb = 0; gb < 4; gb++) {
                for (int g = gb * 64; g < min(W, gb * 64 + 64); g++) {
                    const int lo = bands[((size_t)k * W + g) * 2], hi = bands[((size_t)k * W + g) * 2 + 1];  // uniform
                    const bool hit = rows_hit(rows[q], lo, hi);
                    const unsigned long long m = __ballot(hit);
                    if (m == 0ull) continue;
                    const int32_t b0 = __builtin_amdgcn_readlane(base[gb], g & 63);
                    const size_t pos = (size_t)b0 + __popcll(m & lt);
                    if (hit && (!SLAB || (int64_t)pos < (int64_t)seg.off[g * B + kk + 1])) {
                        float *dst = msg + pos * 11;
#pragma unroll
                        for (int c = 0; c < 11; c++) dst[c] = rec[q][c];
                        send_idx[pos] = (int32_t)(kk * P + i);  // row of the [cameras of this launch, P] state
                    }
                    if (lane == (g & 63)) base[gb] += (int32_t)__popcll(m);
                }
            }
        }
    }
}

// SLAB: the unused tail of every segment -- rows [min(count, capacity), capacity) -- becomes an all-zero record
// (radius 0 = "culled": the receiver's K3 drops it) with send_idx -1 (the backward's scatter-add skips it).
// grid (segments, TAIL_SPLIT)
constexpr int TAIL_SPLIT = 64;  // (8 workgroups per slab took 20 us for ~1 MB of padding: too few to fill the chip)
__global__ void __launch_bounds__(256)
exchange_slab_tail_kernel(int B, int cntB, int cnt0, const int32_t *__restrict__ counts, SegOffsets seg,
                          float *__restrict__ msg, int32_t *__restrict__ send_idx) {
    const int s = blockIdx.x, g = s / B, kk = s - g * B;
    const int32_t lo = seg.off[s], hi = seg.off[s + 1];
    const int32_t n = min(counts[(size_t)g * cntB + cnt0 + kk], hi - lo);
    const long long first = (long long)(lo + n) * 11, last = (long long)hi * 11;
    for (long long e = first + (long long)blockIdx.y * 256 + threadIdx.x; e < last; e += 256LL * TAIL_SPLIT) msg[e] = 0.f;
    for (int32_t r = lo + n + (int32_t)blockIdx.y * 256 + (int32_t)threadIdx.x; r < hi; r += 256 * TAIL_SPLIT)
        send_idx[r] = -1;
}

// The received slab [n][11] -> the five dense tensors the render op takes.  A workgroup moves 256 rows through LDS:
// the slab is read as one contiguous 16-byte-per-lane stream and every output array is written as one contiguous run
// per workgroup (float2 / float4 stores), instead of the 24-92 byte runs per wave that one lane per float produced
// (and of the 44-byte-stride column reads of the generic row mover, gsr_gather_rows: 43 us for 0.65 M rows).
// Row stride 11 words in LDS: odd, the per-row reads below are conflict-free.
__global__ void __launch_bounds__(256)
exchange_unpack_kernel(long long n, int vec, const float *__restrict__ recv, float *__restrict__ means2D,
                       float *__restrict__ rgb, float *__restrict__ conic_opacity, int32_t *__restrict__ radii,
                       float *__restrict__ depths) {
    __shared__ __attribute__((aligned(16))) float s[256 * 11];
    const int t = threadIdx.x;
    for (long long row0 = (long long)blockIdx.x * 256; row0 < n; row0 += (long long)gridDim.x * 256) {
        const int rows = (int)min(256LL, n - row0);
        const float *src = recv + row0 * 11;  // 11264-byte blocks: 16-byte aligned with the buffer
        if (rows == 256 && vec) {  // (vec: the slab starts on a 16-byte boundary)
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            float4 *d4 = reinterpret_cast<float4 *>(s);
            for (int e = t; e < 256 * 11 / 4; e += 256) d4[e] = s4[e];
        } else {
            for (int e = t; e < rows * 11; e += 256) s[e] = src[e];
        }
        __syncthreads();
        if (t < rows) {
            const float *r = s + t * 11;
            reinterpret_cast<float2 *>(means2D)[row0 + t] = make_float2(r[0], r[1]);
            reinterpret_cast<float4 *>(conic_opacity)[row0 + t] = make_float4(r[5], r[6], r[7], r[8]);
            radii[row0 + t] = __float_as_int(r[9]);
            depths[row0 + t] = r[10];
        }
        for (int e = t; e < rows * 3; e += 256) rgb[row0 * 3 + e] = s[(e / 3) * 11 + 2 + (e % 3)];
        __syncthreads();
    }
}

// dst[idx[r]][0:9] += src[r][0:9]; 9 adjacent lanes per row; rows with idx < 0 (slab padding) are skipped
__global__ void __launch_bounds__(256)
scatter_add_rows_kernel(long long n, const int32_t *__restrict__ idx, const float *__restrict__ src,
                        float *__restrict__ dst) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n * 9;
         e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / 9;
        const int c = (int)(e - r * 9);
        const float v = src[e];
        const int32_t row = idx[r];
        if (v != 0.f && row >= 0) atomicAdd(dst + (size_t)row * 9 + c, v);
    }
}
}  // namespace

extern "C" size_t gsr_exchange_chunks(int P) { return P <= 0 ? 0 : (size_t)gsr_div_up(P, XCHUNK); }

extern "C" int gsr_exchange_count(int P, int B_total, int k0, int B, int W, int width, int height,
                                  const float *means2D, const int32_t *radii, const int32_t *bands,
                                  int32_t *chunkcnt, int32_t *counts, gsr_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || B < 1 || B > 65535 || k0 < 0 || k0 + B > B_total || W < 1 || W > 256 || W * B > 512 ||
        width <= 0 || height <= 0 || !counts)
        return GSR_EINVAL;
    if (P == 0) {
        GSR_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)W * B, stream));
        return 0;
    }
    if (!means2D || !radii || !bands || !chunkcnt) return GSR_EINVAL;
    const int gx = (width + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    const int nchunk = gsr_div_up(P, XCHUNK);
    hipLaunchKernelGGL(exchange_count_kernel, dim3(gsr_div_up(nchunk, 4), B), dim3(256), 0, stream, P, B, W, k0, gx, gy,
                       nchunk, reinterpret_cast<const float2 *>(means2D), radii, bands, chunkcnt, counts);
    hipLaunchKernelGGL(exchange_sum_kernel, dim3(W * B), dim3(256), 0, stream, nchunk, chunkcnt, counts);
    GSR_LAUNCH_CHECK();
    return 0;
}

static int exchange_pack_impl(bool slab, int P, int B_total, int k0, int B, int W, int width, int height,
                              int count_cameras, int count_first, const float *means2D, const float *rgb,
                              const float *conic_opacity, const int32_t *radii, const float *depths,
                              const int32_t *bands, const int32_t *chunkcnt, const int32_t *counts,
                              const int32_t *layout, int64_t n_send, float *msg, int32_t *send_idx,
                              hipStream_t stream) {
    if (P < 0 || B < 1 || B > 65535 || k0 < 0 || k0 + B > B_total || W < 1 || W > 256 || W * B > 512 ||
        width <= 0 || height <= 0 || n_send < 0 || !layout || count_cameras < B || count_first < 0 ||
        k0 < count_first || k0 - count_first + B > count_cameras)
        return GSR_EINVAL;
    SegOffsets seg;
    if (slab) {  // layout = capacities: prefix sums give the segment starts, the total must be the buffer's row count
        int64_t o = 0;
        for (int i = 0; i < W * B; i++) {
            if (layout[i] < 0) return GSR_EINVAL;
            seg.off[i] = (int32_t)o;
            o += layout[i];
        }
        if (o != n_send || o > 0x7fffffffLL) return GSR_EINVAL;
        seg.off[W * B] = (int32_t)o;
    } else {
        for (int i = 0; i < W * B; i++) seg.off[i] = layout[i];
        seg.off[W * B] = (int32_t)(n_send > 0x7fffffffLL ? 0x7fffffff : n_send);
    }
    if (n_send == 0) return 0;
    if (!msg || !send_idx) return GSR_EINVAL;
    const int gx = (width + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    const int nchunk = gsr_div_up(P, XCHUNK);
    if (P > 0) {
        if (!means2D || !rgb || !conic_opacity || !radii || !depths || !bands || !chunkcnt) return GSR_EINVAL;
        if (slab)
            hipLaunchKernelGGL(exchange_pack_kernel<true>, dim3(gsr_div_up(nchunk, 4), B), dim3(256), 0, stream, P, B, W,
                               k0, count_cameras, k0 - count_first, gx, gy, nchunk,
                               reinterpret_cast<const float2 *>(means2D), rgb,
                               reinterpret_cast<const float4 *>(conic_opacity), radii, depths, bands, chunkcnt, seg, msg,
                               send_idx);
        else
            hipLaunchKernelGGL(exchange_pack_kernel<false>, dim3(gsr_div_up(nchunk, 4), B), dim3(256), 0, stream, P, B, W,
                               k0, count_cameras, k0 - count_first, gx, gy, nchunk,
                               reinterpret_cast<const float2 *>(means2D), rgb,
                               reinterpret_cast<const float4 *>(conic_opacity), radii, depths, bands, chunkcnt, seg, msg,
                               send_idx);
        GSR_LAUNCH_CHECK();
    }
    if (slab) {
        if (!counts) return GSR_EINVAL;
        hipLaunchKernelGGL(exchange_slab_tail_kernel, dim3(W * B, TAIL_SPLIT), dim3(256), 0, stream, B, count_cameras,
                           k0 - count_first, counts, seg, msg, send_idx);
        GSR_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int gsr_exchange_pack(int P, int B_total, int k0, int B, int W, int width, int height, int count_cameras,
                                 int count_first, const float *means2D, const float *rgb, const float *conic_opacity,
                                 const int32_t *radii, const float *depths, const int32_t *bands,
                                 const int32_t *chunkcnt, const int32_t *segment_offsets, int64_t n_send, float *msg,
                                 int32_t *send_idx, gsr_stream_t stream_) {
    if (P == 0 || n_send == 0) {
        if (P < 0 || n_send < 0 || !segment_offsets) return GSR_EINVAL;
        return 0;
    }
    return exchange_pack_impl(false, P, B_total, k0, B, W, width, height, count_cameras, count_first, means2D, rgb,
                              conic_opacity, radii, depths, bands, chunkcnt, nullptr, segment_offsets, n_send, msg,
                              send_idx, reinterpret_cast<hipStream_t>(stream_));
}

extern "C" int gsr_exchange_pack_slab(int P, int B_total, int k0, int B, int W, int width, int height,
                                      int count_cameras, int count_first, const float *means2D, const float *rgb,
                                      const float *conic_opacity, const int32_t *radii, const float *depths,
                                      const int32_t *bands, const int32_t *chunkcnt, const int32_t *counts,
                                      const int32_t *capacities, int64_t n_rows, float *msg, int32_t *send_idx,
                                      gsr_stream_t stream_) {
    return exchange_pack_impl(true, P, B_total, k0, B, W, width, height, count_cameras, count_first, means2D, rgb,
                              conic_opacity, radii, depths, bands, chunkcnt, counts, capacities, n_rows, msg, send_idx,
                              reinterpret_cast<hipStream_t>(stream_));
}

extern "C" int gsr_exchange_unpack(int64_t n, const float *recv, float *means2D, float *rgb, float *conic_opacity,
                                   int32_t *radii, float *depths, gsr_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (n < 0) return GSR_EINVAL;
    if (n == 0) return 0;
    if (!recv || !means2D || !rgb || !conic_opacity || !radii || !depths) return GSR_EINVAL;
    if (((uintptr_t)conic_opacity & 15) || ((uintptr_t)means2D & 7)) return GSR_EINVAL;
    long long blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(exchange_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (long long)n,
                       (int)(((uintptr_t)recv & 15) == 0), recv, means2D, rgb, conic_opacity, radii, depths);
    GSR_LAUNCH_CHECK();
    return 0;
}

// ---- capacity checks of a captured (hipGraph) iteration: nothing reaches the host inside a replay, so the kernels
// that depend on a capacity chosen at capture time raise bits of ONE device word when it does not hold; the fused
// K11 + Adam launch of the same replay then changes nothing (preprocess.hip: K11Adam::skip) and the host, which looks
// at the word after the replay, repeats the iteration eagerly.
namespace {
__global__ void flag_if_greater_kernel(const uint32_t *__restrict__ value, uint32_t limit, uint32_t *__restrict__ flag,
                                       uint32_t bit, uint32_t *__restrict__ host_copy) {
    if (threadIdx.x == 0) {
        const uint32_t v = *value;
        if (v > limit) atomicOr(flag, bit);
        if (host_copy) __hip_atomic_store(host_copy, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// all_counts [W][W][B] (rows rank i sends rank j of camera k, all-gathered), caps the same shape: bit_over when a
// count exceeds its slab; bit_few when a band this rank renders (bit k of rendered_mask) receives fewer than `few`
// rows in total (the reference's stand-in rule needs the true count, gaussian_renderer/__init__.py:1260-1269)
__global__ void exchange_check_kernel(const int32_t *__restrict__ all_counts, const int32_t *__restrict__ caps, int W,
                                      int B, int me, unsigned long long rendered_mask, int few,
                                      uint32_t *__restrict__ flag, uint32_t bit_over, uint32_t bit_few,
                                      int32_t *__restrict__ host_copy) {
    const int n = W * W * B;
    bool over = false;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int32_t c = all_counts[i];
        over |= c > caps[i];
        if (host_copy) __hip_atomic_store(host_copy + i, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    bool lack = false;
    for (int k = threadIdx.x; k < B; k += blockDim.x) {
        if (!((rendered_mask >> k) & 1ull)) continue;
        long long rows = 0;
        for (int i = 0; i < W; i++) rows += all_counts[((size_t)i * W + me) * B + k];
        lack |= rows < few;
    }
    if (__any(over) && (threadIdx.x & 63) == 0) atomicOr(flag, bit_over);
    if (__any(lack) && (threadIdx.x & 63) == 0) atomicOr(flag, bit_few);
}
// end of a captured iteration: the flag word and the replay's sequence number (a device word the host refreshes in
// front of every replay) go to slot seq % slots of a pinned, device-mapped ring; the host polls the stamp (as it polls
// the pair count, binning.hip) and so learns EXACTLY which replay was the first one that did not count
__global__ void publish_flag_kernel(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ seq,
                                    uint32_t *__restrict__ host_ring, uint32_t slots) {
    if (threadIdx.x == 0) {
        const uint32_t s = *seq;
        uint32_t *w = host_ring + 2 * (size_t)(s % slots);
        __hip_atomic_store(w, *flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(w + 1, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// a device timestamp (s_memrealtime: 100 MHz, the same clock on every CU) into word `index` of slot seq % slots of a pinned,
// device-mapped ring of `per_slot` words per slot: what a HIP event pair measures in the eager loop, for an iteration that
// runs as a hipGraph replay (events recorded inside a capture cannot be read)
__global__ void stamp_kernel(const uint32_t *__restrict__ seq, unsigned long long *__restrict__ host_ring, uint32_t slots,
                             uint32_t per_slot, uint32_t index) {
    if (threadIdx.x == 0) {
        const uint32_t s = seq ? *seq : 0u;
        __hip_atomic_store(host_ring + (size_t)(s % slots) * per_slot + index,
                           (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// compute_locally of B row bands whose rows are DEVICE data: mask[k][ty][tx] = lo_k <= ty < hi_k, band_rows = B records
// of `stride` int32 words that begin with { lo, hi }
__global__ void __launch_bounds__(256) band_mask_kernel(int gx, int gy, int B, const int32_t *__restrict__ band_rows,
                                                         int stride, uint8_t *__restrict__ mask) {
    const int tiles = gx * gy;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < tiles * B; i += gridDim.x * 256) {
        const int k = i / tiles, ty = (i - k * tiles) / gx;
        const int lo = band_rows[(size_t)k * stride], hi = band_rows[(size_t)k * stride + 1];
        mask[i] = (ty >= lo && ty < hi) ? 1 : 0;
    }
}
}  // namespace

extern "C" int gsr_band_mask(int grid_x, int grid_y, int B, const int32_t *band_rows_dev, int stride_words,
                             uint8_t *mask, gsr_stream_t stream_) {
    if (grid_x <= 0 || grid_y <= 0 || B <= 0 || !band_rows_dev || stride_words < 2 || !mask) return GSR_EINVAL;
    const int n = grid_x * grid_y * B;
    hipLaunchKernelGGL(band_mask_kernel, dim3(min(gsr_div_up(n, 256), 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream_), grid_x, grid_y, B, band_rows_dev, stride_words, mask);
    GSR_LAUNCH_CHECK();
    return 0;
}

extern "C" int gsr_publish_flag(const uint32_t *flag_dev, const uint32_t *seq_dev, uint32_t *host_ring_pinned,
                                uint32_t slots, gsr_stream_t stream_) {
    if (!flag_dev || !seq_dev || !host_ring_pinned || slots == 0) return GSR_EINVAL;
    hipLaunchKernelGGL(publish_flag_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream_), flag_dev,
                       seq_dev, host_ring_pinned, slots);
    GSR_LAUNCH_CHECK();
    return 0;
}

extern "C" int gsr_stamp(const uint32_t *seq_dev, uint64_t *host_ring_pinned, uint32_t slots, uint32_t per_slot,
                         uint32_t index, gsr_stream_t stream_) {
    if (!host_ring_pinned || slots == 0 || index >= per_slot) return GSR_EINVAL;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream_), seq_dev,
                       reinterpret_cast<unsigned long long *>(host_ring_pinned), slots, per_slot, index);
    GSR_LAUNCH_CHECK();
    return 0;
}

extern "C" int gsr_flag_if_greater(const uint32_t *value_dev, uint32_t limit, uint32_t *flag_dev, uint32_t bit,
                                   uint32_t *host_copy_pinned, gsr_stream_t stream_) {
    if (!value_dev || !flag_dev) return GSR_EINVAL;
    hipLaunchKernelGGL(flag_if_greater_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream_), value_dev,
                       limit, flag_dev, bit, host_copy_pinned);
    GSR_LAUNCH_CHECK();
    return 0;
}

extern "C" int gsr_exchange_check(const int32_t *all_counts_dev, const int32_t *caps_dev, int W, int B, int me,
                                  uint64_t rendered_mask, int few, uint32_t *flag_dev, uint32_t bit_over,
                                  uint32_t bit_few, int32_t *host_copy_pinned, gsr_stream_t stream_) {
    if (!all_counts_dev || !caps_dev || !flag_dev || W < 1 || B < 1 || B > 64 || me < 0 || me >= W) return GSR_EINVAL;
    hipLaunchKernelGGL(exchange_check_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                       all_counts_dev, caps_dev, W, B, me, (unsigned long long)rendered_mask, few, flag_dev, bit_over,
                       bit_few, host_copy_pinned);
    GSR_LAUNCH_CHECK();
    return 0;
}

extern "C" int gsr_zero_async(void *ptr, size_t bytes, gsr_stream_t stream_) {
    if (bytes == 0) return 0;
    if (!ptr) return GSR_EINVAL;
    GSR_HIP(hipMemsetAsync(ptr, 0, bytes, reinterpret_cast<hipStream_t>(stream_)));
    return 0;
}

extern "C" int gsr_scatter_add_rows(int64_t n, const int32_t *idx, const float *src, float *dst,
                                    gsr_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (n < 0) return GSR_EINVAL;
    if (n == 0) return 0;
    if (!idx || !src || !dst) return GSR_EINVAL;
    long long blocks = (n * 9 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (long long)n, idx, src,
                       dst);
    GSR_LAUNCH_CHECK();
    return 0;
}

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
