// binning.hip -- K3..K7 for gfx950: which locally computed tiles each Gaussian touches, and the
// per-tile front-to-back lists.
//
// The reference stage list (analyze_statistic.py:1972-1991 of the reference) is a device-wide
// 64-bit (tile<<32 | depth) CUB sort over all D (tile, Gaussian) pairs.  Here the same ORDER is
// produced with far less HBM traffic by splitting the key:
//   1. stable radix sort of the P Gaussians by depth bits (4 x 8-bit passes over P pairs);
//   2. emit the D pairs in that depth order, (ty, tx)-major inside a Gaussian;
//   3. stable radix sort of the D pairs by tile id only (ceil(log2(tiles)/8) = 2 passes at 1080p/4K).
// A stable sort by tile id of a depth-ordered sequence is exactly the (tile, depth, arrival)
// order of SURVEY.md A.3.  Traffic: ~64 B per Gaussian + ~40 B per pair instead of ~200 B per pair.
//
// All primitives are hand-written for wave64: ballot-based stable multisplit inside a wave,
// LDS per-wave digit tables, three-phase device scan.
#include "common.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048 elements per workgroup

constexpr int RADIX_THREADS = 256;
constexpr int RADIX_ITEMS = 16;
constexpr int RADIX_TILE = RADIX_THREADS * RADIX_ITEMS;  // 4096 elements per workgroup
constexpr int RADIX_WAVES = RADIX_THREADS / GSR_WAVE;
constexpr int RADIX_DIGITS = 256;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread workgroup; returns the exclusive
// prefix, *total = workgroup sum.  `smem` holds >= 4 words.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *smem, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        const uint32_t s = smem[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ---------------------------------------------------------------- three-phase device scan (u32)
__global__ void __launch_bounds__(SCAN_THREADS) scan_reduce_kernel(const uint32_t *__restrict__ in, long long n,
                                                                   uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t smem[4];
    const long long base = (long long)blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const long long j = base + k * SCAN_THREADS + threadIdx.x;
        if (j < n) s += in[j];
    }
    uint32_t tot;
    block_exclusive_scan(s, smem, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of block_sums[0..nb) in place, grand total to block_sums[nb]
__global__ void __launch_bounds__(SCAN_THREADS) scan_spine_kernel(uint32_t *__restrict__ block_sums, int nb) {
    __shared__ uint32_t smem[4];
    uint32_t carry = 0;
    for (int base = 0; base < nb; base += SCAN_THREADS) {
        const int j = base + threadIdx.x;
        const uint32_t v = j < nb ? block_sums[j] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan(v, smem, &tot);
        if (j < nb) block_sums[j] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) block_sums[nb] = carry;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_final_kernel(const uint32_t *__restrict__ in,
                                                                  uint32_t *__restrict__ out, long long n,
                                                                  const uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t smem[4];
    // each thread owns SCAN_ITEMS CONSECUTIVE elements so that one workgroup scan suffices
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        s += v[k];
    }
    uint32_t tot;
    uint32_t run = block_exclusive_scan(s, smem, &tot) + block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    // grand total lands one past the end
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = block_sums[gridDim.x];
}

// single-workgroup exclusive scan for short arrays (histogram tables of the depth sort): one launch
// instead of three.  out[n] = total.  n <= 1024 * SMALL_SCAN_ITEMS.
constexpr int SMALL_SCAN_THREADS = 1024;
constexpr int SMALL_SCAN_ITEMS = 64;
__global__ void __launch_bounds__(SMALL_SCAN_THREADS) scan_small_kernel(const uint32_t *__restrict__ in,
                                                                        uint32_t *__restrict__ out, int n) {
    __shared__ uint32_t wsum[SMALL_SCAN_THREADS / 64];
    const int per = (n + SMALL_SCAN_THREADS - 1) / SMALL_SCAN_THREADS;
    const int base = threadIdx.x * per;
    uint32_t s = 0;
    for (int k = 0; k < per; k++) s += (base + k < n) ? in[base + k] : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(s);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
    for (int w = 0; w < SMALL_SCAN_THREADS / 64; w++) {
        const uint32_t v = wsum[w];
        if (w < wave) wbase += v;
        tot += v;
    }
    uint32_t run = wbase + inc - s;
    for (int k = 0; k < per; k++) {
        if (base + k < n) {
            const uint32_t v = in[base + k];
            out[base + k] = run;
            run += v;
        }
    }
    if (threadIdx.x == 0) out[n] = tot;
}

// ------------------------------------------------------------------------------ radix sort pass
// digit = (key >> shift) & mask, mask = 2^nbits - 1, nbits <= 8
__global__ void __launch_bounds__(RADIX_THREADS) radix_hist_kernel(const uint32_t *__restrict__ keys, long long n,
                                                                   int shift, uint32_t mask,
                                                                   uint32_t *__restrict__ hist, int nb) {
    __shared__ uint32_t h[RADIX_DIGITS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * RADIX_TILE;
#pragma unroll
    for (int k = 0; k < RADIX_ITEMS; k++) {
        const long long j = base + k * RADIX_THREADS + threadIdx.x;
        if (j < n) atomicAdd(&h[(keys[j] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= mask) hist[(size_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];  // digit-major
}

// lanes holding the same digit (restricted to `valid` lanes)
__device__ __forceinline__ unsigned long long match_digit(uint32_t d, bool valid, int nbits) {
    unsigned long long m = __ballot(valid);
    for (int b = 0; b < nbits; b++) {
        const bool bit = (d >> b) & 1;
        const unsigned long long bm = __ballot(bit);
        m &= bit ? bm : ~bm;
    }
    return m;
}

// One workgroup = 4096 consecutive pairs.  Stable ranks come from wave64 ballots (match_digit) plus
// per-wave LDS cursors; the pairs are first placed in digit order INSIDE LDS and then streamed out,
// so that each digit's run leaves the workgroup as contiguous, coalesced stores.
__global__ void __launch_bounds__(RADIX_THREADS)
radix_scatter_kernel(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                     uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, long long n, int shift,
                     int nbits, const uint32_t *__restrict__ offsets, int nb) {
    __shared__ uint32_t wtab[RADIX_WAVES][RADIX_DIGITS];
    __shared__ uint32_t gbase[RADIX_DIGITS];  // global start of the digit's run minus its local start
    __shared__ uint32_t skey[RADIX_TILE], sval[RADIX_TILE];
    __shared__ uint32_t scan_tmp[4];
    const uint32_t mask = (1u << nbits) - 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < RADIX_WAVES; w++) wtab[w][threadIdx.x] = 0;
    __syncthreads();
    // wave w owns the contiguous sub-chunk [base + w*1024, +1024), walked in 16 rounds of 64
    const long long bbase = (long long)blockIdx.x * RADIX_TILE;
    const long long wbase = bbase + (long long)wave * (RADIX_ITEMS * 64);
    uint32_t key[RADIX_ITEMS];
#pragma unroll
    for (int r = 0; r < RADIX_ITEMS; r++) {
        const long long j = wbase + r * 64 + lane;
        key[r] = j < n ? keys_in[j] : 0xFFFFFFFFu;
        if (j < n) atomicAdd(&wtab[wave][(key[r] >> shift) & mask], 1u);
    }
    __syncthreads();
    {  // thread d: per-wave counts of digit d -> local (in-workgroup) start positions
        const int d = threadIdx.x;
        uint32_t cnt[RADIX_WAVES], tot = 0;
#pragma unroll
        for (int w = 0; w < RADIX_WAVES; w++) {
            cnt[w] = wtab[w][d];
            tot += cnt[w];
        }
        uint32_t all;
        uint32_t run = block_exclusive_scan(tot, scan_tmp, &all);  // local start of digit d
        gbase[d] = ((uint32_t)d <= mask ? offsets[(size_t)d * nb + blockIdx.x] : 0u) - run;
#pragma unroll
        for (int w = 0; w < RADIX_WAVES; w++) {
            wtab[w][d] = run;
            run += cnt[w];
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RADIX_ITEMS; r++) {
        const long long j = wbase + r * 64 + lane;
        const bool valid = j < n;
        const uint32_t d = (key[r] >> shift) & mask;
        const unsigned long long m = match_digit(d, valid, nbits);
        const uint32_t rank = __popcll(m & lt);
        volatile uint32_t *cursor = wtab[wave];
        uint32_t pos = 0;
        if (valid) pos = cursor[d] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) cursor[d] = pos + (uint32_t)__popcll(m);  // group leader advances the cursor
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            skey[pos] = key[r];
            sval[pos] = vals_in[j];
        }
    }
    __syncthreads();
    const long long rem = n - bbase;
    const int count = rem < RADIX_TILE ? (int)rem : RADIX_TILE;
#pragma unroll
    for (int r = 0; r < RADIX_ITEMS; r++) {
        const int i = r * RADIX_THREADS + threadIdx.x;
        if (i < count) {
            const uint32_t k = skey[i];
            const uint32_t dst = gbase[(k >> shift) & mask] + (uint32_t)i;
            keys_out[dst] = k;
            vals_out[dst] = sval[i];
        }
    }
}

// ---------------------------------------------------------------------------------- K3 and friends
// Row hull of the locally computed tiles: hull[0] = first tile row with a local tile, hull[1] = one
// past the last.  Grendel's final mode always passes whole-row bands, for which the hull IS the mask;
// for a general mask the tiles inside the hull that are not local are emitted with a sentinel key.
__global__ void __launch_bounds__(256) mask_hull_kernel(const uint8_t *__restrict__ mask, int gx, int gy,
                                                        int32_t *__restrict__ hull) {
    __shared__ int lo, hi;
    if (threadIdx.x == 0) { lo = gy; hi = 0; }
    __syncthreads();
    for (int y = threadIdx.x; y < gy; y += blockDim.x) {
        bool any = false;
        for (int x = 0; x < gx; x++) any = any || mask[(size_t)y * gx + x];
        if (any) {
            atomicMin(&lo, y);
            atomicMax(&hi, y + 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { hull[0] = lo; hull[1] = hi; }
}

// K3: per Gaussian, the tile rect it can CONTRIBUTE to and the number of its tiles; depth sort keys.
// The rect is the reference's 3-sigma-radius rect intersected with the
// bounding box of the alpha >= 1/255 ellipse (gsr_alpha_extent) and with the mask's row hull: tiles
// dropped by the intersection cannot receive a contribution under the alpha < 1/255 rule of A.4, so
// the image is unchanged while D (pairs to sort and to walk) shrinks.  Gaussians touching nothing get
// key 0xFFFFFFFF (depths are > 0.2, so real keys are < 0x7F800000) and sort to the end.
__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
touch_count_kernel(int P, int gx, int gy, const float2 *__restrict__ means2D, const float *__restrict__ depths,
                   const int32_t *__restrict__ radii, const float4 *__restrict__ conic_opacity,
                   const int32_t *__restrict__ hull, uint32_t *__restrict__ tt, uint32_t *__restrict__ keys,
                   uint32_t *__restrict__ vals, uint2 *__restrict__ rects) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t n = 0;
    uint2 rect = make_uint2(0u, 0u);
    const int rad = radii[i];
    if (rad > 0) {
        const float2 xy = means2D[i];
        const float4 co = conic_opacity[i];
        float ex, ey;
        if (gsr_alpha_extent(co, ex, ey)) {
            int minx, miny, maxx, maxy;
            gsr_get_rect(xy.x, xy.y, rad, gx, gy, minx, miny, maxx, maxy);
            // tile t covers pixel centres [16t, 16t+15]
            minx = max(minx, (int)ceilf((xy.x - ex - (GSR_BLOCK_X - 1)) * (1.0f / GSR_BLOCK_X)));
            maxx = min(maxx, (int)floorf((xy.x + ex) * (1.0f / GSR_BLOCK_X)) + 1);
            miny = max(max(miny, hull[0]), (int)ceilf((xy.y - ey - (GSR_BLOCK_Y - 1)) * (1.0f / GSR_BLOCK_Y)));
            maxy = min(min(maxy, hull[1]), (int)floorf((xy.y + ey) * (1.0f / GSR_BLOCK_Y)) + 1);
            if (maxx > minx && maxy > miny) {
                n = (uint32_t)((maxx - minx) * (maxy - miny));
                rect = make_uint2((uint32_t)minx | ((uint32_t)maxx << 16), (uint32_t)miny | ((uint32_t)maxy << 16));
            }
        }
    }
    tt[i] = n;
    rects[i] = rect;
    keys[i] = n ? __float_as_uint(depths[i]) : 0xFFFFFFFFu;
    vals[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
gather_u32_kernel(int n, const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx,
                  uint32_t *__restrict__ dst) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[j] = src[idx[j]];
}

// K5: the D output pairs are cut into chunks of EMIT_CHUNK slots, one wave per chunk, so the work is
// balanced by OUTPUT (a near-camera splat covering thousands of tiles no longer serialises one wave).
// The wave finds the Gaussian that owns its first slot with a 64-ary search over the offsets (one
// coalesced probe per round), then walks windows of 64 depth-consecutive Gaussians staged in LDS; every
// slot locates its owner by a 6-step binary search in the window and is written with coalesced stores.
constexpr int EMIT_CHUNK = 2048;
__global__ void __launch_bounds__(256)
emit_pairs_kernel(int P, long long D, int gx, int tiles, const uint2 *__restrict__ rects,
                  const uint8_t *__restrict__ mask, const uint32_t *__restrict__ sorted_ids,
                  const uint32_t *__restrict__ offsets, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    __shared__ uint32_t s_off[4][65];
    __shared__ uint32_t s_g[4][64];
    __shared__ uint2 s_rect[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long chunk = (long long)blockIdx.x * 4 + wave;
    const long long sb = chunk * EMIT_CHUNK;
    if (sb >= D) return;
    const uint32_t s_begin = (uint32_t)sb;
    const uint32_t s_end = (uint32_t)(sb + EMIT_CHUNK < D ? sb + EMIT_CHUNK : D);
    // largest j in [0, P] with offsets[j] <= s_begin  (offsets is non-decreasing, offsets[P] = D > s_begin)
    int lo = 0, hi = P;  // invariant: offsets[lo] <= s_begin < offsets[hi]
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) / 64;
        const int idx = min(lo + lane * step, hi);
        const bool le = offsets[idx] <= s_begin;
        const int c = __popcll(__ballot(le));  // probes are monotone: the first c lanes say "<="
        const int nlo = lo + (c - 1) * step;
        hi = min(hi, nlo + step);
        lo = nlo;
    }
    int g0 = lo;
    uint32_t s = s_begin + lane;
    while (true) {  // one window of 64 Gaussians per iteration (wave-uniform control flow)
        const int j = g0 + lane;
        const uint32_t off = offsets[min(j, P)];
        const uint32_t end = offsets[min(j + 1, P)];
        const uint32_t g = (j < P && end > off) ? sorted_ids[j] : 0u;
        __builtin_amdgcn_wave_barrier();
        s_off[wave][lane] = off;
        if (lane == 63) s_off[wave][64] = end;
        s_g[wave][lane] = g;
        s_rect[wave][lane] = (j < P && end > off) ? rects[g] : make_uint2(0u, 0u);
        __builtin_amdgcn_wave_barrier();
        const uint32_t wend = min(__builtin_amdgcn_readlane(end, 63), s_end);
        for (; s < wend; s += 64) {
            int a = 0, bnd = 63;
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const int mid = (a + bnd + 1) >> 1;
                if (s_off[wave][mid] <= s) a = mid; else bnd = mid - 1;
            }
            const uint2 r = s_rect[wave][a];
            const uint32_t t = s - s_off[wave][a];
            const uint32_t minx = r.x & 0xFFFFu, w = (r.x >> 16) - minx, miny = r.y & 0xFFFFu;
            const uint32_t y = miny + t / w, x = minx + t % w;
            const uint32_t tile = y * (uint32_t)gx + x;
            keys[s] = mask[tile] ? tile : (uint32_t)tiles;  // non-local tile inside the hull: sentinel, sorts last
            vals[s] = s_g[wave][a];
        }
        if (wend >= s_end) break;
        g0 += 64;
    }
}

// K7
__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
tile_ranges_kernel(long long D, uint32_t tiles, const uint32_t *__restrict__ tile_of, int2 *__restrict__ ranges) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    const uint32_t t = tile_of[j];
    if (t >= tiles) return;
    if (j == 0 || tile_of[j - 1] != t) ranges[t].x = (int)j;
    if (j == D - 1 || tile_of[j + 1] != t) ranges[t].y = (int)(j + 1);
}

__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
copy_u32_kernel(long long n, const uint32_t *__restrict__ src, uint32_t *__restrict__ dst) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[j] = src[j];
}

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

// ----------------------------------------------------------------------------------- primitives
size_t gsr_scan_temp_bytes(long long n) {
    const long long nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    return align_up((size_t)(nb + 1) * sizeof(uint32_t));
}

// out must hold n+1 words: out[i] = sum(in[0..i)), out[n] = total.  in == out is allowed.
int gsr_exclusive_scan_u32(const uint32_t *in, uint32_t *out, long long n, void *temp, hipStream_t stream) {
    if (n <= 0) {
        GSR_HIP(hipMemsetAsync(out, 0, sizeof(uint32_t), stream));
        return 0;
    }
    if (n <= SMALL_SCAN_THREADS) {  // one element per thread; longer arrays: the coalesced three-phase scan
        hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(SMALL_SCAN_THREADS), 0, stream, in, out, (int)n);
        GSR_LAUNCH_CHECK();
        return 0;
    }
    const int nb = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    uint32_t *bs = reinterpret_cast<uint32_t *>(temp);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(SCAN_THREADS), 0, stream, in, n, bs);
    hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, bs, nb);
    hipLaunchKernelGGL(scan_final_kernel, dim3(nb), dim3(SCAN_THREADS), 0, stream, in, out, n, bs);
    GSR_LAUNCH_CHECK();
    return 0;
}

size_t gsr_radix_temp_bytes(long long n) {
    const long long nb = (n + RADIX_TILE - 1) / RADIX_TILE;
    const long long nh = nb * RADIX_DIGITS;
    return align_up((size_t)(nh + 1) * sizeof(uint32_t)) + gsr_scan_temp_bytes(nh);
}

int gsr_radix_sort_pairs(uint32_t *k0, uint32_t *v0, uint32_t *k1, uint32_t *v1, long long n, int bit_lo, int bit_hi,
                         void *temp, int *result_in_first, hipStream_t stream, uint32_t *final_vals) {
    *result_in_first = 1;
    if (n <= 0) return 0;
    const int nb = (int)((n + RADIX_TILE - 1) / RADIX_TILE);
    uint32_t *hist = reinterpret_cast<uint32_t *>(temp);
    void *scan_temp = reinterpret_cast<char *>(temp) + align_up((size_t)((long long)nb * RADIX_DIGITS + 1) * sizeof(uint32_t));
    // split the key bits evenly over the minimum number of <= 8-bit passes (13 tile bits -> 7 + 6)
    const int total = bit_hi - bit_lo, passes = (total + 7) / 8;
    uint32_t *ki = k0, *vi = v0, *ko = k1, *vo = v1;
    int shift = bit_lo;
    for (int p = 0; p < passes; p++) {
        const int nbits = (total - (shift - bit_lo) + (passes - p) - 1) / (passes - p);
        const long long nh = (long long)nb << nbits;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nb), dim3(RADIX_THREADS), 0, stream, ki, n, shift,
                           (1u << nbits) - 1u, hist, nb);
        int rc = gsr_exclusive_scan_u32(hist, hist, nh, scan_temp, stream);
        if (rc) return rc;
        uint32_t *vdst = (p == passes - 1 && final_vals) ? final_vals : vo;  // last pass can land the values
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nb), dim3(RADIX_THREADS), 0, stream, ki, vi, ko, vdst, n, shift,
                           nbits, hist, nb);
        shift += nbits;
        uint32_t *t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
        *result_in_first ^= 1;
    }
    GSR_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------------- K3..K7 API
namespace {
struct PrepLayout {
    size_t tt, kA, vA, kB, vB, offsets, rects, hull, temp, total;
};
PrepLayout prep_layout(int P, int W, int H) {
    (void)W; (void)H;
    PrepLayout L;
    size_t o = 0;
    const size_t np = align_up((size_t)(P + 1) * 4);
    L.tt = o; o += np;
    L.kA = o; o += np;
    L.vA = o; o += np;
    L.kB = o; o += np;
    L.vB = o; o += np;
    L.offsets = o; o += np;
    L.rects = o; o += align_up((size_t)(P + 1) * 8);
    L.hull = o; o += 256;
    L.temp = o;
    const size_t t1 = gsr_radix_temp_bytes(P), t2 = gsr_scan_temp_bytes(P);
    o += t1 > t2 ? t1 : t2;
    L.total = o;
    return L;
}
int tile_bits(int tiles) {  // bits of the largest key value, the sentinel `tiles`
    int b = 1;
    while ((1ll << b) <= tiles) b++;
    return b;
}
}  // namespace

extern "C" size_t gsr_bin_prepare_bytes(int P, int width, int height) {
    if (P < 0 || width <= 0 || height <= 0) return 0;
    return prep_layout(P, width, height).total;
}

extern "C" int gsr_bin_prepare(int P, int width, int height, const float *means2D, const float *depths,
                               const int32_t *radii, const float *conic_opacity, const uint8_t *compute_locally,
                               void *prep, size_t prep_bytes, int64_t *num_rendered_host, gsr_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || width <= 0 || height <= 0 || !num_rendered_host) return GSR_EINVAL;
    *num_rendered_host = 0;
    if (P == 0) return 0;
    if (!means2D || !depths || !radii || !conic_opacity || !compute_locally || !prep) return GSR_EINVAL;
    const PrepLayout L = prep_layout(P, width, height);
    if (prep_bytes < L.total) return GSR_ENOSPACE;
    const int gx = (width + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    if (gx > 0xFFFF || gy > 0xFFFF) return GSR_EINVAL;
    char *base = reinterpret_cast<char *>(prep);
    uint32_t *tt = reinterpret_cast<uint32_t *>(base + L.tt);
    uint32_t *kA = reinterpret_cast<uint32_t *>(base + L.kA), *vA = reinterpret_cast<uint32_t *>(base + L.vA);
    uint32_t *kB = reinterpret_cast<uint32_t *>(base + L.kB), *vB = reinterpret_cast<uint32_t *>(base + L.vB);
    uint32_t *offsets = reinterpret_cast<uint32_t *>(base + L.offsets);
    uint2 *rects = reinterpret_cast<uint2 *>(base + L.rects);
    int32_t *hull = reinterpret_cast<int32_t *>(base + L.hull);
    void *temp = base + L.temp;

    hipLaunchKernelGGL(mask_hull_kernel, dim3(1), dim3(256), 0, stream, compute_locally, gx, gy, hull);
    hipLaunchKernelGGL(touch_count_kernel, dim3(gsr_div_up(P, GSR_ONE_DIM_BLOCK)), dim3(GSR_ONE_DIM_BLOCK), 0, stream,
                       P, gx, gy, reinterpret_cast<const float2 *>(means2D), depths, radii,
                       reinterpret_cast<const float4 *>(conic_opacity), hull, tt, kA, vA, rects);
    int in_first = 1;
    int rc = gsr_radix_sort_pairs(kA, vA, kB, vB, P, 0, 32, temp, &in_first, stream, nullptr);
    if (rc) return rc;
    // 4 passes -> back in (kA, vA); keep the sorted ids in vA, reuse kB for the gathered counts
    uint32_t *sorted_ids = in_first ? vA : vB;
    if (!in_first) {
        hipLaunchKernelGGL(copy_u32_kernel, dim3(gsr_div_up(P, GSR_ONE_DIM_BLOCK)), dim3(GSR_ONE_DIM_BLOCK), 0, stream,
                           (long long)P, vB, vA);
        sorted_ids = vA;
    }
    hipLaunchKernelGGL(gather_u32_kernel, dim3(gsr_div_up(P, GSR_ONE_DIM_BLOCK)), dim3(GSR_ONE_DIM_BLOCK), 0, stream, P,
                       tt, sorted_ids, kB);
    rc = gsr_exclusive_scan_u32(kB, offsets, P, temp, stream);
    if (rc) return rc;
    GSR_LAUNCH_CHECK();
    uint32_t total = 0;
    GSR_HIP(hipMemcpyAsync(&total, offsets + P, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GSR_HIP(hipStreamSynchronize(stream));
    *num_rendered_host = (int64_t)total;
    return 0;
}

namespace {
struct SortLayout {
    size_t kA, vA, kB, vB, temp, total;
};
SortLayout sort_layout(int64_t D) {
    SortLayout L;
    size_t o = 0;
    const size_t nd = align_up((size_t)(D + 1) * 4);
    L.kA = o; o += nd;
    L.vA = o; o += nd;
    L.kB = o; o += nd;
    L.vB = o; o += nd;
    L.temp = o; o += gsr_radix_temp_bytes(D);
    L.total = o;
    return L;
}
}  // namespace

extern "C" size_t gsr_bin_sort_bytes(int P, int64_t num_rendered, int width, int height) {
    (void)P; (void)width; (void)height;
    if (num_rendered < 0) return 0;
    return sort_layout(num_rendered).total;
}

extern "C" int gsr_bin_sort(int P, int width, int height, const uint8_t *compute_locally, const void *prep,
                            int64_t D, void *scratch, size_t scratch_bytes, uint32_t *point_list, int32_t *ranges,
                            gsr_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || width <= 0 || height <= 0 || D < 0 || !ranges) return GSR_EINVAL;
    const int gx = (width + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    GSR_HIP(hipMemsetAsync(ranges, 0, sizeof(int32_t) * 2 * (size_t)gx * gy, stream));
    if (D == 0 || P == 0) return 0;
    if (!compute_locally || !prep || !scratch || !point_list) return GSR_EINVAL;
    const SortLayout S = sort_layout(D);
    if (scratch_bytes < S.total) return GSR_ENOSPACE;
    const PrepLayout L = prep_layout(P, width, height);
    const char *pbase = reinterpret_cast<const char *>(prep);
    const uint32_t *sorted_ids = reinterpret_cast<const uint32_t *>(pbase + L.vA);
    const uint32_t *offsets = reinterpret_cast<const uint32_t *>(pbase + L.offsets);
    const uint2 *rects = reinterpret_cast<const uint2 *>(pbase + L.rects);
    char *sbase = reinterpret_cast<char *>(scratch);
    uint32_t *kA = reinterpret_cast<uint32_t *>(sbase + S.kA), *vA = reinterpret_cast<uint32_t *>(sbase + S.vA);
    uint32_t *kB = reinterpret_cast<uint32_t *>(sbase + S.kB), *vB = reinterpret_cast<uint32_t *>(sbase + S.vB);
    void *temp = sbase + S.temp;

    hipLaunchKernelGGL(emit_pairs_kernel, dim3(gsr_div_up(gsr_div_up(D, EMIT_CHUNK), 4)), dim3(256), 0, stream, P,
                       (long long)D, gx, gx * gy, rects, compute_locally, sorted_ids, offsets, kA, vA);
    int in_first = 1;
    // the last pass writes the Gaussian indices straight into point_list
    int rc = gsr_radix_sort_pairs(kA, vA, kB, vB, D, 0, tile_bits(gx * gy), temp, &in_first, stream, point_list);
    if (rc) return rc;
    const uint32_t *ks = in_first ? kA : kB;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(gsr_div_up(D, GSR_ONE_DIM_BLOCK)), dim3(GSR_ONE_DIM_BLOCK), 0, stream,
                       (long long)D, (uint32_t)(gx * gy), ks, reinterpret_cast<int2 *>(ranges));
    GSR_LAUNCH_CHECK();
    return 0;
}
