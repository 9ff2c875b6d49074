// composite.hip -- K8 (alpha-composite forward) and K10 (its backward) for gfx950 / wave64.
//
// Unit of work: ONE WAVE = one 8x8 pixel quadrant of a 16x16 tile (lane = pixel); a 256-thread
// workgroup is the four quadrants of a tile.  In the forward the waves never synchronise with each other:
//   * each wave walks the tile's depth-sorted list in chunks of 64 entries, one entry per lane
//     (coalesced index read + one 36-byte gather per lane);
//   * each lane tests ITS entry against the wave's quadrant (exact: minimum of the conic's quadratic form over
//     the box vs the alpha >= 1/255 level) and a 64-bit ballot gives the entries that can touch the quadrant;
//   * the chunk is staged in a wave-private LDS slab (lane k stores entry k) and the wave iterates over the set
//     bits only, reading entry k back with a wave-uniform address: an LDS broadcast costs no VALU issue slot
//     (9 v_readlane per entry did), and VALU issue is what bounds both kernels (profiles/r02_pmc.txt:
//     SQ_ACTIVE_INST_VALU covers 0.9-1.07 of the SIMD cycles);
//   * early termination is per wave: __all(done) leaves the loop.
// Entries skipped by the quadrant test would have been rejected per pixel by the alpha < 1/255 rule,
// so the image and n_contrib are those of the plain algorithm.
//
// Backward: the same walk in reverse order; the per-pixel products are transposed through LDS and contracted
// over the 64 pixels of the quadrant on the matrix pipe (exact-fp32 MFMA) -- see the comment above K10.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_STOP = 0.0001f;

#ifdef GSR_STATS
__device__ unsigned long long g_stats[8];
#define GSR_STAT(i, v) do { const unsigned long long sv__ = (unsigned long long)(v); if (lane == 0) atomicAdd(&g_stats[i], sv__); } while (0)
#else
#define GSR_STAT(i, v) do { } while (0)
#endif

// lane-resident entry of the tile list
struct Entry {
    float x, y;        // pixel centre
    float a2, b2, c2;  // conic pre-scaled to log2 units: p2 = a2 dx^2 + b2 dx dy + c2 dy^2 = power * log2(e)
    float o;           // opacity, or log2(opacity) with GSR_LOG2O (see entry_exponent)
    bool relevant;
};

// alpha = min(0.99, o exp(power)) of a broadcast entry (q0 = x, y, a2, b2; q1 = c2, o, r, g) at offset (dx, dy).
// GSR_LOG2O (default): the opacity is folded into the exponent, o exp2(p2) = exp2(p2 + log2 o), one multiply less
// per (pixel, entry); K8 and K10 share these helpers, so the forward and the backward always take the SAME
// skip / blend decisions.  `power > 0 -> skip` becomes  exponent > log2 o.
#ifndef GSR_NO_LOG2O
#define GSR_LOG2O 1
#endif
__device__ __forceinline__ float entry_exponent(const float4 q0, const float4 q1, float dx, float dy) {
#ifdef GSR_LOG2O
    return fmaf(fmaf(q0.z, dx, q0.w * dy), dx, fmaf(q1.x * dy, dy, q1.y));
#else
    return (q0.z * dx + q0.w * dy) * dx + q1.x * dy * dy;
#endif
}
__device__ __forceinline__ float entry_alpha_raw(const float4 q1, float pe) {
#ifdef GSR_LOG2O
    return __builtin_amdgcn_exp2f(pe);
#else
    return q1.y * __builtin_amdgcn_exp2f(pe);
#endif
}
__device__ __forceinline__ bool entry_power_ok(const float4 q1, float pe) {
#ifdef GSR_LOG2O
    return pe <= q1.y;
#else
    return pe <= 0.f;
#endif
}

__device__ __forceinline__ float entry_opacity_term(float o) {
#ifdef GSR_LOG2O
    return __log2f(o);
#else
    return o;
#endif
}

// Load entry `idx` (or an inert one) and test it against the quadrant [qx0,qx0+7]x[qy0,qy0+7].
__device__ __forceinline__ Entry load_entry(bool have, uint32_t id, const float2 *__restrict__ means2D,
                                            const float4 *__restrict__ conic_opacity, float qx0, float qy0) {
    Entry e;
    e.x = e.y = e.a2 = e.b2 = e.c2 = e.o = 0.f;
    e.relevant = false;
    if (have) {
        const float2 xy = means2D[id];
        const float4 co = conic_opacity[id];
        e.x = xy.x;
        e.y = xy.y;
        e.a2 = -0.5f * LOG2E * co.x;
        e.b2 = -LOG2E * co.y;
        e.c2 = -0.5f * LOG2E * co.z;
        e.o = entry_opacity_term(co.w);
        e.relevant = gsr_can_touch_box(xy, co, qx0, qy0, qx0 + 7.0f, qy0 + 7.0f);
    }
    return e;
}

// ------------------------------------------------------------------------------------------- K8
__global__ void __launch_bounds__(256)
composite_forward_kernel(int W, int H, int gx, const int2 *__restrict__ ranges,
                         const uint32_t *__restrict__ point_list, const float2 *__restrict__ means2D,
                         const float4 *__restrict__ conic_opacity, const float *__restrict__ rgb,
                         const uint8_t *__restrict__ compute_locally, const float *__restrict__ bg,
                         float *__restrict__ out_color, float *__restrict__ final_T, int32_t *__restrict__ n_contrib) {
    const int tile = gsr_xcd_span_of_block(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx = tile % gx, ty = tile / gx;
    const int qx0 = tx * GSR_BLOCK_X + (wave & 1) * 8, qy0 = ty * GSR_BLOCK_Y + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t pid = (size_t)py * W + px;
    const size_t HW = (size_t)H * W;

    if (!compute_locally[tile]) {  // not ours: pixels must be exactly 0 (SUM all-reduce assembly)
        if (inside) {
            out_color[pid] = 0.f;
            out_color[HW + pid] = 0.f;
            out_color[2 * HW + pid] = 0.f;
            final_T[pid] = 1.f;
            n_contrib[pid] = 0;
        }
        return;
    }
    const int2 range = ranges[tile];
    const int n = range.y - range.x;
    const float pxf = (float)px, pyf = (float)py;

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    int last = 0;
    bool done = !inside;
    __shared__ float4 slab[4][64][3];

    for (int c = 0; c < n; c += 64) {
        if (__all(done)) break;
        const bool have = c + lane < n;
        const uint32_t id = have ? point_list[range.x + c + lane] : 0u;
        const Entry e = load_entry(have, id, means2D, conic_opacity, (float)qx0, (float)qy0);
        float r = 0.f, g = 0.f, b = 0.f;
        if (e.relevant) {
            r = rgb[3 * (size_t)id];
            g = rgb[3 * (size_t)id + 1];
            b = rgb[3 * (size_t)id + 2];
        }
        unsigned long long m = __ballot(e.relevant);
        GSR_STAT(0, __popcll(__ballot(have)));
        GSR_STAT(1, __popcll(m));
        // Stage the chunk in this wave's private LDS slab (lane k writes entry k) and broadcast-read it back:
        // an LDS read with a wave-uniform address returns the entry to all 64 lanes WITHOUT spending VALU
        // issue slots (9 v_readlane per entry before), and the VALU is what bounds this kernel.  No barrier:
        // the slab is private to the wave and LDS operations of one wave execute in order.
        slab[wave][lane][0] = make_float4(e.x, e.y, e.a2, e.b2);
        slab[wave][lane][1] = make_float4(e.c2, e.o, r, g);
        slab[wave][lane][2] = make_float4(b, 0.f, 0.f, 0.f);
        // Entries are taken UF at a time: their alphas are independent (ILP across the LDS and v_exp latency),
        // only the short T / colour chain is sequential, and the wave tests "everybody done?" once per group.
        constexpr int UF = 4;
        while (m) {
            int kk[UF];
            float al[UF], cr[UF], cg[UF], cb[UF];
            bool ok[UF];
#pragma unroll
            for (int u = 0; u < UF; u++) {
                const bool live = m != 0;  // wave-uniform
                const int k = live ? __builtin_ctzll(m) : 0;
                m &= m - 1;  // 0 & ~0 stays 0
                kk[u] = k;
                const float4 q0 = slab[wave][k][0], q1 = slab[wave][k][1];
                cb[u] = slab[wave][k][2].x;
                cr[u] = q1.z;
                cg[u] = q1.w;
                const float dx = q0.x - pxf, dy = q0.y - pyf;
                const float pe = entry_exponent(q0, q1, dx, dy);
                al[u] = fminf(0.99f, entry_alpha_raw(q1, pe));
                ok[u] = live && entry_power_ok(q1, pe) && al[u] >= ALPHA_MIN;
            }
#pragma unroll
            for (int u = 0; u < UF; u++) {
                const bool take = !done && ok[u];
                const float test_T = T * (1.0f - al[u]);
                const bool stop = take && test_T < T_STOP;
                done = done || stop;
                const bool blend = take && !stop;
                const float w = blend ? al[u] * T : 0.f;
                C0 += cr[u] * w;
                C1 += cg[u] * w;
                C2 += cb[u] * w;
                T = blend ? test_T : T;
                last = blend ? c + kk[u] + 1 : last;
            }
            if (__all(done)) break;
        }
    }
    if (inside) {
        out_color[pid] = C0 + T * bg[0];
        out_color[HW + pid] = C1 + T * bg[1];
        out_color[2 * HW + pid] = C2 + T * bg[2];
        final_T[pid] = T;
        n_contrib[pid] = last;
    }
}

// ------------------------------------------------------------------------------------------ K10
// Backward of the composite.  Per (pixel, entry) the blend gives two numbers: w = alpha * T (weight of the entry's
// colour in the pixel) and q = o G dL/dalpha (with G = exp(power); the min(0.99, .) clamp is the identity in the
// backward).  Every gradient of the entry is a sum over the pixels of those two numbers times a
// polynomial of the pixel position:
//     dL/drgb_c        = sum_p w_p g_pc                      (g = dL/dpixel)
//     dL/dopacity      = sum_p q_p / o
//     dL/dmean, dconic = linear maps (applied ONCE per (tile, entry)) of  sum_p q_p {dx, dy, dx^2, dx dy, dy^2},
// and with dx = x_e - px the five moments follow from the RAW pixel moments  sum_p q_p {1, px, py, px^2, px py, py^2}
// (px, py relative to the tile origin: integers 0..15) by the binomial expansion.  So the per-tile reduction IS a
// dense contraction over the 64 pixels of a quadrant:  [entries x pixels] . [pixels x 9].  It runs on the MATRIX
// pipe (v_mfma_f32_16x16x4_f32: exact fp32, an fmaf chain) instead of as a cross-lane VALU reduction:
//   phase A (lane = pixel): walk the entries back to front, two per iteration; recompute alpha, carry T and
//       rho = R.g (R = colour behind the entry: ONE scalar recurrence instead of three), and store (q, w) of the entry
//       into a wave-private LDS matrix [slot][pixel] -- ~28 VALU instructions per (wave, entry) where the transposed
//       butterfly reduction this replaces needed ~80 (SQ_INSTS_VALU 277 M -> 160 M per launch on the bench view);
//   phase B (every 8 entries): 16 K-steps of 4 pixels.  Lane (k = lane >> 4, i = lane & 15) reads component i >> 3 of
//       slot i & 7, pixel 4 t + k -- the A operand: rows 0-7 are the q-rows of the eight slots, rows 8-15 their w-rows;
//       conflict-free with the row stride of 66 -- and issues ONE v_mfma_f32_16x16x4_f32 against a per-lane constant
//       B operand (columns 0-5 the position polynomials, 6-8 the pixel's dL/dcolour).  Rows 0-7 x columns 0-5 of the
//       result are the q moments, rows 8-15 x columns 6-8 the colour gradients; they are STORED into the wave's own
//       per-chunk table (an entry is in exactly one batch of a wave; LDS float atomics measured ~50 cycles each);
//   flush (once per 64-entry chunk, after a workgroup barrier): the four quadrant waves' tables are added, moments ->
//       gradients, and 9 adjacent lanes add one (tile, entry) pair's nine values into that Gaussian's 36-byte row
//       of the [P,9] record (means2D 0:2, rgb 2:5, conic_opacity 5:9); positions / conics / ids come from the slab
//       of the wave that walks the longest list (no global loads in the flush).
// The list entries of a chunk are prefetched one chunk ahead (their indices two).  The matrix pipe runs beside the
// VALU (16 MFMAs per 8 entries = 64 matrix-pipe cycles per entry, ~19 % busy); 4 workgroups per CU (38 KB of LDS,
// <= 128 VGPRs).  Measured (profiles/r02_*): 0.496 -> 0.384 ms per launch at 1 M Gaussians / 1080p, VALU busy 0.61,
// LDS 0.32, half of each wave's life still waits on LDS / barriers -- the kernel is latency-bound now, not VALU-bound.
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MB = 8;      // entries per MFMA batch: rows 0-7 of the 16 x 16 result are their q-rows, rows 8-15 their w-rows
constexpr int MSTR = 66;   // row stride of the (q, w) matrix in 8-byte elements: phase A writes and phase B reads conflict-free

// what a lane holds of one list entry between the load and the chunk that consumes it (software prefetch)
struct RawEntry {
    float2 xy;
    float4 co;
    float r, g, b;
};
__device__ __forceinline__ RawEntry load_raw(bool have, uint32_t id, const float2 *__restrict__ means2D,
                                             const float4 *__restrict__ conic_opacity, const float *__restrict__ rgb) {
    RawEntry e;
    e.xy = make_float2(0.f, 0.f);
    e.co = make_float4(0.f, 0.f, 0.f, 0.f);
    e.r = e.g = e.b = 0.f;
    if (have) {
        e.xy = means2D[id];
        e.co = conic_opacity[id];
        e.r = rgb[3 * (size_t)id];
        e.g = rgb[3 * (size_t)id + 1];
        e.b = rgb[3 * (size_t)id + 2];
    }
    return e;
}

__global__ void __launch_bounds__(256, 4)  // 4 waves per SIMD: <= 128 VGPRs (LDS admits 4 workgroups per CU)
composite_backward_kernel(int W, int H, int gx, const int2 *__restrict__ ranges,
                          const uint32_t *__restrict__ point_list, const float2 *__restrict__ means2D,
                          const float4 *__restrict__ conic_opacity, const float *__restrict__ rgb,
                          const uint8_t *__restrict__ compute_locally, const float *__restrict__ bg,
                          const float *__restrict__ final_T, const int32_t *__restrict__ n_contrib,
                          const float *__restrict__ dL_dpixels, float *__restrict__ dL_record) {
    const int tile = gsr_xcd_span_of_block(blockIdx.x, gridDim.x);
    if (!compute_locally[tile]) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx = tile % gx, ty = tile / gx;
    const int qx0 = tx * GSR_BLOCK_X + (wave & 1) * 8, qy0 = ty * GSR_BLOCK_Y + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t pid = (size_t)py * W + px;
    const size_t HW = (size_t)H * W;
    const int2 range = ranges[tile];
    const float pxf = (float)px, pyf = (float)py;

    const float T_final = inside ? final_T[pid] : 0.f;
    const int last = inside ? n_contrib[pid] : 0;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (inside) {
        g0 = dL_dpixels[pid];
        g1 = dL_dpixels[HW + pid];
        g2 = dL_dpixels[2 * HW + pid];
    }
    const float tb = T_final * (bg[0] * g0 + bg[1] * g1 + bg[2] * g2);
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    __shared__ float2 smat[4][MB * MSTR];   // [wave][slot * MSTR + pixel] = (q, w)
    __shared__ float sacc[4][64 * 9];       // [wave][entry * 9 + moment]: plain stores (LDS float atomics cost ~50 cycles)
    __shared__ float4 slab[4][64][3];       // [wave][entry] = (x, y, a2, b2 | c2, log2 o, r, g | b, id, o, -)
    __shared__ int s_wmax[4];
    int wmax = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    if (lane == 0) s_wmax[wave] = wmax;
    for (int i = threadIdx.x; i < 4 * 64 * 9; i += 256) (&sacc[0][0])[i] = 0.f;
    __syncthreads();
    const int bmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    if (bmax == 0) return;
    // the wave that walks the longest list has every entry of every chunk in its slab: the flush reads from it
    int wbest = 0;
#pragma unroll
    for (int w = 1; w < 4; w++)
        if (s_wmax[w] > s_wmax[wbest]) wbest = w;

    // ---- phase B operands of this lane (constant for the whole kernel).  v_mfma_f32_16x16x4_f32, lane l:
    //   A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], C[i = 4 (l >> 4) + r][j = l & 15] in register r.
    // K-step t contracts the four pixels 4 t + k.  Row i of A is slot i & 7; rows 0-7 carry q, rows 8-15 carry w.
    // Column j of B: 0-5 the position polynomials 1, x, y, x^2, x y, y^2 (pixel position about the TILE origin, so the
    // four quadrant waves add into one table), 6-8 the pixel's dL/dcolour.  Rows 0-7 x columns 0-5 are the q moments,
    // rows 8-15 x columns 6-8 the colour gradients; the other products are not used.
    const int kq = lane >> 4, j = lane & 15;
    float Bop[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int p = 4 * t + kq;
        const float xt = (float)((wave & 1) * 8 + (p & 7)), yt = (float)((wave >> 1) * 8 + (p >> 3));
        const int gxp = qx0 + (p & 7), gyp = qy0 + (p >> 3);
        float v = j == 0 ? 1.f : j == 1 ? xt : j == 2 ? yt : j == 3 ? xt * xt : j == 4 ? xt * yt : j == 5 ? yt * yt : 0.f;
        if (j >= 6 && j < 9 && gxp < W && gyp < H) v = dL_dpixels[(size_t)(j - 6) * HW + (size_t)gyp * W + gxp];
        Bop[t] = v;
    }
    // A operand of K-step t: component (i >> 3) of element [slot i & 7][pixel 4 t + kq]
    const float *arow = reinterpret_cast<const float *>(&smat[wave][(j & 7) * MSTR + kq]) + (j >> 3);
    const bool my_cols = kq < 2 ? j < 6 : (j >= 6 && j < 9);  // the columns that mean something in this lane's rows
    const float4 *wslab = &slab[wave][0][0];

    float T = T_final;
    float rho = 0.f;  // (colour accumulated BEHIND the current position) . g

    // software prefetch: the list entries of a chunk are loaded one chunk ahead, their indices two chunks ahead (three
    // dependent gathers otherwise sit in front of every chunk, and only 4 waves per SIMD are there to hide them)
    const int c0 = ((bmax - 1) / 64) * 64;
    uint32_t id_cur = (c0 + lane < wmax) ? point_list[range.x + c0 + lane] : 0u;
    uint32_t id_nxt = (c0 >= 64 && c0 - 64 + lane < wmax) ? point_list[range.x + c0 - 64 + lane] : 0u;
    RawEntry raw = load_raw(c0 + lane < wmax, id_cur, means2D, conic_opacity, rgb);

    for (int c = c0; c >= 0; c -= 64) {
        float *acc_tab = sacc[wave];
        // issue the loads of the NEXT chunk now; they are consumed at the top of the next iteration
        const bool have_nxt = c >= 64 && c - 64 + lane < wmax;
        const RawEntry raw_nxt = load_raw(have_nxt, id_nxt, means2D, conic_opacity, rgb);
        const uint32_t id_nxt2 = (c >= 128 && c - 128 + lane < wmax) ? point_list[range.x + c - 128 + lane] : 0u;
        if (c < wmax) {  // wave-uniform
            const bool have = c + lane < wmax;
            Entry e;
            e.x = raw.xy.x;
            e.y = raw.xy.y;
            e.a2 = -0.5f * LOG2E * raw.co.x;
            e.b2 = -LOG2E * raw.co.y;
            e.c2 = -0.5f * LOG2E * raw.co.z;
            e.o = entry_opacity_term(raw.co.w);
            e.relevant = have && gsr_can_touch_box(raw.xy, raw.co, (float)qx0, (float)qy0, (float)qx0 + 7.0f, (float)qy0 + 7.0f);
            unsigned long long m = __ballot(e.relevant);
            GSR_STAT(0, __popcll(__ballot(have)));
            GSR_STAT(1, __popcll(m));
            GSR_STAT(6, 1);
            // wave-private LDS slab + broadcast reads (see K8): entry k is read back by all lanes with a uniform address
            slab[wave][lane][0] = make_float4(e.x, e.y, e.a2, e.b2);
            slab[wave][lane][1] = make_float4(e.c2, e.o, raw.r, raw.g);
            slab[wave][lane][2] = make_float4(raw.b, __uint_as_float(id_cur), raw.co.w, 0.f);
            // chunk index of the entry in each slot: 8 bytes in a scalar register pair (wave-uniform, SALU only)
            unsigned long long sk = 0ull;
            // phase B: [rows x pixels] . [pixels x columns] on the matrix pipe.  Result element r of a lane: row
            // 4 kq + r, column j.  Rows of unfilled slots hold whatever the matrix held before: rows are independent,
            // they are simply not added.
            auto contract_batch = [&](int filled) {
                __builtin_amdgcn_wave_barrier();  // the wave's own LDS stores above precede its loads below
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains: back-to-back issue
#pragma unroll
                for (int t = 0; t < 16; t += 2) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[8 * t], Bop[t], acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[8 * t + 8], Bop[t + 1], acc1, 0, 0, 0);
                }
                acc += acc1;
                const uint32_t sk4 = (kq & 1) ? (uint32_t)(sk >> 32) : (uint32_t)sk;  // this lane's four slots
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const int slot = 4 * (kq & 1) + r4;
                    // an entry is in exactly one batch of this wave: a plain store into the wave's own table
                    if (slot < filled && my_cols) acc_tab[(int)((sk4 >> (8 * r4)) & 0xffu) * 9 + j] = acc[r4];
                }
                GSR_STAT(5, 1);
                __builtin_amdgcn_wave_barrier();  // the next batch overwrites the matrix only after these reads
                sk = 0ull;
            };
            int s = 0;  // filled slots of the current batch (wave-uniform)
            // Entries are taken two at a time: their LDS reads and alphas are independent (one wait for both), only
            // the short T / rho chain is sequential.  MB is even, so a pair never straddles a batch.
            while (m) {
                const int ka = 63 - __builtin_clzll(m);  // back to front
                m &= ~(1ull << ka);
                const bool two = m != 0ull;              // wave-uniform
                const int kb = two ? 63 - __builtin_clzll(m) : ka;
                m &= ~(1ull << kb);
                const float4 *ea = wslab + 3 * ka, *eb = wslab + 3 * kb;  // wave-uniform addresses: LDS broadcasts
                const float4 a0 = ea[0], a1 = ea[1], b0 = eb[0], b1 = eb[1];
                const float acb = ea[2].x, bcb = eb[2].x;
                const float adx = a0.x - pxf, ady = a0.y - pyf, bdx = b0.x - pxf, bdy = b0.y - pyf;
                const float ape = entry_exponent(a0, a1, adx, ady), bpe = entry_exponent(b0, b1, bdx, bdy);
                const float aar = entry_alpha_raw(a1, ape), bar = entry_alpha_raw(b1, bpe);   // o * exp(power), unclamped
                const float aal = fminf(0.99f, aar), bal = fminf(0.99f, bar);
                const bool atake = (c + ka + 1 <= last) && entry_power_ok(a1, ape) && aal >= ALPHA_MIN;
                const bool btake = two && (c + kb + 1 <= last) && entry_power_ok(b1, bpe) && bal >= ALPHA_MIN;
                const float acg = fmaf(acb, g2, fmaf(a1.w, g1, a1.z * g0)), bcg = fmaf(bcb, g2, fmaf(b1.w, g1, b1.z * g0));
                GSR_STAT(2, two ? 2 : 1);
                GSR_STAT(3, (__builtin_amdgcn_ballot_w64(atake) != 0ull) + (__builtin_amdgcn_ballot_w64(btake) != 0ull));
                GSR_STAT(4, __popcll(__builtin_amdgcn_ballot_w64(atake)) + __popcll(__builtin_amdgcn_ballot_w64(btake)));
                // No "does any lane take it?" branch: the quadrant test above is exact, 97.5-99.7 % of the evaluated
                // entries are taken by at least one pixel (profiles/r02_kstats_bwd.txt), and the wave-wide test costs
                // two VALU instructions per entry.  A lane that does not take an entry runs the same arithmetic with
                // alpha = 0: T / (1 - 0) and rho + 0 * (...) leave its state untouched and it stores (0, 0) (the
                // factors are SELECTED, never multiplied by an inf / NaN).
                {
                    const float ae = atake ? aal : 0.f, are = atake ? aar : 0.f;
                    const float inv_1ma = __builtin_amdgcn_rcpf(1.f - ae);  // 1 ulp, inside the 1e-4 budget
                    const float Tn = T * inv_1ma;                           // transmittance in front of the entry
                    const float dot = acg - rho;                            // (c - R) . g
                    const float da = fmaf(dot, Tn, -(tb * inv_1ma));        // dL/dalpha
                    smat[wave][s * MSTR + lane] = make_float2(are * da, ae * Tn);   // (q = o G dL/dalpha, w)
                    rho = fmaf(ae, dot, rho);
                    T = Tn;
                }
                sk |= (unsigned long long)ka << (8 * s);
                s++;
                if (two) {
                    const float ae = btake ? bal : 0.f, are = btake ? bar : 0.f;
                    const float inv_1ma = __builtin_amdgcn_rcpf(1.f - ae);
                    const float Tn = T * inv_1ma;
                    const float dot = bcg - rho;
                    const float da = fmaf(dot, Tn, -(tb * inv_1ma));
                    smat[wave][s * MSTR + lane] = make_float2(are * da, ae * Tn);
                    rho = fmaf(ae, dot, rho);
                    T = Tn;
                    sk |= (unsigned long long)kb << (8 * s);
                    s++;
                }
                if (s == MB) {
                    contract_batch(MB);
                    s = 0;
                }
            }
            if (s > 0) contract_batch(s);
        }
        __syncthreads();
        // flush, (entry, value)-major: 576 (entry, column) outputs per chunk over 256 threads; 9 ADJACENT lanes add the 9
        // values of one (tile, entry) pair into that Gaussian's 36-byte gradient record, so a wave instruction touches
        // ~8 records instead of 64 scattered words.  Everything the maps need is in the slab of the longest-walking wave
        // (no global loads here).  The four quadrant waves' raw pixel moments (about the TILE origin) add up; then
        //   M0 = sum q, Mx = sum q dx = ex M0 - Ax, Mxx = sum q dx^2 = ex^2 M0 - 2 ex Ax + Axx, ... (ex = x_e - tile x0)
        //   dL/dmean = -(A Mx + B My, B Mx + C My) * (W/2, H/2);  dL/d(A,B,C) = -(Mxx / 2, Mxy, Myy / 2);
        //   dL/dopacity = sum G dL/dalpha = M0 / o  (q carries the factor o);  dL/drgb = sum w g.
        const float tx0 = (float)(tx * GSR_BLOCK_X), ty0 = (float)(ty * GSR_BLOCK_Y);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int idx = threadIdx.x + 256 * r;
            const int e = idx / 9, col = idx - 9 * e;
            if (idx < 576 && c + e < bmax) {
                const int b9 = 9 * e;
                auto S = [&](int v) { return (sacc[0][b9 + v] + sacc[1][b9 + v]) + (sacc[2][b9 + v] + sacc[3][b9 + v]); };
                const float4 *se = &slab[wbest][e][0];
                float val;
                if (col >= 2 && col < 5) {
                    val = S(6 + (col - 2));
                } else {
                    const float A0 = S(0), Ax = S(1), Ay = S(2);
                    const float4 e0 = se[0];
                    const float ex = e0.x - tx0, ey = e0.y - ty0;
                    // conic back from the log2-scaled copy: A = a2 / (-log2(e) / 2), B = b2 / (-log2 e), C likewise
                    const float cA = e0.z * (-2.0f / LOG2E), cB = e0.w * (-1.0f / LOG2E), cC = se[1].x * (-2.0f / LOG2E);
                    const float Mx = fmaf(ex, A0, -Ax), My = fmaf(ey, A0, -Ay);
                    if (col == 0) val = -(cA * Mx + cB * My) * ddelx_dx;
                    else if (col == 1) val = -(cB * Mx + cC * My) * ddely_dy;
                    else if (col == 5) val = -0.5f * (fmaf(fmaf(ex, A0, -2.f * Ax), ex, S(3)));
                    else if (col == 6) val = -(fmaf(ex, My, -ey * Ax) + S(4));   // ex ey M0 - ex Ay - ey Ax + Axy
                    else if (col == 7) val = -0.5f * (fmaf(fmaf(ey, A0, -2.f * Ay), ey, S(5)));
                    else val = A0 == 0.f ? 0.f : A0 / se[2].z;
                }
#ifndef GSR_ABL_NOATOMIC
                if (val != 0.f) atomicAdd(dL_record + 9 * (size_t)__float_as_uint(se[2].y) + col, val);
#else
                asm volatile("" ::"v"(val));
#endif
            }
        }
        __syncthreads();
        // every wave clears ITS OWN table behind the reads above (LDS operations of one wave execute in order, so the
        // next chunk's stores of this wave follow the clears)
#pragma unroll
        for (int i = 0; i < 9; i++) acc_tab[64 * i + lane] = 0.f;
        raw = raw_nxt;
        id_cur = id_nxt;
        id_nxt = id_nxt2;
    }
}

}  // namespace

#ifdef GSR_STATS
extern "C" int gsr_debug_stats(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

int gsr_launch_composite_forward(int P, int W, int H, const int32_t *ranges, const uint32_t *point_list,
                                 const float *means2D, const float *conic_opacity, const float *rgb,
                                 const uint8_t *compute_locally, const float *bg, float *out_color, float *final_T,
                                 int32_t *n_contrib, hipStream_t stream) {
    (void)P;
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    hipLaunchKernelGGL(composite_forward_kernel, dim3(gx * gy), dim3(256), 0, stream, W, H, gx,
                       reinterpret_cast<const int2 *>(ranges), point_list, reinterpret_cast<const float2 *>(means2D),
                       reinterpret_cast<const float4 *>(conic_opacity), rgb, compute_locally, bg, out_color, final_T,
                       n_contrib);
    GSR_LAUNCH_CHECK();
    return 0;
}

int gsr_launch_composite_backward(int P, int W, int H, const int32_t *ranges, const uint32_t *point_list,
                                  const float *means2D, const float *conic_opacity, const float *rgb,
                                  const uint8_t *compute_locally, const float *bg, const float *final_T,
                                  const int32_t *n_contrib, const float *dL_dpixels, float *dL_record,
                                  hipStream_t stream) {
    GSR_HIP(hipMemsetAsync(dL_record, 0, sizeof(float) * 9 * (size_t)P, stream));
    if (P == 0) return 0;
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    hipLaunchKernelGGL(composite_backward_kernel, dim3(gx * gy), dim3(256), 0, stream, W, H, gx,
                       reinterpret_cast<const int2 *>(ranges), point_list, reinterpret_cast<const float2 *>(means2D),
                       reinterpret_cast<const float4 *>(conic_opacity), rgb, compute_locally, bg, final_T, n_contrib,
                       dL_dpixels, dL_record);
    GSR_LAUNCH_CHECK();
    return 0;
}
