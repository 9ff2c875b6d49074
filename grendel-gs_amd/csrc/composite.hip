// composite.hip -- K8 (alpha-composite forward) and K10 (its backward) for gfx950 / wave64.
//
// Unit of work: ONE WAVE = one 8x8 pixel quadrant of a 16x16 tile (lane = pixel); a 256-thread
// workgroup is the four quadrants of a tile, but the waves never synchronise with each other:
//   * each wave walks the tile's depth-sorted list in chunks of 64 entries, one entry per lane
//     (coalesced index read + one 36-byte gather per lane);
//   * each lane tests ITS entry against the wave's quadrant (bounding box of the alpha >= 1/255
//     ellipse) and a 64-bit ballot gives the entries that can touch the quadrant at all;
//   * the wave then iterates over the set bits only, broadcasting the entry lane -> SGPRs with
//     v_readlane (no LDS traffic, no barriers) and blending per pixel;
//   * early termination is per wave: __all(done) leaves the loop.
// Entries skipped by the quadrant test would have been rejected per pixel by the alpha < 1/255 rule,
// so the image and n_contrib are those of the plain algorithm.
//
// Backward: same walk in reverse order; the 9 per-pixel partial gradients are reduced across the
// wave with DPP adds, accumulated in the registers of the lane that holds the entry, and flushed
// with one vector atomic per value per 64-entry chunk.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_STOP = 0.0001f;

// Workgroup b is dispatched to XCD b % 8 (each XCD has its own 4 MiB L2).  Neighbouring tiles share most
// of their Gaussians, so hand every XCD a CONTIGUOUS span of tile ids instead of every 8th tile: the
// gathers of one span then hit one L2.  Bijective for any tile count (cdna_hip_programming.md §5).
__device__ __forceinline__ int xcd_tile_of_block(int b, int nwg) {
#ifdef GSR_NO_XCD_MAP
    return b;
#else
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
#endif
}

#ifdef GSR_STATS
__device__ unsigned long long g_stats[8];
#define GSR_STAT(i, v) do { const unsigned long long sv__ = (unsigned long long)(v); if (lane == 0) atomicAdd(&g_stats[i], sv__); } while (0)
#else
#define GSR_STAT(i, v) do { } while (0)
#endif

__device__ __forceinline__ float bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// lane-resident entry of the tile list
struct Entry {
    float x, y;        // pixel centre
    float a2, b2, c2;  // conic pre-scaled to log2 units: p2 = a2 dx^2 + b2 dx dy + c2 dy^2
    float o;           // opacity
    bool relevant;
};

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
        e.o = co.w;
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
    const int tile = xcd_tile_of_block(blockIdx.x, gridDim.x);
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
                const float p2 = (q0.z * dx + q0.w * dy) * dx + q1.x * dy * dy;
                al[u] = fminf(0.99f, q1.y * __builtin_amdgcn_exp2f(p2));
                ok[u] = live && p2 <= 0.f && al[u] >= ALPHA_MIN;
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
// sum over the 64 lanes, result valid in lane 63 (GFX9 DPP reduction ladder)
__device__ __forceinline__ float wave_sum_to_63(float v) {
#define GSR_DPP(x, ctrl, rmask) \
    __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, false))
    v += GSR_DPP(v, 0x111, 0xf);  // row_shr:1
    v += GSR_DPP(v, 0x112, 0xf);  // row_shr:2
    v += GSR_DPP(v, 0x114, 0xf);  // row_shr:4
    v += GSR_DPP(v, 0x118, 0xf);  // row_shr:8   -> lane 15 of each row holds the row sum
    v += GSR_DPP(v, 0x142, 0xa);  // row_bcast:15 -> rows 1,3
    v += GSR_DPP(v, 0x143, 0xc);  // row_bcast:31 -> rows 2,3
#undef GSR_DPP
    return v;
}

// Transposed wave reduction: every lane brings 64 values x[0..63]; on return lane L holds
// sum over all 64 lanes of x[L].  Butterfly over lane-index bits 5..0; at each stage a lane keeps the
// half of the values whose index bit equals its lane bit and hands the other half to its partner:
// v_permlane32_swap / v_permlane16_swap (gfx950) for distances 32 / 16, DPP row rotations and quad
// permutes below.  ~150 VALU instructions for 64 sums instead of 64 x (6 DPP adds + readlane).
#define GSR_DPPF(x, ctrl, rmask, bmask, oldv) \
    __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(oldv), __float_as_int(x), ctrl, rmask, bmask, false))
__device__ __forceinline__ float transpose_reduce64(float (&x)[64], int lane) {
    float y[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[i]), __float_as_uint(x[i + 32]), false, false);
        y[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    float z[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(y[i]), __float_as_uint(y[i + 16]), false, false);
        z[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    float w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float send = b3 ? z[i] : z[i + 8], keep = b3 ? z[i + 8] : z[i];
        w[i] = keep + GSR_DPPF(send, 0x128, 0xf, 0xf, 0.f);  // row_ror:8 == lane ^ 8 within the row
    }
    float u[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float send = b2 ? w[i] : w[i + 4], keep = b2 ? w[i + 4] : w[i];
        float recv = GSR_DPPF(send, 0x12C, 0xf, 0x5, 0.f);   // row_ror:12: lanes of banks 0,2 read lane+4
        recv = GSR_DPPF(send, 0x124, 0xf, 0xa, recv);        // row_ror:4 : lanes of banks 1,3 read lane-4
        u[i] = keep + recv;
    }
    float t[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float send = b1 ? u[i] : u[i + 2], keep = b1 ? u[i + 2] : u[i];
        t[i] = keep + GSR_DPPF(send, 0x4E, 0xf, 0xf, 0.f);   // quad_perm [2,3,0,1]
    }
    const float send = b0 ? t[0] : t[1], keep = b0 ? t[1] : t[0];
    return keep + GSR_DPPF(send, 0xB1, 0xf, 0xf, 0.f);       // quad_perm [1,0,3,2]
}

__global__ void __launch_bounds__(256)
composite_backward_kernel(int W, int H, int gx, const int2 *__restrict__ ranges,
                          const uint32_t *__restrict__ point_list, const float2 *__restrict__ means2D,
                          const float4 *__restrict__ conic_opacity, const float *__restrict__ rgb,
                          const uint8_t *__restrict__ compute_locally, const float *__restrict__ bg,
                          const float *__restrict__ final_T, const int32_t *__restrict__ n_contrib,
                          const float *__restrict__ dL_dpixels, float *__restrict__ dL_record) {
    const int tile = xcd_tile_of_block(blockIdx.x, gridDim.x);
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
    const float bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // entries beyond the furthest contributor of any pixel of this quadrant are dead for the wave;
    // the four waves of the tile walk the SAME chunk sequence so that their per-entry sums can be
    // combined in LDS and leave the workgroup as ONE set of atomics per (tile, entry)
    __shared__ float sacc[4][64 * 9];  // [wave][entry * 9 + value]: conflict-free for (entry, value)-major lanes
    __shared__ float4 slab[4][64][3];
    __shared__ int s_wmax[4];
    int wmax = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    const int bmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    if (bmax == 0) return;

    float T = T_final;
    float R0 = 0.f, R1 = 0.f, R2 = 0.f;  // colour accumulated BEHIND the current position
    const float tb = T_final * bg_dot;

    constexpr int EB = 7;  // entries per reduction batch: 7 x 9 = 63 of the 64 butterfly slots
    const int slot_j = lane / 9, slot_v = lane - 9 * (lane / 9);

    for (int c = ((bmax - 1) / 64) * 64; c >= 0; c -= 64) {
        // per-(entry, value) sums of this wave for this chunk live in its own LDS region
#pragma unroll
        for (int v = 0; v < 9; v++) sacc[wave][v * 64 + lane] = 0.f;
        if (c < wmax) {  // wave-uniform
            const bool have = c + lane < wmax;
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
            GSR_STAT(6, 1);
            // wave-private LDS slab + broadcast reads instead of v_readlane (see K8): the VALU bounds this kernel
            slab[wave][lane][0] = make_float4(e.x, e.y, e.a2, e.b2);
            slab[wave][lane][1] = make_float4(e.c2, e.o, r, g);
            slab[wave][lane][2] = make_float4(b, 0.f, 0.f, 0.f);
            while (m) {
                float xs[64];
                int ks[EB];
#pragma unroll
                for (int j = 0; j < EB; j++) {
                    // next entry (back to front) that at least one pixel of the quadrant takes
                    bool got = false, take = false;
                    int k = 0;
                    float dx = 0.f, dy = 0.f, o = 0.f, G = 0.f, alpha = 0.f, a2 = 0.f, b2 = 0.f, c2 = 0.f;
                    float cr = 0.f, cg = 0.f;
                    while (m) {
                        k = 63 - __builtin_clzll(m);
                        m &= ~(1ull << k);
                        const float4 q0 = slab[wave][k][0], q1 = slab[wave][k][1];
                        a2 = q0.z;
                        b2 = q0.w;
                        c2 = q1.x;
                        o = q1.y;
                        cr = q1.z;
                        cg = q1.w;
                        dx = q0.x - pxf;
                        dy = q0.y - pyf;
                        const float p2 = (a2 * dx + b2 * dy) * dx + c2 * dy * dy;
                        G = __builtin_amdgcn_exp2f(p2);
                        alpha = fminf(0.99f, o * G);
                        take = (c + k + 1 <= last) && p2 <= 0.f && alpha >= ALPHA_MIN;
                        GSR_STAT(2, 1);
                        if (__any(take)) {
                            got = true;
                            break;
                        }
                    }
                    ks[j] = k;
                    // branch-free contribution of this entry for this pixel.  A lane that does not `take` runs
                    // the same arithmetic with alpha = 0 and G = 0: T / (1 - 0) and R + 0 * (c - R) leave its state
                    // untouched, so only two selects are needed (never a multiply by an inf/NaN: G is SELECTED).
                    // R = colour accumulated behind the current position, updated eagerly:
                    // R <- alpha c + (1 - alpha) R is the reference's lazily evaluated accum_rec recurrence.
                    if (got) {
                        GSR_STAT(3, 1);
                        GSR_STAT(4, __popcll(__ballot(take)));
                    }
                    if (!got) {  // wave-uniform: the chunk ran out of entries inside this batch
#pragma unroll
                        for (int v = 0; v < 9; v++) xs[j * 9 + v] = 0.f;
                        continue;
                    }
                    const float cb = slab[wave][k][2].x;
                    const float ae = take ? alpha : 0.f;
                    const float Ge = take ? G : 0.f;
                    const float inv_1ma = __builtin_amdgcn_rcpf(1.f - ae);  // 1 ulp, inside the 1e-4 budget
                    const float Tn = T * inv_1ma;                           // transmittance in front of the entry
                    const float d0 = cr - R0, d1 = cg - R1, d2 = cb - R2;
                    const float da = (d0 * g0 + d1 * g1 + d2 * g2) * Tn - tb * inv_1ma;  // dL/dalpha
                    const float q = o * da * Ge;  // G * dL/dG, min(0.99,.) treated as identity
                    const float qdx = q * dx, qdy = q * dy;
                    const float wm = ae * Tn;
                    // raw moment sums; the per-entry linear maps to dL/dmean2D and dL/dconic are applied ONCE per
                    // (tile, entry) in the flush below instead of once per pixel here
                    xs[j * 9 + 0] = qdx;
                    xs[j * 9 + 1] = qdy;
                    xs[j * 9 + 2] = qdx * dx;
                    xs[j * 9 + 3] = qdx * dy;
                    xs[j * 9 + 4] = qdy * dy;
                    xs[j * 9 + 5] = Ge * da;
                    xs[j * 9 + 6] = wm * g0;
                    xs[j * 9 + 7] = wm * g1;
                    xs[j * 9 + 8] = wm * g2;
                    T = Tn;
                    R0 += ae * d0;
                    R1 += ae * d1;
                    R2 += ae * d2;
                }
                xs[63] = 0.f;
                GSR_STAT(5, 1);
                const float total = transpose_reduce64(xs, lane);  // lane L: sum over pixels of value L
                int kk = ks[0];
#pragma unroll
                for (int j = 1; j < EB; j++) kk = slot_j == j ? ks[j] : kk;
                if (lane < EB * 9 && total != 0.f) sacc[wave][kk * 9 + slot_v] += total;
            }
        }
        __syncthreads();
        // flush, (entry, value)-major: 576 (entry, value) sums per chunk over 256 threads; 9 ADJACENT lanes add the
        // 9 sums of one (tile, entry) pair into that Gaussian's 36-byte gradient record, so a wave instruction
        // touches ~8 records instead of 64 scattered words (the memory-side atomic requests were the cost:
        // 0.54 -> 0.39 ms on views with large splats).  The moment sums are mapped to gradients here, once per
        // (tile, entry):  S1 = sum q dx, S2 = sum q dy:  dL/dmean = -(A S1 + B S2, B S1 + C S2) * (W/2, H/2);
        // dL/d(A,B,C) = -(1/2 sum q dx^2, sum q dx dy, 1/2 sum q dy^2).
        // record columns: means2D 0:2, rgb 2:5, conic_opacity 5:9 (the exchange's record order).
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int idx = threadIdx.x + 256 * r;
            const int e = idx / 9, v = idx - 9 * e;
            if (idx < 576 && c + e < bmax) {
                const uint32_t id = point_list[range.x + c + e];
                float val = sacc[0][idx] + sacc[1][idx] + sacc[2][idx] + sacc[3][idx];
                if (v < 2) {
                    const int u = idx + 1 - 2 * v;
                    const float other = sacc[0][u] + sacc[1][u] + sacc[2][u] + sacc[3][u];
                    if (val != 0.f || other != 0.f) {
                        const float4 co = conic_opacity[id];
                        val = v == 0 ? -(co.x * val + co.y * other) * ddelx_dx : -(co.z * val + co.y * other) * ddely_dy;
                    }
                } else if (v == 2 || v == 4) {
                    val *= -0.5f;
                } else if (v == 3) {
                    val = -val;
                }
                const int col = v < 2 ? v : (v < 6 ? v + 3 : v - 4);  // moments 2..5 -> conic_opacity, 6..8 -> rgb
#ifndef GSR_ABL_NOATOMIC
                if (val != 0.f) atomicAdd(dL_record + 9 * (size_t)id + col, val);
#else
                asm volatile("" ::"v"(val), "v"(dL_record + 9 * (size_t)id + col));
#endif
            }
        }
        __syncthreads();
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
