// composite.hip -- K8 (alpha-composite forward) and K10 (its backward) for gfx950 / wave64.
//
// Unit of work: ONE WAVE = one 8x8 pixel quadrant of a 16x16 tile (lane = pixel); a 256-thread
// workgroup is the four quadrants of a tile.  In the forward the waves never synchronise with each other:
//   * each wave walks the tile's depth-sorted list in chunks of 64 entries, one entry per lane
//     (coalesced index read + one 36-byte gather per lane);
//   * each lane tests ITS entry against the wave's quadrant (exact: minimum of the conic's quadratic form over
//     the box vs the alpha >= 1/255 level) and a 64-bit ballot gives the entries that can touch the quadrant;
//   * the relevant entries of the chunk are compacted (in list order) into PAIR RECORDS in a wave-private LDS slab and
//     read back with wave-uniform addresses: an LDS broadcast costs no VALU issue slot (9 v_readlane per entry did),
//     one read returns the values of two entries in adjacent registers, and what is independent between entries
//     (offsets, the exponent's quadratic form, colour accumulation) runs as packed fp32 for two entries per instruction;
//     VALU issue is what bounds the forward (profiles/r02_pmc.txt: SQ_ACTIVE_INST_VALU covers 0.85 of the SIMD cycles);
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

// list entries WALKED per launch group (always on; one atomic per workgroup): [0] K8, [1] K10 -- per tile, the entries
// of the tile's list that its longest-walking quadrant wave went through (rounded up to the 64-entry chunks the walk
// loads).  bench.py turns them into the bytes the kernels can have asked HBM for: early termination leaves most of
// every list untouched, so the formula that credits the WHOLE list (40 / 76 B x D) is not a roofline.
__device__ unsigned long long g_walked[2];

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


// ---- packed pairs.  A wave instruction on this chip retires in four cycles whether it is v_fma_f32 or v_pk_fma_f32,
// so the arithmetic that is independent between two list entries (offsets, the exponent's quadratic form, the colour
// terms) is done for TWO entries per instruction.  For that the RELEVANT entries of a chunk are compacted (in walk
// order) into pair records in the wave's LDS slab, the two entries' values of each field next to each other, so
// that one broadcast read returns register PAIRS: rec = {xa xb ya yb | a2a a2b b2a b2b | c2a c2b oa ob | ra rb ga gb |
// ba bb - -}.  Slots behind the last entry are zeroed (K8) or never evaluated (K10).
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int PREC = 20;  // floats per pair record (80 bytes: 16-byte aligned reads)
struct PairRec {
    v2f x, y, a2, b2, c2, o, r, g, b;
};
__device__ __forceinline__ void pair_store(float *__restrict__ slab, int pos, const Entry &e, float r, float g, float b) {
    float *rec = slab + (pos >> 1) * PREC + (pos & 1);
    rec[0] = e.x;  rec[2] = e.y;  rec[4] = e.a2;  rec[6] = e.b2;  rec[8] = e.c2;
    rec[10] = e.o; rec[12] = r;   rec[14] = g;    rec[16] = b;
}
__device__ __forceinline__ PairRec pair_load(const float *__restrict__ slab, int pair) {  // wave-uniform address
    const float4 *q = reinterpret_cast<const float4 *>(slab + pair * PREC);
    const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const float2 q4 = *reinterpret_cast<const float2 *>(q + 4);
    PairRec p;
    p.x = v2f{q0.x, q0.y};  p.y = v2f{q0.z, q0.w};  p.a2 = v2f{q1.x, q1.y};  p.b2 = v2f{q1.z, q1.w};
    p.c2 = v2f{q2.x, q2.y}; p.o = v2f{q2.z, q2.w};  p.r = v2f{q3.x, q3.y};   p.g = v2f{q3.z, q3.w};
    p.b = v2f{q4.x, q4.y};
    return p;
}
// exponents of both entries of a pair at this lane's pixel: a2 dx^2 + b2 dx dy + c2 dy^2 + log2 o  (7 packed instructions)
__device__ __forceinline__ v2f pair_exponent(const PairRec &p, float pxf, float pyf) {
    const v2f dx = p.x - pxf, dy = p.y - pyf;
    return __builtin_elementwise_fma(__builtin_elementwise_fma(p.a2, dx, p.b2 * dy), dx,
                                     __builtin_elementwise_fma(p.c2 * dy, dy, p.o));
}

// block -> tile: XCD-contiguous spans of the BAND this rank renders.  The binning leaves the band (the row hull of the
// mask, [lo, hi) tile rows) in row `tiles` of the range table (include/gsraster.h: gsr_bin_sort); anything implausible
// there falls back to spans of the whole grid.
__device__ __forceinline__ int composite_tile_of_block(const int2 *__restrict__ ranges, int gx, int tiles) {
    const int2 hull = ranges[tiles];
    const int gy = tiles / gx;
    if (hull.x < 0 || hull.y > gy || hull.x >= hull.y) return gsr_xcd_span_of_block(blockIdx.x, tiles);
    return gsr_xcd_span_of_block_band(blockIdx.x, tiles, hull.x * gx, (hull.y - hull.x) * gx);
}

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

// ---- list SEGMENTS for the backward (round 4).  One workgroup per tile walks the tile's list serially, so the kernel
// lasts at least as long as its longest list takes ONE wave -- on a thin row band (one round of resident workgroups,
// world size 8) that IS the kernel time (measured: K8 / K10 of a 1/8 band take 1/2.8 of the full image's time, and
// neither a software prefetch nor another block -> tile map changes it, profiles/r04_ab_*).  The backward can be cut
// exactly: K10 needs, per pixel, the transmittance T and the colour accumulated IN FRONT of a list position, both of
// which the forward knows when it passes that position.  So K8 leaves a CHECKPOINT (T, C.rgb per pixel) every SEG_LEN
// entries it really walks and queues the segment that starts there; K10 runs segment 0 of every tile in its usual
// workgroups and hands the queued segments to persistent worker workgroups (atomic ticket).  Work is only ever created
// for entries the forward walked: early termination is untouched, nothing is recomputed.
#ifndef GSR_SEG_LEN
#define GSR_SEG_LEN 256
#endif
#ifndef GSR_SEG_MAXJ
#define GSR_SEG_MAXJ 16
#endif
constexpr int SEG_LEN = GSR_SEG_LEN;    // list entries per segment (a multiple of the 64-entry chunk)
constexpr int SEG_MAXJ = GSR_SEG_MAXJ;  // boundaries per tile that can carry a checkpoint (<= 31; a list walked deeper keeps a long tail)
static_assert(SEG_LEN % 64 == 0 && (SEG_LEN & (SEG_LEN - 1)) == 0 && SEG_MAXJ <= 31,
              "segment geometry: K8 finds boundaries with c & (SEG_LEN - 1), so SEG_LEN is a power of two");
constexpr int SEG_WORKERS = 1024;
struct SegWs {
    uint32_t *hdr;      // [0] segments queued by K8 (may exceed cap: only the first cap exist), [1] K10's ticket
    uint32_t *queue;    // [cap]: tile * 32 + segment index (>= 1)
    int32_t *seg_slot;  // [tiles][SEG_MAXJ]: checkpoint slot of the boundary at entry (j + 1) * SEG_LEN, -1: none
    float *ckpt;        // [cap][4][256]: T, C.r, C.g, C.b of the tile's 256 pixels (quadrant-major: wave * 64 + lane)
    uint32_t cap;
};
__host__ __device__ inline size_t seg_ws_bytes(int tiles, uint32_t cap) {
    return 64 + sizeof(uint32_t) * (size_t)cap + sizeof(int32_t) * (size_t)tiles * SEG_MAXJ + sizeof(float) * 1024 * (size_t)cap;
}
inline SegWs seg_ws_of(void *ws, int tiles, uint32_t cap) {
    SegWs w{};
    if (!ws) return w;
    char *b = reinterpret_cast<char *>(ws);
    w.hdr = reinterpret_cast<uint32_t *>(b);
    w.queue = reinterpret_cast<uint32_t *>(b + 64);
    w.seg_slot = reinterpret_cast<int32_t *>(b + 64 + sizeof(uint32_t) * (size_t)cap);
    w.ckpt = reinterpret_cast<float *>(b + 64 + sizeof(uint32_t) * (size_t)cap + sizeof(int32_t) * (size_t)tiles * SEG_MAXJ);
    w.cap = cap;
    return w;
}
constexpr uint32_t SEG_CAP = 8192;

// A segment boundary the calling wave really crosses (every SEG_LEN walked entries): the first wave of the tile to get here claims a checkpoint slot and queues the
// segment for the backward; every wave that gets here leaves its pixels' state (T, colour in front of the boundary).
__device__ __forceinline__ void k8_checkpoint(const SegWs &seg, int *s_slot, int tile, int c, int lane,
                                                        int wave, float T, float c0, float c1, float c2) {
    const int j = c / SEG_LEN - 1;
    int slot = 0;
    if (lane == 0) {
        int v = atomicCAS(&s_slot[j], -1, -2);
        if (v == -1) {
            const uint32_t k = atomicAdd(seg.hdr, 1u);
            v = k < seg.cap ? (int)k : -3;
            if (v >= 0) {
                seg.queue[k] = (uint32_t)tile * 32u + (uint32_t)(j + 1);
                seg.seg_slot[(size_t)tile * SEG_MAXJ + j] = v;
            }
            __hip_atomic_store(&s_slot[j], v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            while (v == -2) v = __hip_atomic_load(&s_slot[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        slot = v;
    }
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot >= 0) {
        float *ck = seg.ckpt + (size_t)slot * 1024 + wave * 64 + lane;
        ck[0] = T;
        ck[256] = c0;
        ck[512] = c1;
        ck[768] = c2;
    }
}

// ------------------------------------------------------------------------------------------- K8
template <bool SEG, bool BAND>  // SEG: leave checkpoints for the backward's list segments; BAND: the grid covers the
                                // band's tiles only (each costs registers: the whole-image launch keeps its 64)
__global__ void __launch_bounds__(256)
composite_forward_kernel(int W, int H, int gx, const int2 *__restrict__ ranges,
                         const uint32_t *__restrict__ point_list, const float2 *__restrict__ means2D,
                         const float4 *__restrict__ conic_opacity, const float *__restrict__ rgb,
                         const uint8_t *__restrict__ compute_locally, const float *__restrict__ bg,
                         float *__restrict__ out_color, float *__restrict__ final_T, int32_t *__restrict__ n_contrib,
                         const SegWs seg, int band_first, int band_tiles, uint4 *__restrict__ zero16,
                         size_t zero16_n) {
    const size_t HW = (size_t)H * W;
    // Round 6: a buffer the BACKWARD needs zeroed (K10's [P,9] gradient record, 36 MB per 10^6 Gaussians) is cleared
    // here, by every workgroup's share of 16-byte stores before it starts on its tile: this kernel is bound by VALU issue
    // and leaves the memory pipes idle, the fill launch it replaces ran at the head of every backward.
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < zero16_n; i += (size_t)gridDim.x * 256)
        zero16[i] = make_uint4(0u, 0u, 0u, 0u);
    int tile;
    bool idle = false;
    if (BAND) {
        // The caller knows the rows of its band on the HOST (Grendel's strategies do): the grid covers the band's tiles
        // only.  With a grid over all tiles the 7/8 of the workgroups that are not ours on a 1/8 band still have to be
        // dispatched one by one behind the resident ones -- measured as a CONSTANT ~30 us (K8) / ~60 us (K10) of the
        // launch whatever the band (K10: 0.388 / 0.224 / 0.154 / 0.117 ms for 1 / 2 / 4 / 8 bands = 0.06 + 0.33 / W).
        // The pixels outside the band must still be exactly 0 (SUM assembly): the band's workgroups clear them together,
        // with coalesced stores, before they start on their tiles.
        // band_first < 0 (a launch captured in a hipGraph that has to serve every band, graphed_step.py): `band_tiles`
        // is a CAPACITY, the band itself is the row hull the binning left behind the range table; the workgroups above
        // the band's tile count only help to clear the pixels outside.
        if (band_first < 0) {
            const int gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
            int2 hull = ranges[gx * gy];
            if (hull.x < 0 || hull.y > gy || hull.x >= hull.y) hull = make_int2(0, 0);
            band_first = hull.x * gx;
            band_tiles = min((hull.y - hull.x) * gx, band_tiles);
            idle = (int)blockIdx.x >= band_tiles;
        }
        tile = band_first + (idle ? 0 : gsr_xcd_span_of_block(blockIdx.x, band_tiles));
    } else {
        tile = composite_tile_of_block(ranges, gx, (int)gridDim.x);
    }
    // (every workgroup clears its share of the pixels outside the band when it is done with its tile: workgroups with
    // short lists do it while the long ones still walk)
    auto clear_outside = [&]() {
        if (!BAND) return;
        const size_t n0 = (size_t)min(H, (band_first / gx) * GSR_BLOCK_Y) * W;
        const size_t p1 = (size_t)min(H, ((band_first + band_tiles) / gx) * GSR_BLOCK_Y) * W;
        const size_t nout = n0 + (HW - p1);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nout; i += (size_t)gridDim.x * 256) {
            const size_t q = i < n0 ? i : p1 + (i - n0);
            out_color[q] = 0.f;
            out_color[HW + q] = 0.f;
            out_color[2 * HW + q] = 0.f;
            final_T[q] = 1.f;
            n_contrib[q] = 0;
        }
    };
    if (idle) {
        clear_outside();
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx = tile % gx, ty = tile / gx;
    const int qx0 = tx * GSR_BLOCK_X + (wave & 1) * 8, qy0 = ty * GSR_BLOCK_Y + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t pid = (size_t)py * W + px;

    if (!compute_locally[tile]) {  // not ours: pixels must be exactly 0 (SUM all-reduce assembly)
        if (inside) {
            out_color[pid] = 0.f;
            out_color[HW + pid] = 0.f;
            out_color[2 * HW + pid] = 0.f;
            final_T[pid] = 1.f;
            n_contrib[pid] = 0;
        }
        clear_outside();
        return;
    }
    const int2 range = ranges[tile];
    const int n = range.y - range.x;
    const float pxf = (float)px, pyf = (float)py;

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    int last = 0;
    bool done = !inside;
    __shared__ __attribute__((aligned(16))) float slab[4][32 * PREC];
    __shared__ int s_walk[2];  // [0] longest walk of the four waves, [1] waves that have finished
    __shared__ int s_slot[SEG_MAXJ];  // checkpoint slot of boundary j: -1 nobody got there yet, -2 being claimed, -3 none
    if (threadIdx.x < 2) s_walk[threadIdx.x] = 0;
    if (SEG && threadIdx.x < SEG_MAXJ) {
        s_slot[threadIdx.x] = -1;
        if (seg.seg_slot) seg.seg_slot[(size_t)tile * SEG_MAXJ + threadIdx.x] = -1;
    }
    __syncthreads();  // the only workgroup barrier of the kernel: all four waves are at their first instructions
    v2f A0 = {0.f, 0.f}, A1 = {0.f, 0.f}, A2 = {0.f, 0.f};  // colour sums of the even / odd pair slots (added at the end)
    float *wslab = slab[wave];
    int walked = 0;

    // (measured, round 4: loading a chunk's entries one chunk ahead -- K10's software prefetch -- does NOT help here: 78
    // instead of 64 VGPRs, 0.146 -> 0.152 ms at 1 M Gaussians / 1080p and no change on a 1/8 row band; the walk of a
    // thin band is bound by the imbalance of ONE round of resident workgroups, not by memory latency)
    for (int c = 0; c < n; c += 64) {
        if (__all(done)) break;
        if (SEG && seg.seg_slot && c > 0 && (c & (SEG_LEN - 1)) == 0 && c <= SEG_LEN * SEG_MAXJ)
            k8_checkpoint(seg, s_slot, tile, c, lane, wave, T, A0.x + A0.y, A1.x + A1.y, A2.x + A2.y);
        walked = c + 64;
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
        // Stage the chunk's relevant entries, compacted in list order, as pair records in this wave's private LDS slab
        // and broadcast-read them back: an LDS read with a wave-uniform address returns the values to all 64 lanes
        // WITHOUT spending VALU issue slots, and the VALU is what bounds this kernel.  No barrier: the slab is private to
        // the wave and LDS operations of one wave execute in order.
        const int nrel = __popcll(m);
        if (e.relevant) pair_store(wslab, __popcll(m & ((1ull << lane) - 1ull)), e, r, g, b);
        {   // the walk below takes four slots at a time: zero the <= 3 slots behind the last entry (a stale colour times
            // a zero weight would still be a NaN if the stale bits are one)
            const int pad = (-nrel) & 3, slot = nrel + lane / 9;
            if (lane < 9 * pad) wslab[(slot >> 1) * PREC + 2 * (lane % 9) + (slot & 1)] = 0.f;
        }
        // Entries are taken four (two pairs) at a time: their alphas are independent (ILP across the LDS and v_exp
        // latency), only the short T chain is sequential, and the wave tests "everybody done?" once per group.
        for (int p0 = 0; p0 < nrel; p0 += 4) {  // wave-uniform
            float al[4];
            bool ok[4];
            int kk[4];
            v2f cr[2], cg[2], cb[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const PairRec pr = pair_load(wslab, (p0 >> 1) + h);
                const v2f pe = pair_exponent(pr, pxf, pyf);
                cr[h] = pr.r;
                cg[h] = pr.g;
                cb[h] = pr.b;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const bool live = m != 0;  // wave-uniform: entry p0 + 2 h + u exists
                    kk[2 * h + u] = live ? __builtin_ctzll(m) : 0;
                    m &= m - 1;  // 0 & ~0 stays 0
                    const float pe1 = u ? pe.y : pe.x, lo = u ? pr.o.y : pr.o.x;
                    al[2 * h + u] = fminf(0.99f, __builtin_amdgcn_exp2f(pe1));
                    ok[2 * h + u] = live && pe1 <= lo && al[2 * h + u] >= ALPHA_MIN;  // power <= 0  <=>  exponent <= log2 o
                }
            }
            float wv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool take = !done && ok[u];
                const float w0 = al[u] * T;
                const float test_T = T - w0;  // T (1 - alpha) >= 0: compared as an integer so that `!stop` below is the
                // complement of ONE compare (a float `<` and its negation are two instructions: NaN semantics)
                const bool stop = take && __float_as_int(test_T) < __float_as_int(T_STOP);
                done = done || stop;
                const bool blend = take && !stop;
                wv[u] = blend ? w0 : 0.f;
                T = blend ? test_T : T;
                last = blend ? c + kk[u] + 1 : last;
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const v2f w2 = {wv[2 * h], wv[2 * h + 1]};
                A0 = __builtin_elementwise_fma(cr[h], w2, A0);
                A1 = __builtin_elementwise_fma(cg[h], w2, A1);
                A2 = __builtin_elementwise_fma(cb[h], w2, A2);
            }
            if (__all(done)) break;
        }
    }
    if (lane == 0) {  // the last wave to get here adds the tile's longest walk to the launch counter
        atomicMax(&s_walk[0], walked < n ? walked : n);
        if (atomicAdd(&s_walk[1], 1) == 3) atomicAdd(&g_walked[0], (unsigned long long)s_walk[0]);
    }
    C0 = A0.x + A0.y;
    C1 = A1.x + A1.y;
    C2 = A2.x + A2.y;
    if (inside) {
        out_color[pid] = C0 + T * bg[0];
        out_color[HW + pid] = C1 + T * bg[1];
        out_color[2 * HW + pid] = C2 + T * bg[2];
        final_T[pid] = T;
        n_contrib[pid] = last;
    }
    clear_outside();
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
//   phase A (lane = pixel): walk the pair records back to front; recompute alpha, carry T and rho = R.g (R = colour
//       behind the entry: ONE scalar recurrence instead of three), and store (q, w) of the entry into a wave-private
//       LDS matrix [slot][pixel] -- ~20 VALU instructions per (wave, entry) where the transposed butterfly reduction
//       this replaces needed ~80 (SQ_INSTS_VALU 277 M -> 98 M per launch on the bench views);
//   phase B (every 8 entries): 16 K-steps of 4 pixels.  Lane (k = lane >> 4, i = lane & 15) reads component i >> 3 of
//       slot i & 7, pixel 4 t + k -- the A operand: rows 0-7 are the q-rows of the eight slots, rows 8-15 their w-rows;
//       conflict-free with the row stride of 66 -- and issues ONE v_mfma_f32_16x16x4_f32 against a per-lane constant
//       B operand (columns 0-5 the position polynomials, 6-8 the pixel's dL/dcolour).  Rows 0-7 x columns 0-5 of the
//       result are the q moments, rows 8-15 x columns 6-8 the colour gradients; they are STORED into the wave's own
//       per-chunk table (an entry is in exactly one batch of a wave; LDS float atomics measured ~50 cycles each);
//   flush (once per 64-entry chunk, between two workgroup barriers): ONE wave (they take turns), lane = entry, adds the
//       four quadrant waves' tables and maps moments -> gradients; then all waves, 9 adjacent lanes per (tile, entry)
//       pair, add the nine values into that Gaussian's 36-byte row of the [P,9] record (means2D 0:2, rgb 2:5,
//       conic_opacity 5:9); positions / conics / ids come from the LDS copy kept by the wave that walks the longest
//       list (no global loads in the flush).
// The list entries of a chunk are prefetched one chunk ahead (their indices two).  4 workgroups per CU (40.7 KB of LDS,
// 128 VGPRs).  Measured (profiles/r02_*): 0.496 -> 0.335 ms per launch at 1 M Gaussians / 1080p; by
// ablation the MFMA phase is 0.087 ms of it -- the fp32 MFMA runs at the VALU's own rate and does NOT overlap it
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0): what it buys is the cross-lane reduction, 16 VALU-equivalents per entry against ~52 --
// the record atomics 0.018 ms, the barriers 0.016 ms, the rest is the walk.
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MB = 8;      // entries per MFMA batch: rows 0-7 of the 16 x 16 result are their q-rows, rows 8-15 their w-rows
constexpr int MSTR = 66;   // row stride of the (q, w) matrix in 8-byte elements: phase A writes and phase B reads conflict-free

// segment `sidx` of tile `tile`: the list entries [sidx * SEG_LEN, next boundary with a checkpoint or the end)
__device__ __forceinline__ void
composite_backward_segment(const int tile, const int sidx, int W, int H, int gx, const int2 *__restrict__ ranges,
                           const uint32_t *__restrict__ point_list, const float2 *__restrict__ means2D,
                           const float4 *__restrict__ conic_opacity, const float *__restrict__ rgb,
                           const uint8_t *__restrict__ compute_locally, const float *__restrict__ bg,
                           const float *__restrict__ final_T, const int32_t *__restrict__ n_contrib,
                           const float *__restrict__ dL_dpixels, float *__restrict__ dL_record,
                           const float *__restrict__ out_color, const SegWs &seg) {
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

    // the segment's end: the next boundary that carries a checkpoint (then every pixel that blends beyond it starts
    // from the checkpointed state), or the end of the list
    const int c_lo = sidx * SEG_LEN;
    int end_slot = -1;
    if (seg.seg_slot && sidx < SEG_MAXJ) end_slot = seg.seg_slot[(size_t)tile * SEG_MAXJ + sidx];
    const int c_hi = end_slot >= 0 ? c_lo + SEG_LEN : 0x7fffffff;
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
    __shared__ __attribute__((aligned(16))) float slab[4][32 * PREC];  // [wave]: pair records of the relevant entries
    __shared__ float fslab[64 * 7];         // [entry of the chunk][x, y, a2, b2, c2, id, o]: what the flush needs
    __shared__ float sout[64 * 10];         // [entry][9 gradient values, id]: mapped by ONE wave, added by all (see the flush)
    __shared__ int s_wmax[4];
    int wmax = min(last, c_hi);  // entries of THIS segment a pixel of the wave still contributes to end here
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    if (wmax <= c_lo) wmax = 0;  // nothing of the wave in this segment
    if (lane == 0) s_wmax[wave] = wmax;
    for (int i = threadIdx.x; i < 4 * 64 * 9; i += 256) (&sacc[0][0])[i] = 0.f;
    __syncthreads();
    const int bmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    if (bmax == 0) return;
    if (threadIdx.x == 0) atomicAdd(&g_walked[1], (unsigned long long)(bmax - c_lo));  // entries this walk goes through
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
    float *wslab = slab[wave];

    float T = T_final;
    float rho = 0.f;  // (colour accumulated BEHIND the current position) . g
    if (end_slot >= 0 && last > c_hi) {
        // this pixel blends entries beyond the segment: start from the forward's state at the boundary.  T = the
        // transmittance in front of entry c_hi; the colour behind it is what the pixel ended with (without the
        // background's share) minus what it had accumulated in front of the boundary, normalised by T
        const float *ck = seg.ckpt + (size_t)end_slot * 1024 + wave * 64 + lane;
        T = ck[0];
        const float r0 = out_color[pid] - T_final * bg[0] - ck[256];
        const float r1 = out_color[HW + pid] - T_final * bg[1] - ck[512];
        const float r2 = out_color[2 * HW + pid] - T_final * bg[2] - ck[768];
        rho = (r0 * g0 + r1 * g1 + r2 * g2) * __builtin_amdgcn_rcpf(T);
    }

    // software prefetch: the list entries of a chunk are loaded one chunk ahead, their indices two chunks ahead (three
    // dependent gathers otherwise sit in front of every chunk, and only 4 waves per SIMD are there to hide them)
    const int c0 = ((bmax - 1) / 64) * 64;  // (>= c_lo: bmax > c_lo and c_lo is a multiple of 64)
    uint32_t id_cur = (c0 + lane < wmax) ? point_list[range.x + c0 + lane] : 0u;
    uint32_t id_nxt = (c0 >= c_lo + 64 && c0 - 64 + lane < wmax) ? point_list[range.x + c0 - 64 + lane] : 0u;
    RawEntry raw = load_raw(c0 + lane < wmax, id_cur, means2D, conic_opacity, rgb);

    for (int c = c0; c >= c_lo; c -= 64) {
        float *acc_tab = sacc[wave];
        // issue the loads of the NEXT chunk now; they are consumed at the top of the next iteration
        const bool have_nxt = c >= c_lo + 64 && c - 64 + lane < wmax;
        const RawEntry raw_nxt = load_raw(have_nxt, id_nxt, means2D, conic_opacity, rgb);
        const uint32_t id_nxt2 = (c >= c_lo + 128 && c - 128 + lane < wmax) ? point_list[range.x + c - 128 + lane] : 0u;
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
            // wave-private LDS slab + broadcast reads (see K8): the relevant entries, compacted in WALK order (back to
            // front: rank = relevant lanes above), as pair records -- slot 0 of a pair is walked first
            if (e.relevant) pair_store(wslab, __popcll((m >> lane) >> 1), e, raw.r, raw.g, raw.b);
            // the flush below reads positions / conics / ids by chunk index from the wave that walks the longest list
            // (it holds every entry of every chunk)
            if (wave == wbest) {
                float *fe = &fslab[7 * lane];
                fe[0] = e.x;  fe[1] = e.y;  fe[2] = e.a2;  fe[3] = e.b2;  fe[4] = e.c2;
                fe[5] = __uint_as_float(id_cur);
                fe[6] = raw.co.w;
            }
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
            for (int pi = 0; m; pi++) {
                const int ka = 63 - __builtin_clzll(m);  // back to front
                m &= ~(1ull << ka);
                const bool two = m != 0ull;              // wave-uniform
                const int kb = two ? 63 - __builtin_clzll(m) : ka;
                m &= ~(1ull << kb);
                const PairRec pr = pair_load(wslab, pi);  // wave-uniform address: LDS broadcasts
                const v2f pe = pair_exponent(pr, pxf, pyf);
                const float aar = __builtin_amdgcn_exp2f(pe.x), bar = __builtin_amdgcn_exp2f(pe.y);  // o exp(power), unclamped
                const float aal = fminf(0.99f, aar), bal = fminf(0.99f, bar);
                const bool atake = (c + ka + 1 <= last) && pe.x <= pr.o.x && aal >= ALPHA_MIN;  // power <= 0 <=> exponent <= log2 o
                const bool btake = two && (c + kb + 1 <= last) && pe.y <= pr.o.y && bal >= ALPHA_MIN;
                const v2f cg = __builtin_elementwise_fma(pr.b, v2f{g2, g2}, __builtin_elementwise_fma(pr.g, v2f{g1, g1}, pr.r * g0));
                GSR_STAT(2, two ? 2 : 1);
                GSR_STAT(3, (__builtin_amdgcn_ballot_w64(atake) != 0ull) + (__builtin_amdgcn_ballot_w64(btake) != 0ull));
                GSR_STAT(4, __popcll(__builtin_amdgcn_ballot_w64(atake)) + __popcll(__builtin_amdgcn_ballot_w64(btake)));
                // No "does any lane take it?" branch: the quadrant test above is exact, 97.5-99.7 % of the evaluated
                // entries are taken by at least one pixel (profiles/r02_kstats_bwd.txt), and the wave-wide test costs
                // two VALU instructions per entry.  A lane that does not take an entry runs the same arithmetic with
                // alpha = 0: T / (1 - 0) and rho + 0 * (...) leave its state untouched and it stores (0, 0) (the
                // factors are SELECTED, never multiplied by an inf / NaN).
                const v2f qa = {atake ? aar : 0.f, atake ? aal : 0.f};   // (o G, alpha) of the first entry, or (0, 0)
                const v2f qb = {btake ? bar : 0.f, btake ? bal : 0.f};
                const v2f inv = {__builtin_amdgcn_rcpf(1.f - qa.y), __builtin_amdgcn_rcpf(1.f - qb.y)};  // 1 ulp, inside the 1e-4 budget
                const v2f tbi = inv * tb;
                {
                    const float Tn = T * inv.x;                     // transmittance in front of the entry
                    const float dot = cg.x - rho;                   // (c - R) . g
                    const float da = fmaf(dot, Tn, -tbi.x);         // dL/dalpha
                    const v2f qw = qa * v2f{da, Tn};                // (q = o G dL/dalpha, w = alpha T)
                    smat[wave][s * MSTR + lane] = make_float2(qw.x, qw.y);
                    rho = fmaf(qa.y, dot, rho);
                    T = Tn;
                }
                sk |= (unsigned long long)ka << (8 * s);
                s++;
                if (two) {
                    const float Tn = T * inv.y;
                    const float dot = cg.y - rho;
                    const float da = fmaf(dot, Tn, -tbi.y);
                    const v2f qw = qb * v2f{da, Tn};
                    smat[wave][s * MSTR + lane] = make_float2(qw.x, qw.y);
                    rho = fmaf(qb.y, dot, rho);
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
        // Flush.  The four quadrant waves' raw pixel moments (about the TILE origin) add up; then
        //   M0 = sum q, Mx = sum q dx = ex M0 - Ax, Mxx = sum q dx^2 = ex^2 M0 - 2 ex Ax + Axx, ... (ex = x_e - tile x0)
        //   dL/dmean = -(A Mx + B My, B Mx + C My) * (W/2, H/2);  dL/d(A,B,C) = -(Mxx / 2, Mxy, Myy / 2);
        //   dL/dopacity = sum G dL/dalpha = M0 / o  (q carries the factor o);  dL/drgb = sum w g.
        // Step 1, ONE wave (they take turns), lane = entry: add the four tables (rows are 9 words apart: conflict-free)
        // and map the moments to the nine gradient values -- straight-line code, each sum formed once.  (With one lane
        // per (entry, value) every lane re-added up to 16 table words and the nine value formulas ran as divergent
        // branches in all four waves: 0.045 ms of the 0.38 ms kernel.)  Everything the maps need is in the slab of the
        // longest-walking wave: no global loads.
        // Step 2, all waves, lane = (entry, value): 9 ADJACENT lanes add the nine values of one (tile, entry) pair into
        // that Gaussian's 36-byte gradient record, so a wave instruction touches ~8 records instead of 64 scattered words.
        if (wave == ((c >> 6) & 3)) {
            const float tx0 = (float)(tx * GSR_BLOCK_X), ty0 = (float)(ty * GSR_BLOCK_Y);
            const int e9 = 9 * lane;
            float Sv[9];
#pragma unroll
            for (int v = 0; v < 9; v++) Sv[v] = (sacc[0][e9 + v] + sacc[1][e9 + v]) + (sacc[2][e9 + v] + sacc[3][e9 + v]);
            const float *fe = &fslab[7 * lane];
            const float ex = fe[0] - tx0, ey = fe[1] - ty0;
            // conic back from the log2-scaled copy: A = a2 / (-log2(e) / 2), B = b2 / (-log2 e), C likewise
            const float cA = fe[2] * (-2.0f / LOG2E), cB = fe[3] * (-1.0f / LOG2E), cC = fe[4] * (-2.0f / LOG2E);
            const float A0 = Sv[0], Ax = Sv[1], Ay = Sv[2];
            const float Mx = fmaf(ex, A0, -Ax), My = fmaf(ey, A0, -Ay);
            float o[9];
            o[0] = -(cA * Mx + cB * My) * ddelx_dx;
            o[1] = -(cB * Mx + cC * My) * ddely_dy;
            o[2] = Sv[6];
            o[3] = Sv[7];
            o[4] = Sv[8];
            o[5] = -0.5f * (fmaf(fmaf(ex, A0, -2.f * Ax), ex, Sv[3]));
            o[6] = -(fmaf(ex, My, -ey * Ax) + Sv[4]);   // ex ey M0 - ex Ay - ey Ax + Axy
            o[7] = -0.5f * (fmaf(fmaf(ey, A0, -2.f * Ay), ey, Sv[5]));
            o[8] = A0 == 0.f ? 0.f : A0 / fe[6];
            const bool live = c + lane < bmax;
            float *so = &sout[10 * lane];
#pragma unroll
            for (int v = 0; v < 9; v++) so[v] = live ? o[v] : 0.f;
            so[9] = fe[5];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int idx = threadIdx.x + 256 * r;
            const int e = idx / 9, col = idx - 9 * e;
            if (idx < 576) {
                const float val = sout[10 * e + col];
#ifndef GSR_ABL_NOATOMIC
                if (val != 0.f) atomicAdd(dL_record + 9 * (size_t)__float_as_uint(sout[10 * e + 9]) + col, val);
#else
                asm volatile("" ::"v"(val));
#endif
            }
        }
        // every wave clears ITS OWN table (the mapping wave read it before the barrier above; LDS operations of one wave
        // execute in order, so the next chunk's stores of this wave follow the clears)
#pragma unroll
        for (int i = 0; i < 9; i++) acc_tab[64 * i + lane] = 0.f;
        raw = raw_nxt;
        id_cur = id_nxt;
        id_nxt = id_nxt2;
    }
}

__global__ void __launch_bounds__(256, 4)  // 4 waves per SIMD: <= 128 VGPRs (LDS admits 4 workgroups per CU)
composite_backward_kernel(int W, int H, int gx, int tiles, const int2 *__restrict__ ranges,
                          const uint32_t *__restrict__ point_list, const float2 *__restrict__ means2D,
                          const float4 *__restrict__ conic_opacity, const float *__restrict__ rgb,
                          const uint8_t *__restrict__ compute_locally, const float *__restrict__ bg,
                          const float *__restrict__ final_T, const int32_t *__restrict__ n_contrib,
                          const float *__restrict__ dL_dpixels, float *__restrict__ dL_record,
                          const float *__restrict__ out_color, const SegWs seg, int band_first, int band_tiles) {
    // workgroups [0, nstatic): segment 0 of every tile (of the band, when the caller named it: see K8), XCD-contiguous
    // spans.  The others are persistent workers: they take the segments K8 queued (their number is only known on the
    // device) by atomic ticket.
    const int nstatic = band_tiles > 0 ? band_tiles : tiles;
    const bool worker = (int)blockIdx.x >= nstatic;
    if (band_first < 0 && band_tiles > 0) {  // a capacity grid (see K8): the band is the hull behind the range table
        const int gy = tiles / gx;
        int2 hull = ranges[tiles];
        if (hull.x < 0 || hull.y > gy || hull.x >= hull.y) hull = make_int2(0, 0);
        band_first = hull.x * gx;
        band_tiles = min((hull.y - hull.x) * gx, band_tiles);
        if (!worker && (int)blockIdx.x >= band_tiles) return;
    }
    __shared__ uint32_t s_item;
    const uint32_t count = worker ? min(seg.hdr[0], seg.cap) : 0u;
    for (;;) {
        int tile, sidx = 0;
        if (worker) {
            __syncthreads();  // (also: the previous segment's LDS is no longer read)
            if (threadIdx.x == 0) s_item = atomicAdd(&seg.hdr[1], 1u);
            __syncthreads();
            const uint32_t t = s_item;
            if (t >= count) return;
            const uint32_t item = seg.queue[t];
            tile = (int)(item >> 5);
            sidx = (int)(item & 31u);
        } else {
            tile = band_tiles > 0 ? band_first + gsr_xcd_span_of_block(blockIdx.x, band_tiles)
                                  : composite_tile_of_block(ranges, gx, tiles);
        }
        composite_backward_segment(tile, sidx, W, H, gx, ranges, point_list, means2D, conic_opacity, rgb,
                                   compute_locally, bg, final_T, n_contrib, dL_dpixels, dL_record, out_color, seg);
        if (!worker) return;
    }
}

}  // namespace

extern "C" int gsr_composite_walked(unsigned long long *out2, int reset) {
    if (!out2) return GSR_EINVAL;
    GSR_HIP(hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_walked), sizeof(unsigned long long) * 2));
    if (reset) {
        const unsigned long long z[2] = {0ull, 0ull};
        GSR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_walked), z, sizeof(z)));
    }
    return 0;
}

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

size_t gsr_composite_seg_bytes(int W, int H) {
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    return seg_ws_bytes(gx * gy, SEG_CAP);
}

int gsr_launch_composite_forward(int P, int W, int H, const int32_t *ranges, const uint32_t *point_list,
                                 const float *means2D, const float *conic_opacity, const float *rgb,
                                 const uint8_t *compute_locally, const float *bg, float *out_color, float *final_T,
                                 int32_t *n_contrib, void *seg_ws, size_t seg_bytes, int row_lo, int row_hi,
                                 void *zero_ptr, size_t zero_bytes, hipStream_t stream) {
    (void)P;
    if (zero_bytes && (!zero_ptr || ((uintptr_t)zero_ptr & 15) || (zero_bytes & 3))) return GSR_EINVAL;
    // the 16-byte body goes through the kernel, a tail of < 16 bytes (P odd: 36 P is a multiple of 4 only) through a fill
    const size_t zero16_n = zero_bytes / 16;
    if (zero_bytes & 15)
        GSR_HIP(hipMemsetAsync(reinterpret_cast<char *>(zero_ptr) + zero16_n * 16, 0, zero_bytes & 15, stream));
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    // row_lo == -1: `row_hi` tile rows are a CAPACITY, the band is read from the device (see composite_forward_kernel)
    const bool dyn = row_lo == -1 && row_hi > 0 && row_hi <= gy;
    const bool band = dyn || (row_lo >= 0 && row_lo < row_hi && row_hi <= gy && !(row_lo == 0 && row_hi == gy));
    const int band_first = dyn ? -1 : band ? row_lo * gx : 0;
    const int band_tiles = dyn ? row_hi * gx : band ? (row_hi - row_lo) * gx : 0;
    if (seg_ws && seg_bytes < seg_ws_bytes(gx * gy, SEG_CAP)) return GSR_ENOSPACE;
    const SegWs seg = seg_ws_of(seg_ws, gx * gy, SEG_CAP);
    if (seg_ws) GSR_HIP(hipMemsetAsync(seg_ws, 0, 64, stream));  // { segments queued, the backward's ticket }
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(band ? band_tiles : gx * gy), dim3(256), 0, stream, W, H, gx,
                           reinterpret_cast<const int2 *>(ranges), point_list, reinterpret_cast<const float2 *>(means2D),
                           reinterpret_cast<const float4 *>(conic_opacity), rgb, compute_locally, bg, out_color, final_T,
                           n_contrib, seg, band_first, band_tiles, reinterpret_cast<uint4 *>(zero_ptr), zero16_n);
    };
    if (seg_ws && band) launch(composite_forward_kernel<true, true>);
    else if (seg_ws) launch(composite_forward_kernel<true, false>);
    else if (band) launch(composite_forward_kernel<false, true>);
    else launch(composite_forward_kernel<false, false>);
    GSR_LAUNCH_CHECK();
    return 0;
}

int gsr_launch_composite_backward(int P, int W, int H, const int32_t *ranges, const uint32_t *point_list,
                                  const float *means2D, const float *conic_opacity, const float *rgb,
                                  const uint8_t *compute_locally, const float *bg, const float *final_T,
                                  const int32_t *n_contrib, const float *dL_dpixels, float *dL_record,
                                  const float *out_color, void *seg_ws, size_t seg_bytes, int row_lo, int row_hi,
                                  int record_is_zero, hipStream_t stream) {
    // (record_is_zero: the forward launch cleared it, gsr_render_forward_seg_z)
    if (!record_is_zero) GSR_HIP(hipMemsetAsync(dL_record, 0, sizeof(float) * 9 * (size_t)P, stream));
    if (P == 0) return 0;
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    // row_lo == -1: `row_hi` tile rows are a CAPACITY, the band is read from the device (see composite_forward_kernel)
    const bool dyn = row_lo == -1 && row_hi > 0 && row_hi <= gy;
    const bool band = dyn || (row_lo >= 0 && row_lo < row_hi && row_hi <= gy && !(row_lo == 0 && row_hi == gy));
    const int band_first = dyn ? -1 : band ? row_lo * gx : 0;
    const int band_tiles = dyn ? row_hi * gx : band ? (row_hi - row_lo) * gx : 0;
    if (seg_ws && (seg_bytes < seg_ws_bytes(gx * gy, SEG_CAP) || !out_color)) return GSR_EINVAL;
    const SegWs seg = seg_ws_of(seg_ws, gx * gy, SEG_CAP);
    // the workers' ticket is reset per launch, so that a second backward over the same forward (retain_graph, gradcheck)
    // walks the queued segments again (rearming it inside the kernel cost K10 its 128-register budget: 196 B of spills,
    // +20 % on every launch; this is a 4-byte fill on the thin-band path only)
    if (seg_ws) GSR_HIP(hipMemsetAsync(reinterpret_cast<char *>(seg_ws) + sizeof(uint32_t), 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(composite_backward_kernel, dim3((band ? band_tiles : gx * gy) + (seg_ws ? SEG_WORKERS : 0)),
                       dim3(256), 0, stream, W, H, gx, gx * gy, reinterpret_cast<const int2 *>(ranges), point_list,
                       reinterpret_cast<const float2 *>(means2D), reinterpret_cast<const float4 *>(conic_opacity), rgb,
                       compute_locally, bg, final_T, n_contrib, dL_dpixels, dL_record, out_color, seg, band_first,
                       band_tiles);
    GSR_LAUNCH_CHECK();
    return 0;
}
