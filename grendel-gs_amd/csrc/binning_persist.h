// binning_persist.h -- K3..K7 as TWO persistent launches with grid-wide barriers (round 5).  Included by binning.hip.
//
// Why: at one camera's sizes (1e6 Gaussians, 1e7 pairs) the nine launches of the look-back pipeline are latency chains.
// The P-sized chain (K3 + four depth passes + the offsets scan) moves ~100 MB -- 15 us of HBM time -- in 118 us: six
// launch ramps and six look-back ripples.  The D-sized passes run in lock-step generations of 1024 workgroups that all
// wait ~9 of their ~20 us for the generation's slowest aggregate to ripple through the look-back
// (profiles/r04_radix_timeline.txt).  Both disappear when the workgroups of a pass are all RESIDENT and meet at a
// grid barrier: a workgroup publishes its digit counts, everybody waits once, and every workgroup derives its global
// offsets from the published counts -- no state words, no ripple, no tickets.
//
//   bin_prepare_persist_kernel  (<= one 1024-thread workgroup per CU, contiguous 4096-element tiles per workgroup)
//       T   K3: rect / tile count / depth key per Gaussian (keys stay in registers), digit counts of pass 0
//       B0 A1 B1 A2 B2 A3 B3   four LSD passes: B = scatter with offsets from the published counts, A = count
//       S1 S2   offsets = exclusive scan of tiles_touched[sorted id]; the pair count D to the pinned host slot
//   bin_sort_persist_kernel     (four 512-thread workgroups per CU, contiguous 4096-pair tiles per workgroup)
//       E0  column-digit counts of the workgroup's slots FROM THE RECTS (O(Gaussians), nothing is decoded), and the
//           owner Gaussian of every 512-slot chunk (so the emission never searches)
//       E1  decode + scatter by column;  R0 count rows;  R1 scatter by row -> point_list;  T  tile ranges
//
// Coherence without fences (tools/probes/grid_barrier_probe.hip, profiles/r05_grid_barrier_probe.txt): the eight XCDs'
// L2s are not coherent with each other, and an agent-scope fence pair per barrier (L2 write-back + invalidate) costs
// 9-28 us on this chip -- more than the launches it replaces.  Instead every word that one workgroup writes and ANOTHER
// reads inside the same launch (sort keys / values, digit counts, tile counts) is written through and read at agent
// scope (global_store / global_load sc1: st_agent / ld_agent): the barrier then only needs the stores to be
// acknowledged (the workgroup barrier's s_waitcnt) before the arrival atomic -- 2-4 us.  Data that only the NEXT kernel
// reads (rects, offsets, point_list) uses plain stores.
//
// Order of the lists: identical to the look-back pipeline (stable LSD passes over the same keys) -- bit-identical
// point_list / ranges / offsets (tests/test_gpu_parity.py::test_persistent_binning_equals_the_lookback_pipeline).
//
// Co-residency: the grid never exceeds what the device holds at once (hipOccupancyMaxActiveBlocksPerMultiprocessor x
// CUs, checked on the host), so all workgroups arrive at the first barrier as soon as earlier kernels drain.  Two
// barrier kernels on DIFFERENT streams could each hold part of the machine and wait for the rest forever: the host
// admits a persistent launch only when the previous one was on the same stream or has passed its last barrier (a
// pinned word); another process sharing the device (or a graph replay, or a collective waiting for a late peer on
// another stream) is covered by a time-out at the FIRST barrier, decided for the whole grid at once -- the prepare kernel
// then reports the pair count 0xFFFFFFFF and the host repeats the call on the look-back pipeline; in the sort kernel
// workgroup 0 finishes the view alone (round 6: no trap anywhere, see bin_sort_persist_kernel and barrier_fault).
#pragma once

namespace {

// K3's word per Gaussian on the (row, column) path: pairs (<= 256 x 256 tiles: 17 bits) | rows of the rect << 20 (9 bits)
constexpr int TT_SHIFT = 20;
constexpr uint32_t TT_MASK = (1u << TT_SHIFT) - 1u;
constexpr uint32_t GB_ABORT = 0x80000000u;
constexpr int GB_FAN = 32;          // workgroups per leaf counter of the barrier tree and per count aggregate
constexpr int GB_LEAF_STRIDE = 32;  // words between leaf counters (128 bytes: one line each)
constexpr uint32_t PAIRS_ABORTED = 0xFFFFFFFFu;  // "pair count" of a prepare kernel that gave up at its first barrier

struct GridSync {
    uint32_t *leaf;   // [ngroups * GB_LEAF_STRIDE], zero before the launch: arrivals per group of GB_FAN workgroups
    uint32_t *root;   // arrived groups (monotone over the kernel's barriers); bit 31: aborted
    uint32_t *flags;  // [ngroups * GB_LEAF_STRIDE]: the epoch the group may leave (written by whoever completes the root)
};
// words of the three arrays for a grid of G workgroups
__host__ __device__ inline size_t grid_sync_words(int G) {
    return (size_t)(2 * ((G + GB_FAN - 1) / GB_FAN) + 1) * GB_LEAF_STRIDE;
}

// Barrier over the G workgroups of the grid; `epoch` counts this workgroup's barriers (uniform over the grid).  Returns
// false when the kernel was aborted (a time-out at some workgroup's barrier): the caller leaves at once.
// No fences: what crosses workgroups is written through and read at agent scope (see the head of this file); every wave
// waits for its own outstanding stores (s_waitcnt vmcnt(0): on gfx9 stores count in vmcnt) in front of the workgroup
// barrier that precedes the arrival atomic -- __syncthreads() alone compiles to `s_waitcnt lgkmcnt(0); s_barrier` here, and
// an arrival that overtakes a store of another wave would let a remote reader see the flag before the data.
// The abort decision is SINGLE-SOURCED in the root word (round 6; advisor r05): a workgroup that times out sets the abort
// bit with a compare-and-swap that fails once the root holds the complete count, and whoever completes the count looks at
// the bit in the value its own arrival returned.  So either every workgroup passes the barrier or none does: a completer
// can no longer release the flags over an abort it did not see (some groups went on, the rest had left).
// `force_abort` (test hook): this workgroup raises the abort bit BEFORE it arrives -- its group, hence the root, cannot be
// complete yet, so the abort always wins.
__device__ __forceinline__ bool grid_barrier_try_abort(const GridSync gs, uint32_t complete) {
    uint32_t r = __hip_atomic_load(gs.root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
        if (r & GB_ABORT) return true;
        if (r == complete) return false;  // the grid arrived after all: the flags are on their way
        if (__hip_atomic_compare_exchange_strong(gs.root, &r, r | GB_ABORT, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT))
            return true;
    }
}
__device__ __forceinline__ bool grid_barrier(const GridSync gs, uint32_t G, uint32_t &epoch, uint64_t timeout_ticks,
                                             uint32_t *s_flag, bool force_abort = false) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch++;
        const uint32_t g = blockIdx.x / GB_FAN, ngroups = (G + GB_FAN - 1) / GB_FAN;
        const uint32_t gsz = min((uint32_t)GB_FAN, G - g * GB_FAN);
        const uint32_t complete = epoch * ngroups;
        if (force_abort && grid_barrier_try_abort(gs, complete))
            for (uint32_t k = 0; k < ngroups; k++) st_agent(&gs.flags[k * GB_LEAF_STRIDE], 0xFFFFFFFFu);
        const uint32_t old =
            __hip_atomic_fetch_add(&gs.leaf[g * GB_LEAF_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == epoch * gsz) {
            const uint32_t r = __hip_atomic_fetch_add(gs.root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((r & ~GB_ABORT) + 1u == complete)       // the grid is complete: let every group go (G pollers on the
                for (uint32_t k = 0; k < ngroups; k++)  // root cost 7.7 us at G = 1024, a flag per group 2.6 us)
                    st_agent(&gs.flags[k * GB_LEAF_STRIDE], (r & GB_ABORT) ? 0xFFFFFFFFu : epoch);
        }
        uint64_t t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
        uint32_t f;
        while ((f = ld_agent(&gs.flags[g * GB_LEAF_STRIDE])) < epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
                // give up -- unless the count completed meanwhile: raise the abort bit of the root, then every flag (a flag
                // at 0xFFFFFFFF lets everybody leave, now and at every later barrier)
                if (grid_barrier_try_abort(gs, complete)) {
                    for (uint32_t k = 0; k < ngroups; k++) st_agent(&gs.flags[k * GB_LEAF_STRIDE], 0xFFFFFFFFu);
                    f = 0xFFFFFFFFu;
                    break;
                }
                t0 = __builtin_amdgcn_s_memrealtime();
            }
        }
        *s_flag = f == 0xFFFFFFFFu;
    }
    __syncthreads();
    return *s_flag == 0u;
}

// A barrier AFTER the first one did not complete although the whole grid is resident (a hung or heavily preempted
// device): leave { code, count } in the pinned status words and return -- no trap (round 6: a trap kills the rank's HIP
// context and its RCCL peers hang, SURVEY 8(b) "Errors").  The lists of this call are incomplete but in bounds (the range
// table only ever receives slot indices below the pair count); the host sees the count move at its next binning call on
// the device, returns GSR_EFAULT (a Python exception in the operator) and keeps to the look-back pipeline from then on.
__device__ __forceinline__ void barrier_fault(uint32_t *status, uint32_t code) {
    if (threadIdx.x == 0) {
        __hip_atomic_store(status + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_add(status + 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Global offsets of one counting pass from the published counts: cnt[G][256] (row w = workgroup w's digit counts)
// and grp[ngroups][256] (sums over groups of GB_FAN workgroups, accumulated with atomics by the producers).
// Thread d < 256 gets (digit d's pairs in workgroups before w) and the digit's total over the grid; every thread of
// the workgroup takes part: the <= ngroups + GB_FAN - 1 rows of a digit are split over THREADS / 256 threads and each
// thread's loads are issued eight at a time BEFORE the first is used (agent-scope loads cost a trip to memory each:
// one row after the other was 30 us per pass).  `red`: 2 * THREADS words of LDS scratch.  Workgroup barriers inside.
template <int THREADS>
__device__ __forceinline__ void counts_before(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ grp,
                                              uint32_t G, uint32_t w, uint32_t *__restrict__ red, uint32_t &before,
                                              uint32_t &total) {
    constexpr uint32_t Q = THREADS / RADIX_DIGITS;
    const uint32_t q = threadIdx.x / RADIX_DIGITS, d = threadIdx.x % RADIX_DIGITS;
    const uint32_t g = w / GB_FAN, ngroups = (G + GB_FAN - 1) / GB_FAN;
    uint32_t b = 0, t = 0;
    for (uint32_t k0 = 0; k0 < ngroups; k0 += 8 * Q) {
        uint32_t x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t k = k0 + q + Q * i;
            x[i] = k < ngroups ? ld_agent(&grp[k * RADIX_DIGITS + d]) : 0u;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            t += x[i];
            if (k0 + q + Q * i < g) b += x[i];
        }
    }
    for (uint32_t k0 = g * GB_FAN; k0 < w; k0 += 8 * Q) {
        uint32_t x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t k = k0 + q + Q * i;
            x[i] = k < w ? ld_agent(&cnt[(size_t)k * RADIX_DIGITS + d]) : 0u;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) b += x[i];
    }
    red[threadIdx.x] = b;
    red[THREADS + threadIdx.x] = t;
    __syncthreads();
    before = total = 0;
    if (threadIdx.x < RADIX_DIGITS) {
#pragma unroll
        for (uint32_t i = 0; i < Q; i++) {
            before += red[i * RADIX_DIGITS + threadIdx.x];
            total += red[THREADS + i * RADIX_DIGITS + threadIdx.x];
        }
    }
    __syncthreads();
}

// The same in two steps for the prepare kernel (THREADS / 256 = 4 threads per digit, at most 256 workgroups: two group
// rows and eight workgroup rows per thread): the loads are issued, the tile is ranked, then the offsets are formed.
struct CountLoads {
    uint32_t xg[2], xc[8];
};
template <int THREADS>
__device__ __forceinline__ bool counts_fit_registers(uint32_t G) {
    constexpr uint32_t Q = THREADS / RADIX_DIGITS;
    return (G + GB_FAN - 1) / GB_FAN <= 2 * Q && GB_FAN - 1 <= 8 * Q;
}
template <int THREADS>
__device__ __forceinline__ void counts_issue(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ grp,
                                             uint32_t G, uint32_t w, CountLoads &cl) {
    constexpr uint32_t Q = THREADS / RADIX_DIGITS;
    const uint32_t q = threadIdx.x / RADIX_DIGITS, d = threadIdx.x % RADIX_DIGITS;
    const uint32_t g = w / GB_FAN, ngroups = (G + GB_FAN - 1) / GB_FAN;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t k = q + Q * i;
        cl.xg[i] = k < ngroups ? ld_agent(&grp[k * RADIX_DIGITS + d]) : 0u;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t k = g * GB_FAN + q + Q * i;
        cl.xc[i] = k < w ? ld_agent(&cnt[(size_t)k * RADIX_DIGITS + d]) : 0u;
    }
}
template <int THREADS>
__device__ __forceinline__ void counts_finish(const CountLoads &cl, uint32_t w, uint32_t *__restrict__ red,
                                              uint32_t &before, uint32_t &total) {
    constexpr uint32_t Q = THREADS / RADIX_DIGITS;
    const uint32_t q = threadIdx.x / RADIX_DIGITS;
    const uint32_t g = w / GB_FAN;
    uint32_t b = 0, t = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        t += cl.xg[i];
        if (q + Q * i < g) b += cl.xg[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) b += cl.xc[i];
    red[threadIdx.x] = b;
    red[THREADS + threadIdx.x] = t;
    __syncthreads();
    before = total = 0;
    if (threadIdx.x < RADIX_DIGITS) {
#pragma unroll
        for (uint32_t i = 0; i < Q; i++) {
            before += red[i * RADIX_DIGITS + threadIdx.x];
            total += red[THREADS + i * RADIX_DIGITS + threadIdx.x];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void publish_counts(uint32_t *__restrict__ cnt, uint32_t *__restrict__ grp, uint32_t w,
                                               uint32_t d, uint32_t c) {
    st_agent(&cnt[(size_t)w * RADIX_DIGITS + d], c);
    if (c) __hip_atomic_fetch_add(&grp[(w / GB_FAN) * RADIX_DIGITS + d], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- exact tile culling (round 5; verdict r04 item 1c) ------------------------------------------------------------
// K3's rect is a bounding box: the 3-sigma rect intersected with the box of the alpha >= 1/255 ellipse.  A tile in a
// corner of that box that the ellipse itself does not reach receives nothing under the reference's alpha < 1/255 rule
// (K8 / K10 skip it with their per-quadrant test, same quadratic form, same tolerance), yet it is a pair that is emitted,
// sorted twice and walked.  With culling on, K3 works out per tile ROW of the rect the exact span of tiles the ellipse
// reaches -- closed form: the ellipse's rightmost / leftmost point inside the row's strip of pixel centres -- and keeps
// it as a 64-bit mask (rects of up to 64 tiles; larger ones keep every tile): bit (r * w + c) = tile (minx + c, miny + r).
// tiles_touched = popcount; the emission enumerates the set bits in row-major order, i.e. the lists stay
// order-preserving subsequences of the uncut ones (D -14..20 %).
// "Tile (x, y) is kept" == gsr_can_touch_box(xy, co, 16x, 16y, 16x + 15, 16y + 15): min of the form over the box <= lim.
struct alignas(16) TileRect {  // what K3 leaves per Gaussian (16 bytes, one gather in the emission)
    uint32_t xs, ys;         // minx | maxx << 16, miny | maxy << 16
    unsigned long long mask; // ~0: every tile of the rect; else the kept tiles of a rect of <= 64 tiles
};
static_assert(sizeof(TileRect) == 16, "one 16-byte gather");

__device__ __forceinline__ unsigned long long gsr_full_mask(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

__device__ __forceinline__ unsigned long long gsr_tile_mask(const float2 xy, const float4 co, int minx, int miny, int maxx,
                                                            int maxy) {
    const int w = maxx - minx, h = maxy - miny;
    if (w * h > 64) return ~0ull;
    const unsigned long long all = gsr_full_mask(w * h);
    const float A = co.x, B = co.y, C = co.z;
    const float det = A * C - B * B;
    // an improper or ill-conditioned conic is not culled (gsr_can_touch_box keeps those too)
    if (!(A > 0.f && C > 0.f && det > 1e-6f * A * C) || !(co.w >= 1.0f / 255.0f)) return all;
    const float lim = (2.0f * 0.6931471805599453f) * __builtin_amdgcn_logf(255.0f * co.w) * 1.002f + 0.01f;
    // (hardware reciprocal / square root, 1 ulp: the IEEE sequences of `/` and sqrtf() were most of this function's ~50
    // instructions per row; the 0.2 % + 0.01 inflation of lim and the 0.02 px slack below dwarf their error)
    const float rA = __builtin_amdgcn_rcpf(A), rC = __builtin_amdgcn_rcpf(C);
    const float dxs = __builtin_amdgcn_sqrtf(lim * C * __builtin_amdgcn_rcpf(det));  // half extent in x of the whole
    const float dys = B * dxs * rC;                                                  // ellipse, reached at dy = -/+ dys
    unsigned long long mask = 0ull;
    for (int r = 0; r < h; r++) {
        const float t0 = (float)((miny + r) * GSR_BLOCK_Y) - xy.y, t1 = t0 + (float)(GSR_BLOCK_Y - 1);  // the row's strip
        float dxR, dxL;
        bool emptyR = false, emptyL = false;
        if (-dys >= t0 && -dys <= t1) {
            dxR = dxs;
        } else {  // the rightmost point of the ellipse lies outside the strip: the extreme is on the nearer edge
            const float t = -dys < t0 ? t0 : t1;
            const float disc = lim * A - det * t * t;
            emptyR = disc < 0.f;
            dxR = (-B * t + __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f))) * rA;
        }
        if (dys >= t0 && dys <= t1) {
            dxL = -dxs;
        } else {
            const float t = dys < t0 ? t0 : t1;
            const float disc = lim * A - det * t * t;
            emptyL = disc < 0.f;
            dxL = (-B * t - __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f))) * rA;
        }
        int xa = 0, xb = w;  // columns kept in this row, relative to minx
        if (emptyR && emptyL) continue;  // the strip misses the ellipse
        if (!emptyR && !emptyL) {        // (they disagree only at a tangency: keep the row)
            dxR += 0.02f + 1e-4f * fabsf(dxR);
            dxL -= 0.02f + 1e-4f * fabsf(dxL);
            // tile x covers pixel centres [16 x, 16 x + 15]: kept when that interval meets [cx + dxL, cx + dxR]
            xa = max(0, (int)ceilf((xy.x + dxL - (float)(GSR_BLOCK_X - 1)) * (1.0f / GSR_BLOCK_X)) - minx);
            xb = min(w, (int)floorf((xy.x + dxR) * (1.0f / GSR_BLOCK_X)) + 1 - minx);
            if (xb <= xa) continue;
        }
        mask |= gsr_full_mask(xb - xa) << (r * w + xa);
    }
    return mask & all;
}

// position of the t-th (0-based) set bit of m; t < popcount(m)
__device__ __forceinline__ uint32_t gsr_select_bit(unsigned long long m, uint32_t t) {
    uint32_t v = (uint32_t)m, base = 0;
    const uint32_t c = __popc(v);
    if (t >= c) { t -= c; v = (uint32_t)(m >> 32); base = 32; }
    const uint32_t c16 = __popc(v & 0xFFFFu);
    if (t >= c16) { t -= c16; v >>= 16; base += 16; }
    const uint32_t c8 = __popc(v & 0xFFu);
    if (t >= c8) { t -= c8; v >>= 8; base += 8; }
    const uint32_t c4 = __popc(v & 0xFu);
    if (t >= c4) { t -= c4; v >>= 4; base += 4; }
    const uint32_t c2 = __popc(v & 3u);
    if (t >= c2) { t -= c2; v >>= 2; base += 2; }
    if (t >= (v & 1u)) base += 1;
    return base;
}

// K3's tail for one Gaussian with a non-empty rect: the tile mask, the tile count and the tile sort's digit histograms
// (dxy[0]: column digits, dxy[1]: row digits, dxy[2]: Gaussians per row -- uncut rects only --, all as difference
// arrays) -- shared by the two pipelines' K3.
__device__ __forceinline__ uint32_t gsr_rect_tiles(const float2 xy, const float4 co, int minx, int miny, int maxx, int maxy,
                                                   bool cull, int32_t (*dxy)[RADIX_DIGITS + 1], TileRect &out) {
    const int w = maxx - minx, h = maxy - miny;
    out.xs = (uint32_t)minx | ((uint32_t)maxx << 16);
    out.ys = (uint32_t)miny | ((uint32_t)maxy << 16);
    out.mask = ~0ull;
    if (!cull || w * h > 64) {  // every tile of the rect (mask ~0: "not culled"): h more in its columns, w in its rows
        if (dxy) {
            atomicAdd(&dxy[0][minx], h);
            atomicAdd(&dxy[0][maxx], -h);
            atomicAdd(&dxy[1][miny], w);
            atomicAdd(&dxy[1][maxy], -w);
            atomicAdd(&dxy[2][miny], 1);  // (round 6) row SEGMENTS per tile row: one per Gaussian and row of its rect
            atomicAdd(&dxy[2][maxy], -1);
        }
        return (uint32_t)(w * h);
    }
    const unsigned long long mask = gsr_tile_mask(xy, co, minx, miny, maxx, maxy);
    out.mask = mask;
    if (dxy) {
        // the tile sort's digit histograms as difference arrays.  Rows of an ellipse mostly keep the same span, so equal
        // consecutive runs are merged into one +m / -m pair (columns) and a row only touches the row-digit array where
        // its tile count differs from the row above (LDS atomics on a handful of hot addresses are what K3 pays here)
        const unsigned long long rowm = gsr_full_mask(w);
        int prev_cnt = 0, run_a0 = 0, run_len = 0, run_mult = 0;
        for (int r = 0; r < h; r++) {
            unsigned long long bits = (mask >> (r * w)) & rowm;
            const int cnt = __popcll(bits);
            if (cnt != prev_cnt) atomicAdd(&dxy[1][miny + r], cnt - prev_cnt);
            prev_cnt = cnt;
            while (bits) {  // one run per row (the kept tiles of a row are contiguous), written for any pattern
                const int a0 = __ffsll((long long)bits) - 1;
                const unsigned long long rest = ~(bits >> a0);
                const int len = rest ? __ffsll((long long)rest) - 1 : 64 - a0;
                if (run_mult && a0 == run_a0 && len == run_len) {
                    run_mult++;
                } else {
                    if (run_mult) {
                        atomicAdd(&dxy[0][minx + run_a0], run_mult);
                        atomicAdd(&dxy[0][minx + run_a0 + run_len], -run_mult);
                    }
                    run_a0 = a0; run_len = len; run_mult = 1;
                }
                bits &= ~(gsr_full_mask(len) << a0);
            }
        }
        if (run_mult) {
            atomicAdd(&dxy[0][minx + run_a0], run_mult);
            atomicAdd(&dxy[0][minx + run_a0 + run_len], -run_mult);
        }
        if (prev_cnt) atomicAdd(&dxy[1][maxy], -prev_cnt);
    }
    return (uint32_t)__popcll(mask);
}

// LDS of a persistent sort workgroup (the staging area doubles as scratch of the phases that do not scatter)
template <int ITEMS, int THREADS>
struct PersistSmem {
    uint16_t wtab[THREADS / 64][RADIX_DIGITS];  // per-wave digit counts, then per-wave cursors
    uint32_t gbase[RADIX_DIGITS];               // global start of the digit's run minus its start inside the tile
    uint32_t skey[ITEMS * THREADS], sval[ITEMS * THREADS];
    uint32_t scan_tmp[THREADS / 64];
    unsigned long long scan64[THREADS / 64];
    uint32_t flag;
};

template <int ITEMS, int THREADS>
__device__ __forceinline__ void clear_wtab(PersistSmem<ITEMS, THREADS> &sm) {
    uint32_t *p = reinterpret_cast<uint32_t *>(&sm.wtab[0][0]);
    for (int i = threadIdx.x; i < (THREADS / 64) * RADIX_DIGITS / 2; i += THREADS) p[i] = 0u;
}

// per-wave digit counts of the ITEMS pairs a thread holds (element r of a lane: wbase + r * 64 + lane < n is valid)
template <int ITEMS, int THREADS>
__device__ __forceinline__ void count_wave_digits(PersistSmem<ITEMS, THREADS> &sm, const uint32_t (&key)[ITEMS],
                                                  long long wbase, long long n, int shift, uint32_t mask) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        if (wbase + r * 64 + lane < n) {
            const uint32_t dg = (key[r] >> shift) & mask;
            atomicAdd(reinterpret_cast<uint32_t *>(sm.wtab[wave]) + (dg >> 1), 1u << (16 * (dg & 1u)));  // no carry: <= 512
        }
    }
}

// One tile of a scatter phase, in two steps so that the caller can have the loads of its global offsets in flight
// during the first.  scatter_rank -- in: per-wave digit counts in sm.wtab (count_wave_digits + a workgroup barrier), the
// pairs in registers; places the pairs in digit order in the LDS staging area (stable: waves in order, rounds in order,
// lanes in order); thread d < 256 gets the tile's count of digit d and the digit's start inside the tile.
// scatter_write -- `first` (thread d < 256) = global position of the first pair of digit d that THIS tile writes;
// streams the staged pairs out (coalesced runs per digit).  Ends with a workgroup barrier (LDS reusable).
template <int ITEMS, int THREADS>
__device__ __forceinline__ void scatter_rank(PersistSmem<ITEMS, THREADS> &sm, const uint32_t (&key)[ITEMS],
                                             const uint32_t (&val)[ITEMS], long long tbase, long long n, int shift,
                                             int nbits, uint32_t &tot_out, uint32_t &run_out) {
    constexpr int WAVES = THREADS / 64;
    const uint32_t mask = (1u << nbits) - 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wbase = tbase + (long long)wave * (ITEMS * 64);
    const uint32_t d = threadIdx.x;
    uint32_t cnt[WAVES], tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) {
        cnt[w] = d < RADIX_DIGITS ? sm.wtab[w][d] : 0u;
        tot += cnt[w];
    }
    uint32_t all;
    uint32_t run = block_exclusive_scan_n<WAVES>(tot, sm.scan_tmp, &all);  // start of digit d inside the tile
    tot_out = tot;
    run_out = run;
    if (d < RADIX_DIGITS) {
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            sm.wtab[w][d] = (uint16_t)run;
            run += cnt[w];
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const bool valid = wbase + r * 64 + lane < n;
        const uint32_t dg = (key[r] >> shift) & mask;
        const unsigned long long m = match_digit(dg, valid, nbits);
        const uint32_t rank = __popcll(m & lt);
        uint16_t *cursor = sm.wtab[wave];
        uint32_t pos = 0;
        if (valid) pos = cursor[dg] + rank;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) cursor[dg] = (uint16_t)(pos + (uint32_t)__popcll(m));  // group leader advances the cursor
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            sm.skey[pos] = key[r];
            sm.sval[pos] = val[r];
        }
    }
}

template <int ITEMS, int THREADS, bool WT_VALS>
__device__ __forceinline__ void scatter_write(PersistSmem<ITEMS, THREADS> &sm, long long tbase, long long n, int shift,
                                              int nbits, uint32_t first, uint32_t run, uint32_t *__restrict__ keys_out,
                                              uint32_t *__restrict__ vals_out) {
    constexpr int TILE = ITEMS * THREADS;
    const uint32_t mask = (1u << nbits) - 1u;
    if (threadIdx.x < RADIX_DIGITS) sm.gbase[threadIdx.x] = first - run;
    __syncthreads();  // (also: every wave has staged its pairs)
    const long long rem = n - tbase;
    const int count = rem < TILE ? (int)rem : TILE;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const int i = r * THREADS + threadIdx.x;
        if (i < count) {
            const uint32_t k = sm.skey[i];
            const uint32_t dst = sm.gbase[(k >> shift) & mask] + (uint32_t)i;
            st_agent(&keys_out[dst], k);  // (read by other workgroups in the next phase: written through)
            if (WT_VALS) st_agent(&vals_out[dst], sm.sval[i]);
            else vals_out[dst] = sm.sval[i];
        }
    }
    __syncthreads();
}

template <int ITEMS, int THREADS, bool WT_VALS>
__device__ __forceinline__ uint32_t scatter_tile(PersistSmem<ITEMS, THREADS> &sm, const uint32_t (&key)[ITEMS],
                                                 const uint32_t (&val)[ITEMS], long long tbase, long long n, int shift,
                                                 int nbits, uint32_t first, uint32_t *__restrict__ keys_out,
                                                 uint32_t *__restrict__ vals_out) {
    uint32_t tot, run;
    scatter_rank(sm, key, val, tbase, n, shift, nbits, tot, run);
    scatter_write<ITEMS, THREADS, WT_VALS>(sm, tbase, n, shift, nbits, first, run, keys_out, vals_out);
    return tot;
}

__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Round 6: the pair count D = sum of tiles_touched is known when K3 has looked at every Gaussian -- not four sort passes
// and a scan later (~100 us at 10^6 Gaussians), which is what the host's poll used to wait for.  Every workgroup adds
// its share to a 64-bit counter (zero before the launch); the one whose arrival completes the grid hands the sum to the
// pinned host words: the value, then (system-scope release) the call's sequence tag.
__device__ __forceinline__ uint32_t clamp_pair_count(unsigned long long d) {
    return d >= 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)d;  // (0xFFFFFFFF is PAIRS_ABORTED)
}
__device__ __forceinline__ void publish_pair_count(unsigned long long *early, unsigned long long mine, uint32_t nwg,
                                                   uint32_t *host_total, uint32_t seq) {
    // (agent scope: the workgroups sit on eight XCDs whose L2s do not see each other's lines)
    __hip_atomic_fetch_add(early, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t *ticket = reinterpret_cast<uint32_t *>(early + 1);
    // release / acquire on the ticket orders every workgroup's add in front of the last arriver's read
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t + 1u == nwg && host_total) {
        const unsigned long long d = __hip_atomic_load(early, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(host_total, clamp_pair_count(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_total + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#define GSR_TS(k)                                                                                       \
    do {                                                                                                \
        if (a.tstamp && threadIdx.x == 0) a.tstamp[(size_t)blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// ======================================================================================= the P-sized chain
constexpr int PP_THREADS = 1024, PP_WAVES = PP_THREADS / 64;
// Elements per thread: 4 (tiles of 4096: up to CUs x 4096 Gaussians with ONE tile per workgroup -- its pairs never leave
// the registers between a count phase and its scatter phase) or 8 (tiles of 8192: twice as many).  Several tiles per
// workgroup work (tests run them through GSR_BIN_GRID_P) but lose to the look-back pipeline: every phase then lasts as
// long as the workgroups with one tile more (measured on configs[2]'s 6 M Gaussians, 5.7 tiles per workgroup: +0.11 ms),
// so production launches need one tile per workgroup (GSR_BIN_MAX_TPW raises it for experiments).
constexpr int PP_MAX_TPW = 8;

struct PrepPersistArgs {
    int P, gx, gy;
    const float2 *means2D;
    const float *depths;
    const int32_t *radii;
    const float4 *conic_opacity;
    const uint8_t *mask;
    uint32_t *tt, *kA, *vA, *kB, *vB, *offsets;
    uint32_t *segoff;  // exclusive scan of the rects' rows in depth order (segoff[P] = row segments R)
    TileRect *rects;
    uint32_t *tile_hist;  // [8 replicas][4][256] or null (frames above 256 x 256 tiles)
    int cull;             // exact tile culling (gsr_tile_mask); only with tile_hist (the (row, column) path)
    int32_t *hull_out;
    unsigned long long *early;  // { sum of tiles_touched, arrivals }: zero before the launch
    uint4 *zero16;              // the control block of the tile sort that follows (cleared here), or null
    size_t zero16_n;
    GridSync sync;
    uint32_t *cnt;  // [4][G][256]
    uint32_t *grp;  // [4][ngroups][256], zero before the launch
    unsigned long long *wtot;  // [G]
    uint32_t *host_total;      // pinned: { pair count, sequence tag }
    uint32_t seq;
    uint32_t *done_word;  // pinned: sequence number of the last persistent launch that passed its last barrier
    uint32_t done_seq;    // 0: do not publish (captured launches)
    uint64_t timeout_ticks;
    int force_abort;  // test hook (GSR_BIN_FORCE_ABORT=p): the first barrier aborts as if it had timed out
    unsigned long long *tstamp;  // diagnostics (GSR_BIN_TIMELINE=1): [G][32] clock stamps of thread 0, else null
};

struct PPExtra {
    int32_t dxy[3][RADIX_DIGITS + 1];
    int s_lo, s_hi;
};


template <int PP_ITEMS>
__global__ void __launch_bounds__(PP_THREADS)
bin_prepare_persist_kernel(const PrepPersistArgs a) {
    constexpr int PP_TILE = PP_THREADS * PP_ITEMS;
    __shared__ PersistSmem<PP_ITEMS, PP_THREADS> sm;
    __shared__ PPExtra ex;
    __shared__ uint32_t red2[2 * PP_THREADS];  // reduction scratch of counts_finish (the staging area is in use then)
    const uint32_t G = gridDim.x, w = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long P = a.P;
    const long long nb = (P + PP_TILE - 1) / PP_TILE;
    const long long t0 = (long long)w * nb / G, t1 = (long long)(w + 1) * nb / G;  // contiguous tiles of this workgroup
    const bool keep = (t1 - t0) == 1;  // one tile: its pairs live in registers from a count phase to its scatter phase
    const uint32_t d = threadIdx.x;
    uint32_t epoch = 0;
    GSR_TS(0);
    uint32_t key[PP_ITEMS], val[PP_ITEMS];
    for (size_t i = (size_t)w * PP_THREADS + threadIdx.x; i < a.zero16_n; i += (size_t)G * PP_THREADS)
        a.zero16[i] = make_uint4(0u, 0u, 0u, 0u);  // (the next kernel's control block: see touch_count_kernel)

    // ------------------------------------------------------------------ T: K3 (see touch_count_kernel)
    if (threadIdx.x <= RADIX_DIGITS) ex.dxy[0][threadIdx.x] = ex.dxy[1][threadIdx.x] = ex.dxy[2][threadIdx.x] = 0;
    if (threadIdx.x == 0) { ex.s_lo = a.gy; ex.s_hi = 0; }
    clear_wtab(sm);
    __syncthreads();
    {
        const int total = a.gx * a.gy;
        const int per = (total + PP_THREADS - 1) / PP_THREADS;
        const int b0 = threadIdx.x * per, b1 = min(b0 + per, total);
        int first = 0x7fffffff, last = -1;  // (branch-free: the byte loads are independent)
#pragma unroll 8
        for (int b = b0; b < b1; b++) {
            const int on = a.mask[b] != 0;
            first = min(first, on ? b : 0x7fffffff);
            last = max(last, on ? b : -1);
        }
        if (last >= 0) {
            atomicMin(&ex.s_lo, first / a.gx);
            atomicMax(&ex.s_hi, last / a.gx + 1);
        }
    }
    __syncthreads();
    const int hull0 = ex.s_lo, hull1 = ex.s_hi;
    if (w == 0 && threadIdx.x == 0) {
        a.hull_out[0] = hull1 > hull0 ? hull0 : 0;
        a.hull_out[1] = hull1 > hull0 ? hull1 : 0;
    }
    uint32_t mytot = 0;  // thread d < 256: this workgroup's count of digit d in the pass being counted
    unsigned long long nsum = 0;  // this thread's share of the pair count
    for (long long t = t0; t < t1; t++) {
        const long long wbase = t * PP_TILE + (long long)wave * (PP_ITEMS * 64);
        // every input of four of the thread's Gaussians is requested before the first is looked at (clamped indices, no
        // branches): fetched where needed, the radius -> position / conic -> depth chain cost 20 us of this phase
#pragma unroll
        for (int r0 = 0; r0 < PP_ITEMS; r0 += 4) {
        int rad[4];
        float2 xy[4];
        float4 co[4];
        float dep[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            const int r = r0 + q4;
            const long long i = wbase + r * 64 + lane;
            const long long ic = i < P ? i : P - 1;
            rad[q4] = a.radii[ic];
            xy[q4] = a.means2D[ic];
            co[q4] = a.conic_opacity[ic];
            dep[q4] = a.depths[ic];
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            const int r = r0 + q4;
            const long long i = wbase + r * 64 + lane;
            key[r] = 0xFFFFFFFFu;
            val[r] = (uint32_t)i;
            if (i < P) {
                uint32_t n = 0;
                TileRect rect{0u, 0u, 0ull};
                if (rad[q4] > 0) {
                    float exx, eyy;
                    if (gsr_alpha_extent(co[q4], exx, eyy)) {
                        int minx, miny, maxx, maxy;
                        gsr_get_rect(xy[q4].x, xy[q4].y, rad[q4], a.gx, a.gy, minx, miny, maxx, maxy);
                        minx = max(minx, (int)ceilf((xy[q4].x - exx - (GSR_BLOCK_X - 1)) * (1.0f / GSR_BLOCK_X)));
                        maxx = min(maxx, (int)floorf((xy[q4].x + exx) * (1.0f / GSR_BLOCK_X)) + 1);
                        miny = max(max(miny, hull0), (int)ceilf((xy[q4].y - eyy - (GSR_BLOCK_Y - 1)) * (1.0f / GSR_BLOCK_Y)));
                        maxy = min(min(maxy, hull1), (int)floorf((xy[q4].y + eyy) * (1.0f / GSR_BLOCK_Y)) + 1);
                        if (maxx > minx && maxy > miny)
                            n = gsr_rect_tiles(xy[q4], co[q4], minx, miny, maxx, maxy, a.cull != 0,
                                               a.tile_hist ? ex.dxy : nullptr, rect);
                    }
                }
                if (n) key[r] = __float_as_uint(dep[q4]);
                nsum += n;
                // (gathered by other workgroups in the scan phase; the rows of the rect -- its row segments -- ride in the
                // upper bits on the (row, column) path)
                st_agent(&a.tt[i], (a.tile_hist && n) ? (n | (((rect.ys >> 16) - (rect.ys & 0xFFFFu)) << TT_SHIFT)) : n);
                a.rects[i] = rect;
                if (!keep) {
                    st_agent(&a.kA[i], key[r]);
                    st_agent(&a.vA[i], val[r]);
                }
            }
        }
        }
        count_wave_digits(sm, key, wbase, P, 0, 0xFFu);
        __syncthreads();
        if (d < RADIX_DIGITS) {
#pragma unroll
            for (int wv = 0; wv < PP_WAVES; wv++) mytot += sm.wtab[wv][d];
        }
        if (!keep) {
            __syncthreads();
            clear_wtab(sm);
            __syncthreads();
        }
    }
    if (d < RADIX_DIGITS) publish_counts(a.cnt, a.grp, w, d, mytot);
    {  // this workgroup's share of the pair count (read by workgroup 0 behind the first barrier): ONE atomic per workgroup
        nsum = wave_sum64(nsum);
        __syncthreads();  // (scan64 is free: nothing has used it yet, but keep the phases apart)
        if (lane == 0) sm.scan64[wave] = nsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long tot = 0;
            for (int wv = 0; wv < PP_WAVES; wv++) tot += sm.scan64[wv];
            if (tot) __hip_atomic_fetch_add(a.early, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a.tile_hist) {  // the tile sort's digit histograms, for the look-back tile sort (large D, contended device)
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;  // HW_REG_XCC_ID[3:0]
        for (int p = 0; p < 3; p++) {
            const uint32_t v = threadIdx.x < RADIX_DIGITS ? (uint32_t)ex.dxy[p][threadIdx.x] : 0u;
            uint32_t all;
            const uint32_t c = block_exclusive_scan_n<PP_WAVES>(v, sm.scan_tmp, &all) + v;  // inclusive: the count
            if (threadIdx.x < RADIX_DIGITS && c)
                __hip_atomic_fetch_add(&a.tile_hist[(xcc * RADIX_MAX_PASSES + p) * RADIX_DIGITS + threadIdx.x], c,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    GSR_TS(1);
    if (!grid_barrier(a.sync, G, epoch, a.timeout_ticks, &sm.flag, a.force_abort != 0)) {
        // gave up waiting for the whole grid to become resident (another barrier kernel holds part of the device):
        // the host repeats the call on the look-back pipeline; the bounded tile sort sees a count above any capacity
        if (threadIdx.x == 0) {
            a.offsets[P] = PAIRS_ABORTED;
            __hip_atomic_store(a.host_total, PAIRS_ABORTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.host_total + 1, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (a.done_seq) __hip_atomic_store(a.done_word, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    GSR_TS(2);
    if (w == 0 && threadIdx.x == 0) {
        // round 6: the host's poll ends HERE (every workgroup's share arrived before its barrier arrival), ~20 us into the
        // kernel instead of at its end: the four sort passes and the scan no longer stand between K3 and the host
        const unsigned long long dsum = __hip_atomic_load(a.early, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.host_total, clamp_pair_count(dsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.host_total + 1, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // later barriers cannot dead-lock (the whole grid is resident): a second without progress is a fault (barrier_fault)
    const uint64_t forever = 100000000ull;
    // ------------------------------------------------------------------ four LSD passes over the depth bits
    const uint32_t ngroups = (G + GB_FAN - 1) / GB_FAN;
    for (int p = 0; p < 4; p++) {
        const int shift = 8 * p;
        uint32_t *const ksrc = (p & 1) ? a.kB : a.kA, *const vsrc = (p & 1) ? a.vB : a.vA;
        uint32_t *const kdst = (p & 1) ? a.kA : a.kB, *const vdst = (p & 1) ? a.vA : a.vB;
        // B: scatter
        const uint32_t *const cnt_p = a.cnt + (size_t)p * G * RADIX_DIGITS;
        const uint32_t *const grp_p = a.grp + (size_t)p * ngroups * RADIX_DIGITS;
        uint32_t before, total, all, first;
        if (keep && counts_fit_registers<PP_THREADS>(G)) {
            // one tile, its pairs and per-wave counts are in place: rank it while the counts of the other workgroups
            // travel (an agent-scope load is a trip to memory), then form the offsets and stream the tile out
            CountLoads cl;
            counts_issue<PP_THREADS>(cnt_p, grp_p, G, w, cl);
            uint32_t tot, run;
            scatter_rank(sm, key, val, t0 * PP_TILE, P, shift, 8, tot, run);
            // (the reduction scratch must not be the staging area: the ranked pairs are in it)
            counts_finish<PP_THREADS>(cl, w, reinterpret_cast<uint32_t *>(red2), before, total);
            first = block_exclusive_scan_n<PP_WAVES>(total, sm.scan_tmp, &all) + before;
            scatter_write<PP_ITEMS, PP_THREADS, true>(sm, t0 * PP_TILE, P, shift, 8, first, run, kdst, vdst);
        } else {
        counts_before<PP_THREADS>(cnt_p, grp_p, G, w, sm.skey, before, total);
        first = block_exclusive_scan_n<PP_WAVES>(total, sm.scan_tmp, &all) + before;
        for (long long t = t0; t < t1; t++) {
            const long long tbase = t * PP_TILE;
            const long long wbase = tbase + (long long)wave * (PP_ITEMS * 64);
            if (!keep) {
#pragma unroll
                for (int r = 0; r < PP_ITEMS; r++) {
                    const long long j = wbase + r * 64 + lane;
                    key[r] = j < P ? ld_agent(&ksrc[j]) : 0xFFFFFFFFu;
                    val[r] = j < P ? ld_agent(&vsrc[j]) : 0u;
                }
                count_wave_digits(sm, key, wbase, P, shift, 0xFFu);
                __syncthreads();
            }
            first += scatter_tile<PP_ITEMS, PP_THREADS, true>(sm, key, val, tbase, P, shift, 8, first, kdst, vdst);
            if (!keep) {
                clear_wtab(sm);
                __syncthreads();
            }
        }
        }
        GSR_TS(3 + 4 * p);
        if (!grid_barrier(a.sync, G, epoch, forever, &sm.flag)) { barrier_fault(a.done_word, 0x100u + epoch); return; }
        GSR_TS(4 + 4 * p);
        if (p == 3) break;
        // A: count the next digit of what this workgroup now owns
        if (keep) {
            clear_wtab(sm);
            __syncthreads();
        }
        mytot = 0;
        for (long long t = t0; t < t1; t++) {
            const long long wbase = t * PP_TILE + (long long)wave * (PP_ITEMS * 64);
#pragma unroll
            for (int r = 0; r < PP_ITEMS; r++) {
                const long long j = wbase + r * 64 + lane;
                key[r] = j < P ? ld_agent(&kdst[j]) : 0xFFFFFFFFu;
                val[r] = j < P ? ld_agent(&vdst[j]) : 0u;
            }
            count_wave_digits(sm, key, wbase, P, shift + 8, 0xFFu);
            __syncthreads();
            if (d < RADIX_DIGITS) {
#pragma unroll
                for (int wv = 0; wv < PP_WAVES; wv++) mytot += sm.wtab[wv][d];
            }
            if (!keep) {
                __syncthreads();
                clear_wtab(sm);
                __syncthreads();
            }
        }
        if (d < RADIX_DIGITS)
            publish_counts(a.cnt + (size_t)(p + 1) * G * RADIX_DIGITS, a.grp + (size_t)(p + 1) * ngroups * RADIX_DIGITS,
                           w, d, mytot);
        GSR_TS(5 + 4 * p);
        if (!grid_barrier(a.sync, G, epoch, forever, &sm.flag)) { barrier_fault(a.done_word, 0x100u + epoch); return; }
        GSR_TS(6 + 4 * p);
    }
    // four passes: the sorted (key, id) pairs are back in (kA, vA)
    // ------------------------------------------------------------------ S: offsets = exclusive scan of tt[vA[.]]
    // (and, round 6, segoff = exclusive scan of hh[vA[.]]: the row segments of the row-major emission, binning_rows.h;
    // the two running sums travel in one 64-bit word -- both stay below 2^31, RADIX_MAX_N -- low half pairs, high half
    // segments)
    uint32_t v[PP_ITEMS], hv[PP_ITEMS];
    unsigned long long wsum = 0;
    for (long long t = t0; t < t1; t++) {
        const long long base = t * PP_TILE + (long long)threadIdx.x * PP_ITEMS;  // four CONSECUTIVE elements per thread
        uint32_t s = 0, sh = 0;
#pragma unroll
        for (int k = 0; k < PP_ITEMS; k++) {
            const uint32_t x = (base + k < P) ? ld_agent(&a.tt[ld_agent(&a.vA[base + k])]) : 0u;
            v[k] = a.tile_hist ? (x & TT_MASK) : x;
            hv[k] = a.tile_hist ? (x >> TT_SHIFT) : 0u;
            s += v[k];
            sh += hv[k];
        }
        wsum += (unsigned long long)s | ((unsigned long long)sh << 32);
    }
    wsum = wave_sum64(wsum);
    if (lane == 0) sm.scan64[wave] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int wv = 0; wv < PP_WAVES; wv++) tot += sm.scan64[wv];
        st_agent64(&a.wtot[w], tot);
    }
    GSR_TS(19);
    if (!grid_barrier(a.sync, G, epoch, forever, &sm.flag)) { barrier_fault(a.done_word, 0x100u + epoch); return; }
    GSR_TS(20);
    if (w == 0 && threadIdx.x == 0 && a.done_seq)  // every workgroup is past the last barrier: nothing left to wait for
        __hip_atomic_store(a.done_word, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned long long carry;
    {
        unsigned long long x = 0;
        for (uint32_t k = threadIdx.x; k < w; k += PP_THREADS) x += ld_agent64(&a.wtot[k]);
        x = wave_sum64(x);
        __syncthreads();  // (scan64 was read above)
        if (lane == 0) sm.scan64[wave] = x;
        __syncthreads();
        carry = 0;
        for (int wv = 0; wv < PP_WAVES; wv++) carry += sm.scan64[wv];
    }
    for (long long t = t0; t < t1; t++) {
        const long long base = t * PP_TILE + (long long)threadIdx.x * PP_ITEMS;
        uint32_t s = 0, sh = 0;
        if (!keep) {
#pragma unroll
            for (int k = 0; k < PP_ITEMS; k++) {
                const uint32_t x = (base + k < P) ? ld_agent(&a.tt[ld_agent(&a.vA[base + k])]) : 0u;
                v[k] = a.tile_hist ? (x & TT_MASK) : x;
                hv[k] = a.tile_hist ? (x >> TT_SHIFT) : 0u;
            }
        }
#pragma unroll
        for (int k = 0; k < PP_ITEMS; k++) {
            s += v[k];
            sh += hv[k];
        }
        uint32_t tot, toth;
        const uint32_t local = block_exclusive_scan_n<PP_WAVES>(s, sm.scan_tmp, &tot);
        const uint32_t localh = block_exclusive_scan_n<PP_WAVES>(sh, sm.scan_tmp, &toth);
        uint32_t run = (uint32_t)carry + local, runh = (uint32_t)(carry >> 32) + localh;
#pragma unroll
        for (int k = 0; k < PP_ITEMS; k++) {
            if (base + k < P) {
                a.offsets[base + k] = run;
                a.segoff[base + k] = runh;
            }
            run += v[k];
            runh += hv[k];
        }
        carry += (unsigned long long)tot | ((unsigned long long)toth << 32);
    }
    if (w == G - 1 && threadIdx.x == 0) {  // the last workgroup owns the last tile: its carry holds the totals
        // (the host has had the count since the first barrier; the device copy comes from the same clean 64-bit sum -- the
        // packed carry's low half would wrap silently at 2^32 pairs, which the callers must see as "too many")
        a.offsets[P] = clamp_pair_count(__hip_atomic_load(a.early, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        a.segoff[P] = (uint32_t)(carry >> 32);
    }
    GSR_TS(31);
}

// ======================================================================================= the D-sized chain
constexpr int PS_THREADS = 512, PS_ITEMS = 8, PS_TILE = PS_THREADS * PS_ITEMS, PS_WAVES = PS_THREADS / 64;
constexpr int PS_CHUNK = PS_ITEMS * 64;  // slots of one wave of one tile
#ifndef GSR_PS_DECODE_BATCH
#define GSR_PS_DECODE_BATCH 4
#endif
#ifndef GSR_PS_MIN_WAVES
#define GSR_PS_MIN_WAVES 4
#endif
constexpr int PS_MIN_WAVES = GSR_PS_MIN_WAVES;
constexpr int PS_OWNERS = 256;           // chunk owners kept in LDS: tiles per workgroup <= 32, else per-chunk search
                                         // (with 512 the workgroup would exceed 40 KB: three per CU instead of four)
static_assert(PS_TILE == RADIX_TILE, "tiles of both pipelines have 4096 elements");

struct SortPersistArgs {
    int P, gx, xbits, ybits;
    long long D;    // the pair count, or (bounded) the capacity of the buffers
    int bounded;    // the pair count is offsets[P] on the device; nothing is written when it exceeds D
    const TileRect *rects;
    const uint32_t *sorted_ids;
    const uint32_t *offsets;
    const uint8_t *mask;
    uint32_t *kA, *kB, *vB, *point_list;
    int32_t *ranges;  // [tiles + 1][2]
    int ranges_words;
    const int32_t *hull;
    GridSync sync;
    uint32_t *cnt;  // [2][G][256]
    uint32_t *grp;  // [2][ngroups][256], zero before the launch
    uint32_t *cnt_solo, *grp_solo;  // [2][256] each (grp_solo zero before the launch): workgroup 0 alone after an abort
    uint32_t *done_word;
    uint32_t done_seq;
    uint64_t timeout_ticks;
    int owners_cap;  // <= PS_OWNERS (tests lower it to exercise the per-chunk search)
    int force_abort;  // test hook (GSR_BIN_FORCE_ABORT=s): the first barrier aborts as if it had timed out
    unsigned long long *tstamp;  // diagnostics (GSR_BIN_TIMELINE=1): [G][32] clock stamps of thread 0, else null
};

struct PSExtra {
    int32_t dcol[RADIX_DIGITS + 1];
    int32_t owner[PS_OWNERS];
    uint32_t cflag[PS_WAVES][PS_CHUNK / 32];  // per wave: bit p = a Gaussian's first slot is slot p of the wave's chunk
    int j0, j1;
};

// largest j in [lo, hi] with offsets[j] <= s, given offsets[lo] <= s < offsets[hi]; one wave, 64-ary
__device__ __forceinline__ int owner_search(const uint32_t *__restrict__ offsets, int lo, int hi, uint32_t s) {
    const int lane = threadIdx.x & 63;
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) / 64;
        const int idx = min(lo + lane * step, hi);
        const bool le = offsets[idx] <= s;
        const int c = __popcll(__ballot(le));  // probes are monotone: the first c lanes say "<="
        const int nlo = lo + (c - 1) * step;
        hi = min(hi, nlo + step);
        lo = nlo;
    }
    return lo;
}

// The pairs of one wave's chunk of the emission order -- slots [wbase, wbase + 512), lane l of round r holds slot
// wbase + 64 r + l -- decoded into registers.  The Gaussians are in depth order, Gaussian j owns the slots
// [offsets[j], offsets[j + 1]) (its rect in row-major order), g0 owns slot wbase.  Every Gaussian in front of the culled
// tail owns at least one slot, so the owner of slot wbase + p is g0 + (number of Gaussians after g0 that start at or
// before p): the starts are marked as bits of a 512-bit LDS word set, and a round's owners follow from one broadcast
// read and a population count -- no search, and the three dependent gathers (offset / id, then rect) of ALL rounds
// are in flight together.
template <int BATCH>  // rounds whose gathers are in flight together (8: all of them; 4 halves the registers held)
__device__ __forceinline__ void decode_chunk(const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ sorted_ids,
                                             const TileRect *__restrict__ rects, int P, long long D, long long wbase, int g0,
                                             int xbits, uint32_t *__restrict__ cflag, uint32_t (&key)[PS_ITEMS],
                                             uint32_t (&val)[PS_ITEMS]) {
    const int lane = threadIdx.x & 63;
    const uint32_t cbeg = (uint32_t)wbase;
    const uint32_t cend = (uint32_t)(wbase + PS_CHUNK < D ? wbase + PS_CHUNK : D);
    __builtin_amdgcn_wave_barrier();
    if (lane < PS_CHUNK / 32) cflag[lane] = 0u;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int base = g0 + 1;; base += 64) {  // wave-uniform trip count: at most 9 rounds (512 slots, >= 1 per Gaussian)
        const int j = base + lane;
        const uint32_t o = offsets[j < P ? j : P];  // (offsets[P] = D >= cend: never marked)
        const bool in = o < cend;                   // o > cbeg: g0 owns slot cbeg
        if (in) atomicOr(&cflag[(o - cbeg) >> 5], 1u << ((o - cbeg) & 31u));
        if (__ballot(in) != ~0ull) break;           // offsets are monotone: a lane past the end ends the marking
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int jr[PS_ITEMS];
    int carry = 0;
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
#pragma unroll
    for (int r = 0; r < PS_ITEMS; r++) {
        const unsigned long long m = (unsigned long long)cflag[2 * r] | ((unsigned long long)cflag[2 * r + 1] << 32);
        jr[r] = g0 + carry + __popcll(m & le);
        carry += __popcll(m);
    }
#pragma unroll
    for (int r0 = 0; r0 < PS_ITEMS; r0 += BATCH) {
        uint32_t off[BATCH], gid[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; i++) {
            const bool live = wbase + (r0 + i) * 64 + lane < D;
            const int j = live ? jr[r0 + i] : g0;
            off[i] = offsets[j];
            gid[i] = sorted_ids[j];
        }
        TileRect rc[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; i++) rc[i] = rects[gid[i]];
#pragma unroll
        for (int i = 0; i < BATCH; i++) {
            const int r = r0 + i;
            if (wbase + r * 64 + lane < D) {
                const uint32_t s = cbeg + (uint32_t)(r * 64 + lane);
                const uint32_t minx = rc[i].xs & 0xFFFFu, wd = (rc[i].xs >> 16) - minx, miny = rc[i].ys & 0xFFFFu;
                uint32_t tq = s - off[i];  // the Gaussian's tq-th kept tile: row-major position inside the rect
                if (rc[i].mask != ~0ull) tq = gsr_select_bit(rc[i].mask, tq);
                // tq / wd without the integer-division sequence (~15 VALU): the frame has <= 256 x 256 tiles on this
                // path, so the quotient is < 256 and (tq + 0.5) / wd stays >= 0.5 / 256 away from every integer -- orders
                // of magnitude more than the error of rcp (1 ulp) and the product: the truncation is exact
                const uint32_t q = (uint32_t)(((float)tq + 0.5f) * __builtin_amdgcn_rcpf((float)wd));
                key[r] = ((miny + q) << xbits) | (minx + (tq - q * wd));
                val[r] = gid[i];
            }
        }
    }
}

// between the phases of a workgroup that runs ALONE (below): its own write-through stores acknowledged, then the barrier
__device__ __forceinline__ void solo_sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence();
    __syncthreads();
}

__global__ void __launch_bounds__(PS_THREADS, PS_MIN_WAVES)  // 8: <= 64 VGPRs, four 8-wave workgroups per CU
bin_sort_persist_kernel(const SortPersistArgs a) {
    __shared__ PersistSmem<PS_ITEMS, PS_THREADS> sm;
    __shared__ PSExtra ex;
    uint32_t G = gridDim.x, w = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t d = threadIdx.x;
    const int P = a.P;
    long long D = a.D;
    if (a.bounded) {
        const long long dd = a.offsets[P];
        if (dd > D) {
            // does not fit (or the prepare kernel aborted): every workgroup leaves and no list is written -- but the range
            // table is left EMPTY, not untouched (round 6: the composite kernel is launched before the host has seen the
            // count; it then draws the background, and the caller repeats sort and composite with exact sizes)
            for (int t = w * PS_THREADS + threadIdx.x; t < a.ranges_words; t += G * PS_THREADS)
                reinterpret_cast<uint32_t *>(a.ranges)[t] = 0u;
            if (w == 0 && threadIdx.x < 2) a.ranges[a.ranges_words + threadIdx.x] = a.hull[threadIdx.x];
            if (w == 0 && threadIdx.x == 0 && a.done_seq)
                __hip_atomic_store(a.done_word, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        D = (long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)dd);  // (uniform: the count is one word)
    }
    // Round 6 -- no trap, no host round trip when the grid cannot become resident (another process's barrier kernel, a
    // graph replay or a collective that waits for a late peer holds part of the device): the first barrier's time-out is
    // decided for the WHOLE grid (grid_barrier), nothing but control words has been written by then, every workgroup but
    // number 0 leaves, and workgroup 0 -- which is running, so nothing can keep it from finishing -- sorts the view ALONE
    // through the same phases with G = 1 (workgroup barriers instead of grid barriers, its own count rows).  Slow (tens of
    // milliseconds for millions of pairs) but correct lists, inside the same launch, captured or not; the host sees the
    // recovery counter move (gsr_bin_persist_status) and keeps to the look-back tile sort for a while.
    bool solo = false;
    uint32_t *cnt = a.cnt, *grp = a.grp;
    // This workgroup's slots [s0, s1): an equal share of the 512-slot wave chunks, walked in tiles of eight chunks, the
    // last one partial.  (Sharing out whole 4096-slot tiles left half the grid with one tile more at 6.5 tiles per
    // workgroup; this evens the waiting at the barriers out but not the phase: a partial tile costs a tile's latency
    // chain, so a phase lasts ceil(tiles per workgroup) rounds either way -- measured 203.6 against 201.8 us at c1.)
    const long long nc = (D + PS_CHUNK - 1) / PS_CHUNK;
    long long s0, s1;
    bool have_owners;
    uint32_t epoch = 0;
    GSR_TS(0);
    uint32_t ngroups;
    const uint32_t xmask = (1u << a.xbits) - 1u;

    for (;;) {  // at most two trips: the grid; after an aborted first barrier, workgroup 0 alone
    const long long c0 = (long long)w * nc / G, c1 = (long long)(w + 1) * nc / G;
    s0 = c0 * PS_CHUNK;
    s1 = (c1 * PS_CHUNK < D) ? c1 * PS_CHUNK : D;
    have_owners = (c1 - c0) <= a.owners_cap;
    ngroups = (G + GB_FAN - 1) / GB_FAN;
    // ------------------------------------------------------------------ E0: column counts of my slots, from the rects
    if (threadIdx.x <= RADIX_DIGITS) ex.dcol[threadIdx.x] = 0;
    clear_wtab(sm);
    if (s1 > s0) {  // workgroup-uniform
        if (wave == 0) {
            const int j = owner_search(a.offsets, 0, P, (uint32_t)s0);
            if (lane == 0) ex.j0 = j;
        } else if (wave == 1) {
            const int j = owner_search(a.offsets, 0, P, (uint32_t)(s1 - 1));
            if (lane == 0) ex.j1 = j;
        }
    }
    __syncthreads();
    if (s1 > s0) {
        const int j0 = ex.j0, j1 = ex.j1;
        // (two dependent gathers per Gaussian -- offset / id, then the rect: the next trip's first gather is issued
        // before this trip's rect is used)
        int j = j0 + (int)threadIdx.x;
        uint32_t off_n = 0, end_n = 0, id_n = 0;
        if (j <= j1) { off_n = a.offsets[j]; end_n = a.offsets[j + 1]; id_n = a.sorted_ids[j]; }
        for (; j <= j1; j += PS_THREADS) {
            const uint32_t off = off_n, end = end_n;
            const TileRect rc = a.rects[id_n];
            if (j + PS_THREADS <= j1) {
                off_n = a.offsets[j + PS_THREADS];
                end_n = a.offsets[j + PS_THREADS + 1];
                id_n = a.sorted_ids[j + PS_THREADS];
            }
            if (end <= off) continue;
            const int minx = (int)(rc.xs & 0xFFFFu), maxx = (int)(rc.xs >> 16);
            const int wd = maxx - minx, ht = (int)(rc.ys >> 16) - (int)(rc.ys & 0xFFFFu);
            // the Gaussian's slots [off, end) are its kept tiles in row-major order; mine are [lo, hi) of them
            const long long glo = off > s0 ? (long long)off : s0, ghi = (long long)end < s1 ? (long long)end : s1;
            const int lo = (int)(glo - off), hi = (int)(ghi - off);
            if (rc.mask != ~0ull) {
                // a tile mask: the bits of rank [lo, hi), row by row -- a run of kept tiles adds 1 to its columns
                unsigned long long m = rc.mask;
                if (lo > 0) m &= ~0ull << gsr_select_bit(rc.mask, (uint32_t)lo);
                if (hi < (int)(end - off)) {
                    const uint32_t ph = gsr_select_bit(rc.mask, (uint32_t)(hi - 1));
                    m &= ph >= 63u ? ~0ull : ((2ull << ph) - 1ull);
                }
                const unsigned long long rowm = gsr_full_mask(wd);
                for (int r = 0; r < ht; r++) {
                    unsigned long long bits = (m >> (r * wd)) & rowm;
                    while (bits) {
                        const int a0 = __ffsll((long long)bits) - 1;
                        const unsigned long long rest = ~(bits >> a0);
                        const int len = rest ? __ffsll((long long)rest) - 1 : 64 - a0;
                        atomicAdd(&ex.dcol[minx + a0], 1);
                        atomicAdd(&ex.dcol[minx + a0 + len], -1);
                        bits &= ~(gsr_full_mask(len) << a0);
                    }
                }
            } else if (lo == 0 && hi == (int)(end - off)) {  // the whole rect (all but the first and the last Gaussian)
                atomicAdd(&ex.dcol[minx], ht);
                atomicAdd(&ex.dcol[maxx], -ht);
            } else {
                const int ra = lo / wd, xa = lo - ra * wd, rb = (hi - 1) / wd, xb = (hi - 1) - rb * wd + 1;
                if (ra == rb) {
                    atomicAdd(&ex.dcol[minx + xa], 1);
                    atomicAdd(&ex.dcol[minx + xb], -1);
                } else {
                    atomicAdd(&ex.dcol[minx + xa], 1);  // the first row from xa on, the last row up to xb, full rows between
                    atomicAdd(&ex.dcol[minx], rb - ra);
                    atomicAdd(&ex.dcol[minx + xb], -1);
                    atomicAdd(&ex.dcol[maxx], ra - rb);
                }
            }
            if (have_owners) {  // the chunk starts (multiples of 512 slots) that fall into this Gaussian's slots
                long long c = (glo + PS_CHUNK - 1) / PS_CHUNK * PS_CHUNK;
                for (; c < ghi; c += PS_CHUNK) ex.owner[(int)((c - s0) / PS_CHUNK)] = j;
            }
        }
    }
    __syncthreads();
    {
        const uint32_t v = d < RADIX_DIGITS ? (uint32_t)ex.dcol[d] : 0u;
        uint32_t all;
        const uint32_t c = block_exclusive_scan_n<PS_WAVES>(v, sm.scan_tmp, &all) + v;  // inclusive prefix: the count
        if (d < RADIX_DIGITS) publish_counts(cnt, grp, w, d, c);
    }
    GSR_TS(1);
    if (solo) {
        solo_sync();
        break;
    }
    if (grid_barrier(a.sync, G, epoch, a.timeout_ticks, &sm.flag, a.force_abort != 0)) break;
    // nobody passed and nobody will: the other workgroups (running now, or when they become resident) leave
    if (blockIdx.x != 0) return;
    if (threadIdx.x == 0) __hip_atomic_fetch_add(a.done_word + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    solo = true;
    G = 1;
    w = 0;
    cnt = a.cnt_solo;
    grp = a.grp_solo;
    __syncthreads();
    }
    GSR_TS(2);
    const uint64_t forever = 100000000ull;  // (one second: see bin_prepare_persist_kernel)
    // K7 only writes the tiles that own pairs: clear the range table (every workgroup takes part, AFTER the first barrier:
    // a workgroup that becomes resident after an abort must not touch what workgroup 0 is producing alone; the next
    // barriers order the clears in front of phase T) and set the hull row.  Written through: the same words are written
    // again in phase T by workgroups of other XCDs, and two L2s must not both hold a dirty copy of a line.
    for (int t = w * PS_THREADS + threadIdx.x; t < a.ranges_words; t += G * PS_THREADS)
        st_agent(reinterpret_cast<uint32_t *>(a.ranges) + t, 0u);
    if (w == 0 && threadIdx.x < 2) a.ranges[a.ranges_words + threadIdx.x] = a.hull[threadIdx.x];
    // ------------------------------------------------------------------ E1: decode my slots, scatter by column
    uint32_t key[PS_ITEMS], val[PS_ITEMS];
    {
        uint32_t before, total;
        counts_before<PS_THREADS>(cnt, grp, G, w, sm.skey, before, total);
        uint32_t all;
        uint32_t first = block_exclusive_scan_n<PS_WAVES>(total, sm.scan_tmp, &all) + before;
        for (long long tbase = s0; tbase < s1; tbase += PS_TILE) {
            const long long wbase = tbase + (long long)wave * PS_CHUNK;
#pragma unroll
            for (int r = 0; r < PS_ITEMS; r++) { key[r] = 0xFFFFFFFFu; val[r] = 0u; }
            if (wbase < s1) {  // wave-uniform
                const int g0 = __builtin_amdgcn_readfirstlane(have_owners ? ex.owner[(int)((wbase - s0) / PS_CHUNK)]
                                                                          : owner_search(a.offsets, 0, P, (uint32_t)wbase));
                decode_chunk<GSR_PS_DECODE_BATCH>(a.offsets, a.sorted_ids, a.rects, P, s1, wbase, g0, a.xbits, ex.cflag[wave], key, val);
            }
            count_wave_digits(sm, key, wbase, s1, 0, xmask);
            __syncthreads();
            first += scatter_tile<PS_ITEMS, PS_THREADS, true>(sm, key, val, tbase, s1, 0, a.xbits, first, a.kB, a.vB);
            clear_wtab(sm);
            __syncthreads();
        }
    }
    GSR_TS(3);
    if (solo) solo_sync();
    else if (!grid_barrier(a.sync, G, epoch, forever, &sm.flag)) { barrier_fault(a.done_word, 0x200u + epoch); return; }
    GSR_TS(4);
    // ------------------------------------------------------------------ R0: row-digit counts of my tiles
    {
        const uint32_t ymask = (1u << a.ybits) - 1u;
        uint32_t mytot = 0;
        for (long long tbase = s0; tbase < s1; tbase += PS_TILE) {
            const long long wbase = tbase + (long long)wave * PS_CHUNK;
#pragma unroll
            for (int r = 0; r < PS_ITEMS; r++) {
                const long long j = wbase + r * 64 + lane;
                key[r] = j < s1 ? ld_agent(&a.kB[j]) : 0xFFFFFFFFu;
            }
            count_wave_digits(sm, key, wbase, s1, a.xbits, ymask);
            __syncthreads();
            if (d < RADIX_DIGITS) {
#pragma unroll
                for (int wv = 0; wv < PS_WAVES; wv++) mytot += sm.wtab[wv][d];
            }
            __syncthreads();
            clear_wtab(sm);
            __syncthreads();
        }
        if (d < RADIX_DIGITS)
            publish_counts(cnt + (size_t)G * RADIX_DIGITS, grp + (size_t)ngroups * RADIX_DIGITS, w, d, mytot);
    }
    GSR_TS(5);
    if (solo) solo_sync();
    else if (!grid_barrier(a.sync, G, epoch, forever, &sm.flag)) { barrier_fault(a.done_word, 0x200u + epoch); return; }
    GSR_TS(6);
    // ------------------------------------------------------------------ R1: scatter by row -> (kA, point_list)
    {
        const uint32_t ymask = (1u << a.ybits) - 1u;
        uint32_t before, total;
        counts_before<PS_THREADS>(cnt + (size_t)G * RADIX_DIGITS, grp + (size_t)ngroups * RADIX_DIGITS, G, w, sm.skey,
                                  before, total);
        uint32_t all;
        uint32_t first = block_exclusive_scan_n<PS_WAVES>(total, sm.scan_tmp, &all) + before;
        for (long long tbase = s0; tbase < s1; tbase += PS_TILE) {
            const long long wbase = tbase + (long long)wave * PS_CHUNK;
#pragma unroll
            for (int r = 0; r < PS_ITEMS; r++) {
                const long long j = wbase + r * 64 + lane;
                key[r] = j < s1 ? ld_agent(&a.kB[j]) : 0xFFFFFFFFu;
                val[r] = j < s1 ? ld_agent(&a.vB[j]) : 0u;
            }
            count_wave_digits(sm, key, wbase, s1, a.xbits, ymask);
            __syncthreads();
            first += scatter_tile<PS_ITEMS, PS_THREADS, false>(sm, key, val, tbase, s1, a.xbits, a.ybits, first, a.kA, a.point_list);
            clear_wtab(sm);
            __syncthreads();
        }
    }
    GSR_TS(7);
    if (solo) solo_sync();
    else if (!grid_barrier(a.sync, G, epoch, forever, &sm.flag)) { barrier_fault(a.done_word, 0x200u + epoch); return; }
    GSR_TS(8);
    if (w == 0 && threadIdx.x == 0 && a.done_seq)
        __hip_atomic_store(a.done_word, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // ------------------------------------------------------------------ T: K7 over my slots (see tile_ranges_yx_kernel)
    {
        int2 *const ranges = reinterpret_cast<int2 *>(a.ranges);
        // four keys per thread and trip, with their neighbours; the loads of the NEXT trip are issued before this one's
        // keys are looked at (every agent-scope load is a trip to memory: one after the other they were 15 us)
        auto load6 = [&](long long j, uint32_t (&k)[6]) {  // k[0] = predecessor, k[1..4] = own, k[5] = successor
            k[0] = j > 0 ? ld_agent(&a.kA[j - 1]) : 0xFFFFFFFFu;
            if (j + 4 <= D) {  // (two 8-byte agent-scope loads: the keys were written by other workgroups)
                const unsigned long long q0 = ld_agent64(reinterpret_cast<const unsigned long long *>(a.kA + j));
                const unsigned long long q1 = ld_agent64(reinterpret_cast<const unsigned long long *>(a.kA + j + 2));
                k[1] = (uint32_t)q0; k[2] = (uint32_t)(q0 >> 32); k[3] = (uint32_t)q1; k[4] = (uint32_t)(q1 >> 32);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) k[1 + i] = j + i < D ? ld_agent(&a.kA[j + i]) : 0xFFFFFFFFu;
            }
            k[5] = j + 4 < D ? ld_agent(&a.kA[j + 4]) : 0xFFFFFFFFu;
        };
        long long j = s0 + (long long)threadIdx.x * 4;
        uint32_t k[6], kn[6];
        if (j < s1) load6(j, k);
        while (j < s1) {
            const long long jn = j + PS_THREADS * 4;
            if (jn < s1) load6(jn, kn);
#pragma unroll
            for (int i = 1; i <= 4; i++) {
                if (j + i - 1 >= D) break;
                const uint32_t kk = k[i];
                if (k[i - 1] != kk || k[i + 1] != kk) {
                    const uint32_t tl = (kk >> a.xbits) * (uint32_t)a.gx + (kk & xmask);
                    if (!a.mask[tl]) continue;
                    if (k[i - 1] != kk) st_agent(reinterpret_cast<uint32_t *>(&ranges[tl].x), (uint32_t)(j + i - 1));
                    if (k[i + 1] != kk) st_agent(reinterpret_cast<uint32_t *>(&ranges[tl].y), (uint32_t)(j + i));
                }
            }
#pragma unroll
            for (int i = 0; i < 6; i++) k[i] = kn[i];
            j = jn;
        }
    }
    GSR_TS(9);
}

// ------------------------------------------------------------------------------------------ host side
struct PersistLayoutP {  // inside the control block of the prepare workspace
    size_t sync, grp, cnt, wtot, zero_bytes, total;
};
inline PersistLayoutP persist_layout_p(int G) {
    PersistLayoutP L;
    const int ngroups = (G + GB_FAN - 1) / GB_FAN;
    size_t o = 0;
    L.sync = o; o += align_up(sizeof(uint32_t) * grid_sync_words(G));
    L.grp = o; o += align_up(sizeof(uint32_t) * 4 * (size_t)ngroups * RADIX_DIGITS);
    L.zero_bytes = o;  // [0, zero_bytes) is cleared before every launch
    L.cnt = o; o += align_up(sizeof(uint32_t) * 4 * (size_t)G * RADIX_DIGITS);
    L.wtot = o; o += align_up(sizeof(unsigned long long) * (size_t)G);
    L.total = o;
    return L;
}
struct PersistLayoutS {
    size_t sync, grp, grp_solo, cnt, cnt_solo, zero_bytes, total;
};
inline PersistLayoutS persist_layout_s(int G) {
    PersistLayoutS L;
    const int ngroups = (G + GB_FAN - 1) / GB_FAN;
    size_t o = 0;
    L.sync = o; o += align_up(sizeof(uint32_t) * grid_sync_words(G));
    L.grp = o; o += align_up(sizeof(uint32_t) * 2 * (size_t)ngroups * RADIX_DIGITS);
    L.grp_solo = o; o += align_up(sizeof(uint32_t) * 2 * RADIX_DIGITS);  // (workgroup 0 alone after an aborted first barrier)
    L.zero_bytes = o;
    L.cnt = o; o += align_up(sizeof(uint32_t) * 2 * (size_t)G * RADIX_DIGITS);
    L.cnt_solo = o; o += align_up(sizeof(uint32_t) * 2 * RADIX_DIGITS);
    L.total = o;
    return L;
}
constexpr int PERSIST_MAX_GRID_P = 1024;  // upper bounds used to size workspaces (CUs x workgroups per CU of any part)
constexpr int PERSIST_MAX_GRID_S = 4096;
constexpr long long PERSIST_SORT_MAX_PAIRS = 6ll << 20;  // above: the look-back tile sort (GSR_BIN_PERSIST_MAXD overrides)

// what the device holds at once, per kernel (queried once per device)
struct PersistCaps {
    int grid_p = 0, grid_s = 0;  // 0: unavailable
    bool init = false;
};
std::mutex g_persist_mutex;
PersistCaps g_persist_caps[64];
uint32_t *g_persist_done[64] = {};       // pinned: { sequence number of the last launch past its last barrier, fault code,
                                         // faults (barrier_fault), solo recoveries of the sort kernel }
uint32_t g_persist_faults_seen[64] = {}; // faults already reported to a caller
uint32_t g_persist_solo_seen[64] = {};   // recoveries already answered with a back-off
int g_persist_backoff[64] = {};          // sort calls that still take the look-back tile sort after a recovery
bool g_persist_faulted[64] = {};         // a barrier after the first timed out once: look-back pipeline from then on
uint32_t g_persist_seq[64] = {};         // last sequence number handed out
hipStream_t g_persist_stream[64] = {};   // stream of that launch
bool g_persist_any[64] = {};

enum { PERSIST_OFF = 0, PERSIST_P = 1, PERSIST_S = 2 };
int persist_mode() {  // GSR_BIN_PERSIST = 0 | 1 (default) | p | s : A/B measurements and shared devices
    static const int mode = [] {
        const char *e = getenv("GSR_BIN_PERSIST");
        if (!e || !*e || strcmp(e, "1") == 0) return PERSIST_P | PERSIST_S;
        if (strcmp(e, "p") == 0) return (int)PERSIST_P;
        if (strcmp(e, "s") == 0) return (int)PERSIST_S;
        return (int)PERSIST_OFF;
    }();
    return mode;
}
std::atomic<int> g_persist_override{-1};  // gsr_set_bin_persistent: -1 = environment

// Diagnostics (GSR_BIN_TIMELINE=1): one device buffer per kernel, [grid][32] stamps of the 100 MHz clock taken by thread 0
// of every workgroup at the phase boundaries of the LAST launch; read with gsr_bin_timeline.
unsigned long long *g_timeline[2] = {nullptr, nullptr};
int g_timeline_grid[2] = {0, 0};
unsigned long long *timeline_buffer(int which, int G) {
    static const bool on = [] { const char *e = getenv("GSR_BIN_TIMELINE"); return e && *e == '1'; }();
    if (!on) return nullptr;
    if (!g_timeline[which]) {
        void *p = nullptr;
        if (hipMalloc(&p, sizeof(unsigned long long) * 32 * PERSIST_MAX_GRID_S) != hipSuccess) return nullptr;
        g_timeline[which] = reinterpret_cast<unsigned long long *>(p);
    }
    (void)hipMemset(g_timeline[which], 0, sizeof(unsigned long long) * 32 * (size_t)G);
    g_timeline_grid[which] = G;
    return g_timeline[which];
}

// test hooks (read per call): GSR_BIN_GRID_P / GSR_BIN_GRID_S cap the grids (many tiles per workgroup on small scenes),
// GSR_BIN_OWNERS caps the chunk-owner table
int env_cap(const char *name, int dflt) {
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    const int v = atoi(e);
    return v > 0 && v < dflt ? v : dflt;
}

int persist_caps(int dev, PersistCaps *out) {
    std::lock_guard<std::mutex> guard(g_persist_mutex);
    PersistCaps &c = g_persist_caps[dev];
    if (!c.init) {
        c.init = true;
        int cus = 0, per_p = 0, per_p8 = 0, per_s = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_p, bin_prepare_persist_kernel<4>, PP_THREADS, 0) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_p8, bin_prepare_persist_kernel<8>, PP_THREADS, 0) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_s, bin_sort_persist_kernel, PS_THREADS, 0) == hipSuccess) {
            // one prepare workgroup per CU (16 waves: the latency chain of a tile wants the CU to itself); the sort
            // fills the device (four per CU when the registers allow)
            c.grid_p = (per_p >= 1 && per_p8 >= 1) ? (cus < PERSIST_MAX_GRID_P ? cus : PERSIST_MAX_GRID_P) : 0;
            const long long gs = (long long)cus * (per_s < 4 ? per_s : 4);
            c.grid_s = per_s >= 1 ? (int)(gs < PERSIST_MAX_GRID_S ? gs : PERSIST_MAX_GRID_S) : 0;
        }
        void *p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocDefault) == hipSuccess) {
            memset(p, 0, 64);
            g_persist_done[dev] = reinterpret_cast<uint32_t *>(p);
        } else {
            c.grid_p = c.grid_s = 0;
        }
        (void)hipGetLastError();
    }
    *out = c;
    out->grid_p = env_cap("GSR_BIN_GRID_P", c.grid_p);
    out->grid_s = env_cap("GSR_BIN_GRID_S", c.grid_s);
    return 0;
}

// May a barrier kernel be launched on `stream` now?  Yes when the previous one ran on the same stream (stream order
// serialises them) or has passed its last barrier.  -> sequence number to publish (0 inside a stream capture: a replay
// cannot be tracked -- graphs replay on the stream that owns the device's binning), or -1: use the look-back pipeline.
long long persist_admit(int dev, hipStream_t stream) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return 0;
    (void)hipGetLastError();
    std::lock_guard<std::mutex> guard(g_persist_mutex);
    if (g_persist_faulted[dev]) return -1;
    if (g_persist_any[dev] && g_persist_stream[dev] != stream) {
        // (round 6: the host now has a view's pair count ~20 us into its prepare kernel, so a caller that alternates
        // views between two streams arrives here while the other stream's ~100 us kernel is still running; giving up at
        // once sent every second view down the nine launches of the look-back pipeline -- measured 2140 -> 1410 views/s.
        // A bounded wait for that kernel's last barrier costs the host what the count poll used to cost it.)
        const volatile uint32_t *done = reinterpret_cast<volatile uint32_t *>(g_persist_done[dev]);
        if (*done != g_persist_seq[dev]) {
            const auto t0 = std::chrono::steady_clock::now();
            while (*done != g_persist_seq[dev]) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(250)) return -1;
            }
        }
    }
    uint32_t s = ++g_persist_seq[dev];
    if (s == 0) s = ++g_persist_seq[dev];
    g_persist_stream[dev] = stream;
    g_persist_any[dev] = true;
    return (long long)s;
}

// Has a persistent kernel of this device reported a barrier fault (barrier_fault) that no caller has been told about?
// -> GSR_EFAULT once per fault; the device keeps to the look-back pipeline afterwards.
int persist_fault_check(int dev) {
    if (dev < 0 || dev >= 64 || !g_persist_done[dev]) return 0;
    std::lock_guard<std::mutex> guard(g_persist_mutex);
    const uint32_t faults = reinterpret_cast<volatile uint32_t *>(g_persist_done[dev])[2];
    if (faults == g_persist_faults_seen[dev]) return 0;
    g_persist_faults_seen[dev] = faults;
    g_persist_faulted[dev] = true;
    return GSR_EFAULT;
}

// test hook: GSR_BIN_FORCE_ABORT = p | s | ps (read per call): the named kernels' first barrier aborts
bool force_abort_env(char which) {
    const char *e = getenv("GSR_BIN_FORCE_ABORT");
    return e && strchr(e, which) != nullptr;
}

// The sort kernel finished a view with workgroup 0 alone (the grid could not become resident within the time-out): the
// next calls take the look-back tile sort instead of waiting out the time-out again.
bool persist_sort_backoff(int dev) {
    std::lock_guard<std::mutex> guard(g_persist_mutex);
    const uint32_t solo = reinterpret_cast<volatile uint32_t *>(g_persist_done[dev])[3];
    if (solo != g_persist_solo_seen[dev]) {
        g_persist_solo_seen[dev] = solo;
        g_persist_backoff[dev] = force_abort_env('s') ? 0 : 64;  // (a forced abort is a test: keep the kernel in use)
    }
    if (g_persist_backoff[dev] > 0) {
        g_persist_backoff[dev]--;
        return true;
    }
    return false;
}


}  // namespace
