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
// Order of the lists: identical to the look-back pipeline (stable LSD passes over the same keys) -- bit-identical
// point_list / ranges / offsets (tests/test_gpu_parity.py::test_persistent_binning_equals_the_lookback_pipeline).
//
// Co-residency: the grid never exceeds what the device holds at once (hipOccupancyMaxActiveBlocksPerMultiprocessor x
// CUs, checked on the host), so all workgroups arrive at the first barrier as soon as earlier kernels drain.  Two
// barrier kernels on DIFFERENT streams could each hold part of the machine and wait for the rest forever: the host
// admits a persistent launch only when the previous one was on the same stream or has passed its last barrier (a
// pinned word); another process sharing the device is covered by a time-out at the FIRST barrier -- the prepare kernel
// then reports the pair count 0xFFFFFFFF and the host repeats the call on the look-back pipeline; the sort kernel
// traps (fails loudly) after two seconds.
#pragma once

namespace {

constexpr uint32_t GB_ABORT = 0x80000000u;
constexpr int GB_FAN = 32;          // workgroups per leaf counter of the barrier tree and per count aggregate
constexpr int GB_LEAF_STRIDE = 32;  // words between leaf counters (128 bytes: one line each)
constexpr uint32_t PAIRS_ABORTED = 0xFFFFFFFFu;  // "pair count" of a prepare kernel that gave up at its first barrier

struct GridSync {
    uint32_t *leaf;  // [ceil(G / GB_FAN) * GB_LEAF_STRIDE], zero before the launch
    uint32_t *root;  // arrived groups (monotone over the kernel's barriers); bit 31: aborted
};

// Barrier over the G workgroups of the grid; `epoch` counts this workgroup's barriers (uniform over the grid).  Returns
// false when the kernel was aborted (a time-out at some workgroup's barrier): the caller leaves at once.
// Release: every thread's global writes precede the workgroup barrier, thread 0's agent-scope fence writes the XCD's
// L2 back before its arrival becomes visible.  Acquire: thread 0's fence after the spin invalidates the CU's / XCD's
// cached copies before the workgroup barrier lets the other threads read what other XCDs wrote.
__device__ __forceinline__ bool grid_barrier(const GridSync gs, uint32_t G, uint32_t &epoch, uint64_t timeout_ticks,
                                             uint32_t *s_flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch++;
        const uint32_t g = blockIdx.x / GB_FAN, ngroups = (G + GB_FAN - 1) / GB_FAN;
        const uint32_t gsz = min((uint32_t)GB_FAN, G - g * GB_FAN);
        __threadfence();
        const uint32_t old =
            __hip_atomic_fetch_add(&gs.leaf[g * GB_LEAF_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == epoch * gsz)
            __hip_atomic_fetch_add(gs.root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t target = epoch * ngroups;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
        uint32_t v;
        while ((((v = ld_agent(gs.root)) & ~GB_ABORT) < target) && !(v & GB_ABORT)) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
                v = __hip_atomic_fetch_or(gs.root, GB_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | GB_ABORT;
                break;
            }
        }
        __threadfence();
        *s_flag = (v & GB_ABORT) ? 1u : 0u;
    }
    __syncthreads();
    return *s_flag == 0u;
}

// Global offsets of one counting pass from the published counts: cnt[G][256] (row w = workgroup w's digit counts)
// and grp[ngroups][256] (sums over groups of GB_FAN workgroups, accumulated with atomics by the producers).
// Thread d < 256 returns (digit d's pairs in workgroups before w) and its total over the grid.
__device__ __forceinline__ void counts_before(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ grp,
                                              uint32_t G, uint32_t w, uint32_t d, uint32_t &before, uint32_t &total) {
    const uint32_t g = w / GB_FAN, ngroups = (G + GB_FAN - 1) / GB_FAN;
    uint32_t b = 0, t = 0;
    for (uint32_t k = 0; k < ngroups; k++) {
        const uint32_t x = grp[k * RADIX_DIGITS + d];
        t += x;
        if (k < g) b += x;
    }
    for (uint32_t k = g * GB_FAN; k < w; k++) b += cnt[(size_t)k * RADIX_DIGITS + d];
    before = b;
    total = t;
}

__device__ __forceinline__ void publish_counts(uint32_t *__restrict__ cnt, uint32_t *__restrict__ grp, uint32_t w,
                                               uint32_t d, uint32_t c) {
    cnt[(size_t)w * RADIX_DIGITS + d] = c;
    if (c) __hip_atomic_fetch_add(&grp[(w / GB_FAN) * RADIX_DIGITS + d], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// One tile of a scatter phase.  In: per-wave digit counts in sm.wtab (count_wave_digits + a workgroup barrier), the
// pairs in registers; `first[d]` (thread d < 256) = global position of the first pair of digit d that THIS tile
// writes.  Stable: waves in order, rounds in order, lanes in order.  Returns the tile's count of thread d's digit.
// Ends with a workgroup barrier (LDS reusable).
template <int ITEMS, int THREADS>
__device__ __forceinline__ uint32_t scatter_tile(PersistSmem<ITEMS, THREADS> &sm, const uint32_t (&key)[ITEMS],
                                                 const uint32_t (&val)[ITEMS], long long tbase, long long n, int shift,
                                                 int nbits, uint32_t first, uint32_t *__restrict__ keys_out,
                                                 uint32_t *__restrict__ vals_out) {
    constexpr int TILE = ITEMS * THREADS;
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
    if (d < RADIX_DIGITS) {
        sm.gbase[d] = first - run;
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
    __syncthreads();
    const long long rem = n - tbase;
    const int count = rem < TILE ? (int)rem : TILE;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const int i = r * THREADS + threadIdx.x;
        if (i < count) {
            const uint32_t k = sm.skey[i];
            const uint32_t dst = sm.gbase[(k >> shift) & mask] + (uint32_t)i;
            keys_out[dst] = k;
            vals_out[dst] = sm.sval[i];
        }
    }
    __syncthreads();
    return tot;
}

__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ======================================================================================= the P-sized chain
constexpr int PP_THREADS = 1024, PP_ITEMS = 4, PP_TILE = PP_THREADS * PP_ITEMS, PP_WAVES = PP_THREADS / 64;
constexpr int PP_MAX_TPW = 8;  // tiles per workgroup above which the look-back pipeline (bandwidth-bound there) is used
static_assert(PP_TILE == RADIX_TILE, "tiles of both pipelines have 4096 elements");

struct PrepPersistArgs {
    int P, gx, gy;
    const float2 *means2D;
    const float *depths;
    const int32_t *radii;
    const float4 *conic_opacity;
    const uint8_t *mask;
    uint32_t *tt, *kA, *vA, *kB, *vB, *offsets;
    uint2 *rects;
    uint32_t *tile_hist;  // [8 replicas][4][256] or null (frames above 256 x 256 tiles)
    int32_t *hull_out;
    GridSync sync;
    uint32_t *cnt;  // [4][G][256]
    uint32_t *grp;  // [4][ngroups][256], zero before the launch
    unsigned long long *wtot;  // [G]
    uint32_t *host_total;      // pinned: { pair count, sequence tag }
    uint32_t seq;
    uint32_t *done_word;  // pinned: sequence number of the last persistent launch that passed its last barrier
    uint32_t done_seq;    // 0: do not publish (captured launches)
    uint64_t timeout_ticks;
};

struct PPExtra {
    int32_t dxy[2][RADIX_DIGITS + 1];
    int s_lo, s_hi;
};

__global__ void __launch_bounds__(PP_THREADS)
bin_prepare_persist_kernel(const PrepPersistArgs a) {
    __shared__ PersistSmem<PP_ITEMS, PP_THREADS> sm;
    __shared__ PPExtra ex;
    const uint32_t G = gridDim.x, w = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long P = a.P;
    const long long nb = (P + PP_TILE - 1) / PP_TILE;
    const long long t0 = (long long)w * nb / G, t1 = (long long)(w + 1) * nb / G;  // contiguous tiles of this workgroup
    const bool keep = (t1 - t0) == 1;  // one tile: its pairs live in registers from a count phase to its scatter phase
    const uint32_t d = threadIdx.x;
    uint32_t epoch = 0;
    uint32_t key[PP_ITEMS], val[PP_ITEMS];

    // ------------------------------------------------------------------ T: K3 (see touch_count_kernel)
    if (threadIdx.x <= RADIX_DIGITS) ex.dxy[0][threadIdx.x] = ex.dxy[1][threadIdx.x] = 0;
    if (threadIdx.x == 0) { ex.s_lo = a.gy; ex.s_hi = 0; }
    clear_wtab(sm);
    __syncthreads();
    {
        const int total = a.gx * a.gy;
        const int per = (total + PP_THREADS - 1) / PP_THREADS;
        const int b0 = threadIdx.x * per, b1 = min(b0 + per, total);
        int first = -1, last = -1;
        for (int b = b0; b < b1; b++)
            if (a.mask[b]) {
                if (first < 0) first = b;
                last = b;
            }
        if (first >= 0) {
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
    for (long long t = t0; t < t1; t++) {
        const long long wbase = t * PP_TILE + (long long)wave * (PP_ITEMS * 64);
#pragma unroll
        for (int r = 0; r < PP_ITEMS; r++) {
            const long long i = wbase + r * 64 + lane;
            key[r] = 0xFFFFFFFFu;
            val[r] = (uint32_t)i;
            if (i < P) {
                uint32_t n = 0;
                uint2 rect = make_uint2(0u, 0u);
                const int rad = a.radii[i];
                if (rad > 0) {
                    const float2 xy = a.means2D[i];
                    const float4 co = a.conic_opacity[i];
                    float exx, eyy;
                    if (gsr_alpha_extent(co, exx, eyy)) {
                        int minx, miny, maxx, maxy;
                        gsr_get_rect(xy.x, xy.y, rad, a.gx, a.gy, minx, miny, maxx, maxy);
                        minx = max(minx, (int)ceilf((xy.x - exx - (GSR_BLOCK_X - 1)) * (1.0f / GSR_BLOCK_X)));
                        maxx = min(maxx, (int)floorf((xy.x + exx) * (1.0f / GSR_BLOCK_X)) + 1);
                        miny = max(max(miny, hull0), (int)ceilf((xy.y - eyy - (GSR_BLOCK_Y - 1)) * (1.0f / GSR_BLOCK_Y)));
                        maxy = min(min(maxy, hull1), (int)floorf((xy.y + eyy) * (1.0f / GSR_BLOCK_Y)) + 1);
                        if (maxx > minx && maxy > miny) {
                            n = (uint32_t)((maxx - minx) * (maxy - miny));
                            rect = make_uint2((uint32_t)minx | ((uint32_t)maxx << 16),
                                              (uint32_t)miny | ((uint32_t)maxy << 16));
                            if (a.tile_hist) {
                                atomicAdd(&ex.dxy[0][minx], maxy - miny);
                                atomicAdd(&ex.dxy[0][maxx], miny - maxy);
                                atomicAdd(&ex.dxy[1][miny], maxx - minx);
                                atomicAdd(&ex.dxy[1][maxy], minx - maxx);
                            }
                        }
                    }
                }
                if (n) key[r] = __float_as_uint(a.depths[i]);
                a.tt[i] = n;
                a.rects[i] = rect;
                if (!keep) {
                    a.kA[i] = key[r];
                    a.vA[i] = val[r];
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
    if (a.tile_hist) {  // the tile sort's digit histograms, for the look-back tile sort (large D, contended device)
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;  // HW_REG_XCC_ID[3:0]
        for (int p = 0; p < 2; p++) {
            const uint32_t v = threadIdx.x < RADIX_DIGITS ? (uint32_t)ex.dxy[p][threadIdx.x] : 0u;
            uint32_t all;
            const uint32_t c = block_exclusive_scan_n<PP_WAVES>(v, sm.scan_tmp, &all) + v;  // inclusive: the count
            if (threadIdx.x < RADIX_DIGITS && c)
                __hip_atomic_fetch_add(&a.tile_hist[(xcc * RADIX_MAX_PASSES + p) * RADIX_DIGITS + threadIdx.x], c,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (!grid_barrier(a.sync, G, epoch, a.timeout_ticks, &sm.flag)) {
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
    const uint64_t forever = ~0ull >> 1;  // later barriers cannot dead-lock: the whole grid is resident
    // ------------------------------------------------------------------ four LSD passes over the depth bits
    const uint32_t ngroups = (G + GB_FAN - 1) / GB_FAN;
    for (int p = 0; p < 4; p++) {
        const int shift = 8 * p;
        uint32_t *const ksrc = (p & 1) ? a.kB : a.kA, *const vsrc = (p & 1) ? a.vB : a.vA;
        uint32_t *const kdst = (p & 1) ? a.kA : a.kB, *const vdst = (p & 1) ? a.vA : a.vB;
        // B: scatter
        uint32_t before = 0, total = 0;
        if (d < RADIX_DIGITS)
            counts_before(a.cnt + (size_t)p * G * RADIX_DIGITS, a.grp + (size_t)p * ngroups * RADIX_DIGITS, G, w, d,
                          before, total);
        uint32_t all;
        uint32_t first = block_exclusive_scan_n<PP_WAVES>(total, sm.scan_tmp, &all) + before;
        for (long long t = t0; t < t1; t++) {
            const long long tbase = t * PP_TILE;
            const long long wbase = tbase + (long long)wave * (PP_ITEMS * 64);
            if (!keep) {
#pragma unroll
                for (int r = 0; r < PP_ITEMS; r++) {
                    const long long j = wbase + r * 64 + lane;
                    key[r] = j < P ? ksrc[j] : 0xFFFFFFFFu;
                    val[r] = j < P ? vsrc[j] : 0u;
                }
                count_wave_digits(sm, key, wbase, P, shift, 0xFFu);
                __syncthreads();
            }
            first += scatter_tile(sm, key, val, tbase, P, shift, 8, first, kdst, vdst);
            if (!keep) {
                clear_wtab(sm);
                __syncthreads();
            }
        }
        if (!grid_barrier(a.sync, G, epoch, forever, &sm.flag)) return;
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
                key[r] = j < P ? kdst[j] : 0xFFFFFFFFu;
                val[r] = j < P ? vdst[j] : 0u;
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
        if (!grid_barrier(a.sync, G, epoch, forever, &sm.flag)) return;
    }
    // four passes: the sorted (key, id) pairs are back in (kA, vA)
    // ------------------------------------------------------------------ S: offsets = exclusive scan of tt[vA[.]]
    uint32_t v[PP_ITEMS];
    unsigned long long wsum = 0;
    for (long long t = t0; t < t1; t++) {
        const long long base = t * PP_TILE + (long long)threadIdx.x * PP_ITEMS;  // four CONSECUTIVE elements per thread
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < PP_ITEMS; k++) {
            v[k] = (base + k < P) ? a.tt[a.vA[base + k]] : 0u;
            s += v[k];
        }
        wsum += s;
    }
    wsum = wave_sum64(wsum);
    if (lane == 0) sm.scan64[wave] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int wv = 0; wv < PP_WAVES; wv++) tot += sm.scan64[wv];
        a.wtot[w] = tot;
    }
    if (!grid_barrier(a.sync, G, epoch, forever, &sm.flag)) return;
    if (w == 0 && threadIdx.x == 0 && a.done_seq)  // every workgroup is past the last barrier: nothing left to wait for
        __hip_atomic_store(a.done_word, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned long long carry;
    {
        unsigned long long x = 0;
        for (uint32_t k = threadIdx.x; k < w; k += PP_THREADS) x += a.wtot[k];
        x = wave_sum64(x);
        __syncthreads();  // (scan64 was read above)
        if (lane == 0) sm.scan64[wave] = x;
        __syncthreads();
        carry = 0;
        for (int wv = 0; wv < PP_WAVES; wv++) carry += sm.scan64[wv];
    }
    for (long long t = t0; t < t1; t++) {
        const long long base = t * PP_TILE + (long long)threadIdx.x * PP_ITEMS;
        uint32_t s = 0;
        if (!keep) {
#pragma unroll
            for (int k = 0; k < PP_ITEMS; k++) v[k] = (base + k < P) ? a.tt[a.vA[base + k]] : 0u;
        }
#pragma unroll
        for (int k = 0; k < PP_ITEMS; k++) s += v[k];
        uint32_t tot;
        const uint32_t local = block_exclusive_scan_n<PP_WAVES>(s, sm.scan_tmp, &tot);
        uint32_t run = (uint32_t)carry + local;
#pragma unroll
        for (int k = 0; k < PP_ITEMS; k++) {
            if (base + k < P) a.offsets[base + k] = run;
            run += v[k];
        }
        carry += tot;
    }
    if (w == G - 1 && threadIdx.x == 0) {  // the last workgroup owns the last tile: its carry is the pair count
        const uint32_t D = carry >= 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)carry;  // (0xFFFFFFFF is PAIRS_ABORTED)
        a.offsets[P] = D;
        __hip_atomic_store(a.host_total, D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.host_total + 1, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ======================================================================================= the D-sized chain
constexpr int PS_THREADS = 512, PS_ITEMS = 8, PS_TILE = PS_THREADS * PS_ITEMS, PS_WAVES = PS_THREADS / 64;
constexpr int PS_CHUNK = PS_ITEMS * 64;  // slots of one wave of one tile
constexpr int PS_OWNERS = 256;           // chunk owners kept in LDS: tiles per workgroup <= 32, else per-chunk search
                                         // (with 512 the workgroup would exceed 40 KB: three per CU instead of four)
static_assert(PS_TILE == RADIX_TILE, "tiles of both pipelines have 4096 elements");

struct SortPersistArgs {
    int P, gx, xbits, ybits;
    long long D;    // the pair count, or (bounded) the capacity of the buffers
    int bounded;    // the pair count is offsets[P] on the device; nothing is written when it exceeds D
    const uint2 *rects;
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
    uint32_t *done_word;
    uint32_t done_seq;
    uint64_t timeout_ticks;
    int owners_cap;  // <= PS_OWNERS (tests lower it to exercise the per-chunk search)
};

struct PSExtra {
    int32_t dcol[RADIX_DIGITS + 1];
    int32_t owner[PS_OWNERS];
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

__global__ void __launch_bounds__(PS_THREADS, 8)  // <= 64 VGPRs: four 8-wave workgroups per CU
bin_sort_persist_kernel(const SortPersistArgs a) {
    __shared__ PersistSmem<PS_ITEMS, PS_THREADS> sm;
    __shared__ PSExtra ex;
    const uint32_t G = gridDim.x, w = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t d = threadIdx.x;
    const int P = a.P;
    long long D = a.D;
    if (a.bounded) {
        const long long dd = a.offsets[P];
        if (dd > D) {  // does not fit (or the prepare kernel aborted): every workgroup leaves, nothing is written
            if (w == 0 && threadIdx.x == 0 && a.done_seq)
                __hip_atomic_store(a.done_word, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        D = dd;
    }
    // K7 only writes the tiles that own pairs: clear the range table (every workgroup takes part) and set the hull row
    for (int t = w * PS_THREADS + threadIdx.x; t < a.ranges_words; t += G * PS_THREADS) a.ranges[t] = 0;
    if (w == 0 && threadIdx.x < 2) a.ranges[a.ranges_words + threadIdx.x] = a.hull[threadIdx.x];
    const long long nb = (D + PS_TILE - 1) / PS_TILE;
    const long long t0 = (long long)w * nb / G, t1 = (long long)(w + 1) * nb / G;
    const long long s0 = t0 * PS_TILE, s1 = (t1 * PS_TILE < D) ? t1 * PS_TILE : D;  // this workgroup's slots
    const bool have_owners = (t1 - t0) * PS_WAVES <= a.owners_cap;
    uint32_t epoch = 0;
    const uint32_t ngroups = (G + GB_FAN - 1) / GB_FAN;
    const uint32_t xmask = (1u << a.xbits) - 1u;

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
        for (int j = j0 + (int)threadIdx.x; j <= j1; j += PS_THREADS) {
            const uint32_t off = a.offsets[j], end = a.offsets[j + 1];
            if (end <= off) continue;
            const uint2 rc = a.rects[a.sorted_ids[j]];
            const int minx = (int)(rc.x & 0xFFFFu), maxx = (int)(rc.x >> 16);
            const int wd = maxx - minx;
            // the Gaussian's slots [off, end) are its rect in row-major order; mine are [lo, hi) of them
            const long long glo = off > s0 ? (long long)off : s0, ghi = (long long)end < s1 ? (long long)end : s1;
            const int lo = (int)(glo - off), hi = (int)(ghi - off);
            if (lo == 0 && hi == (int)(end - off)) {  // the whole rect (all but the first and the last Gaussian)
                const int h = (int)(rc.y >> 16) - (int)(rc.y & 0xFFFFu);
                atomicAdd(&ex.dcol[minx], h);
                atomicAdd(&ex.dcol[maxx], -h);
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
        if (d < RADIX_DIGITS) publish_counts(a.cnt, a.grp, w, d, c);
    }
    if (!grid_barrier(a.sync, G, epoch, a.timeout_ticks, &sm.flag)) {
        __builtin_trap();  // another barrier kernel (another process?) shares the device: fail loudly (GSR_BIN_PERSIST=0)
        return;
    }
    const uint64_t forever = ~0ull >> 1;
    // ------------------------------------------------------------------ E1: decode my slots, scatter by column
    uint32_t key[PS_ITEMS], val[PS_ITEMS];
    {
        uint32_t before = 0, total = 0;
        if (d < RADIX_DIGITS) counts_before(a.cnt, a.grp, G, w, d, before, total);
        uint32_t all;
        uint32_t first = block_exclusive_scan_n<PS_WAVES>(total, sm.scan_tmp, &all) + before;
        // per-wave window of 64 depth-consecutive Gaussians, in the staging area (written behind workgroup barriers only)
        static_assert(260 * PS_WAVES <= PS_TILE, "windows fit the key staging area");
        uint32_t *const s_off = sm.skey + 260 * wave;                  // [65]
        uint32_t *const s_g = s_off + 66;                              // [64]
        uint2 *const s_rect = reinterpret_cast<uint2 *>(s_off + 130);  // [64], 8-byte aligned
        for (long long t = t0; t < t1; t++) {
            const long long tbase = t * PS_TILE;
            const long long wbase = tbase + (long long)wave * PS_CHUNK;
#pragma unroll
            for (int r = 0; r < PS_ITEMS; r++) { key[r] = 0xFFFFFFFFu; val[r] = 0u; }
            if (wbase < D) {  // wave-uniform
                int g0 = have_owners ? ex.owner[(int)((wbase - s0) / PS_CHUNK)]
                                     : owner_search(a.offsets, 0, P, (uint32_t)wbase);
                uint32_t wend = 0;
                bool have_window = false;
#pragma unroll
                for (int r = 0; r < PS_ITEMS; r++) {
                    const long long sj = wbase + r * 64 + lane;
                    const uint32_t s = (uint32_t)sj;
                    bool pending = sj < D;
                    while (__ballot(pending) != 0ull) {
                        if (!have_window || __ballot(pending && s >= wend) == __ballot(pending)) {
                            if (have_window) g0 += 64;
                            const int j = g0 + lane;
                            const uint32_t off = a.offsets[min(j, P)];
                            const uint32_t end = a.offsets[min(j + 1, P)];
                            const uint32_t g = (j < P && end > off) ? a.sorted_ids[j] : 0u;
                            __builtin_amdgcn_wave_barrier();
                            s_off[lane] = off;
                            if (lane == 63) s_off[64] = end;
                            s_g[lane] = g;
                            s_rect[lane] = (j < P && end > off) ? a.rects[g] : make_uint2(0u, 0u);
                            __builtin_amdgcn_wave_barrier();
                            wend = __builtin_amdgcn_readlane(end, 63);
                            have_window = true;
                        }
                        if (pending && s < wend) {
                            int lo = 0, bnd = 63;
#pragma unroll
                            for (int it = 0; it < 6; it++) {
                                const int mid = (lo + bnd + 1) >> 1;
                                if (s_off[mid] <= s) lo = mid; else bnd = mid - 1;
                            }
                            const uint2 rc = s_rect[lo];
                            const uint32_t tq = s - s_off[lo];
                            const uint32_t minx = rc.x & 0xFFFFu, wd = (rc.x >> 16) - minx, miny = rc.y & 0xFFFFu;
                            // tq / wd by reciprocal: exact for quotients < 256 (see emit_scatter_kernel)
                            const uint32_t q = (uint32_t)(((float)tq + 0.5f) * __builtin_amdgcn_rcpf((float)wd));
                            key[r] = ((miny + q) << a.xbits) | (minx + (tq - q * wd));
                            val[r] = s_g[lo];
                            pending = false;
                        }
                    }
                }
            }
            count_wave_digits(sm, key, wbase, D, 0, xmask);
            __syncthreads();  // (also: every wave has finished decoding -- the windows alias the staging area)
            first += scatter_tile(sm, key, val, tbase, D, 0, a.xbits, first, a.kB, a.vB);
            clear_wtab(sm);
            __syncthreads();
        }
    }
    if (!grid_barrier(a.sync, G, epoch, forever, &sm.flag)) return;
    // ------------------------------------------------------------------ R0: row-digit counts of my tiles
    {
        const uint32_t ymask = (1u << a.ybits) - 1u;
        uint32_t mytot = 0;
        for (long long t = t0; t < t1; t++) {
            const long long wbase = t * PS_TILE + (long long)wave * PS_CHUNK;
#pragma unroll
            for (int r = 0; r < PS_ITEMS; r++) {
                const long long j = wbase + r * 64 + lane;
                key[r] = j < D ? a.kB[j] : 0xFFFFFFFFu;
            }
            count_wave_digits(sm, key, wbase, D, a.xbits, ymask);
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
            publish_counts(a.cnt + (size_t)G * RADIX_DIGITS, a.grp + (size_t)ngroups * RADIX_DIGITS, w, d, mytot);
    }
    if (!grid_barrier(a.sync, G, epoch, forever, &sm.flag)) return;
    // ------------------------------------------------------------------ R1: scatter by row -> (kA, point_list)
    {
        const uint32_t ymask = (1u << a.ybits) - 1u;
        uint32_t before = 0, total = 0;
        if (d < RADIX_DIGITS)
            counts_before(a.cnt + (size_t)G * RADIX_DIGITS, a.grp + (size_t)ngroups * RADIX_DIGITS, G, w, d, before, total);
        uint32_t all;
        uint32_t first = block_exclusive_scan_n<PS_WAVES>(total, sm.scan_tmp, &all) + before;
        for (long long t = t0; t < t1; t++) {
            const long long tbase = t * PS_TILE;
            const long long wbase = tbase + (long long)wave * PS_CHUNK;
#pragma unroll
            for (int r = 0; r < PS_ITEMS; r++) {
                const long long j = wbase + r * 64 + lane;
                key[r] = j < D ? a.kB[j] : 0xFFFFFFFFu;
                val[r] = j < D ? a.vB[j] : 0u;
            }
            count_wave_digits(sm, key, wbase, D, a.xbits, ymask);
            __syncthreads();
            first += scatter_tile(sm, key, val, tbase, D, a.xbits, a.ybits, first, a.kA, a.point_list);
            clear_wtab(sm);
            __syncthreads();
        }
    }
    if (!grid_barrier(a.sync, G, epoch, forever, &sm.flag)) return;
    if (w == 0 && threadIdx.x == 0 && a.done_seq)
        __hip_atomic_store(a.done_word, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // ------------------------------------------------------------------ T: K7 over my slots (see tile_ranges_yx_kernel)
    {
        int2 *const ranges = reinterpret_cast<int2 *>(a.ranges);
        for (long long j = s0 + (long long)threadIdx.x * 4; j < s1; j += PS_THREADS * 4) {
            uint32_t k[6];  // k[0] = predecessor, k[1..4] = own, k[5] = successor
            k[0] = j > 0 ? a.kA[j - 1] : 0xFFFFFFFFu;
            if (j + 4 <= D) {
                const uint4 q = *reinterpret_cast<const uint4 *>(a.kA + j);
                k[1] = q.x; k[2] = q.y; k[3] = q.z; k[4] = q.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) k[1 + i] = j + i < D ? a.kA[j + i] : 0xFFFFFFFFu;
            }
            k[5] = j + 4 < D ? a.kA[j + 4] : 0xFFFFFFFFu;
#pragma unroll
            for (int i = 1; i <= 4; i++) {
                if (j + i - 1 >= D) break;
                const uint32_t kk = k[i];
                if (k[i - 1] != kk || k[i + 1] != kk) {
                    const uint32_t tl = (kk >> a.xbits) * (uint32_t)a.gx + (kk & xmask);
                    if (!a.mask[tl]) continue;
                    if (k[i - 1] != kk) ranges[tl].x = (int)(j + i - 1);
                    if (k[i + 1] != kk) ranges[tl].y = (int)(j + i);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
struct PersistLayoutP {  // inside the control block of the prepare workspace
    size_t sync, grp, cnt, wtot, zero_bytes, total;
};
inline PersistLayoutP persist_layout_p(int G) {
    PersistLayoutP L;
    const int ngroups = (G + GB_FAN - 1) / GB_FAN;
    size_t o = 0;
    L.sync = o; o += align_up(sizeof(uint32_t) * ((size_t)ngroups * GB_LEAF_STRIDE + GB_LEAF_STRIDE));
    L.grp = o; o += align_up(sizeof(uint32_t) * 4 * (size_t)ngroups * RADIX_DIGITS);
    L.zero_bytes = o;  // [0, zero_bytes) is cleared before every launch
    L.cnt = o; o += align_up(sizeof(uint32_t) * 4 * (size_t)G * RADIX_DIGITS);
    L.wtot = o; o += align_up(sizeof(unsigned long long) * (size_t)G);
    L.total = o;
    return L;
}
struct PersistLayoutS {
    size_t sync, grp, cnt, zero_bytes, total;
};
inline PersistLayoutS persist_layout_s(int G) {
    PersistLayoutS L;
    const int ngroups = (G + GB_FAN - 1) / GB_FAN;
    size_t o = 0;
    L.sync = o; o += align_up(sizeof(uint32_t) * ((size_t)ngroups * GB_LEAF_STRIDE + GB_LEAF_STRIDE));
    L.grp = o; o += align_up(sizeof(uint32_t) * 2 * (size_t)ngroups * RADIX_DIGITS);
    L.zero_bytes = o;
    L.cnt = o; o += align_up(sizeof(uint32_t) * 2 * (size_t)G * RADIX_DIGITS);
    L.total = o;
    return L;
}
constexpr int PERSIST_MAX_GRID_P = 1024;  // upper bounds used to size workspaces (CUs x workgroups per CU of any part)
constexpr int PERSIST_MAX_GRID_S = 4096;

// what the device holds at once, per kernel (queried once per device)
struct PersistCaps {
    int grid_p = 0, grid_s = 0;  // 0: unavailable
    bool init = false;
};
std::mutex g_persist_mutex;
PersistCaps g_persist_caps[64];
uint32_t *g_persist_done[64] = {};       // pinned: sequence number of the last launch past its last barrier
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
        int cus = 0, per_p = 0, per_s = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_p, bin_prepare_persist_kernel, PP_THREADS, 0) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_s, bin_sort_persist_kernel, PS_THREADS, 0) == hipSuccess) {
            // one prepare workgroup per CU (16 waves: the latency chain of a tile wants the CU to itself); the sort
            // fills the device (four per CU when the registers allow)
            c.grid_p = per_p >= 1 ? (cus < PERSIST_MAX_GRID_P ? cus : PERSIST_MAX_GRID_P) : 0;
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
    if (g_persist_any[dev] && g_persist_stream[dev] != stream) {
        const uint32_t done = *reinterpret_cast<volatile uint32_t *>(g_persist_done[dev]);
        if (done != g_persist_seq[dev]) return -1;
    }
    uint32_t s = ++g_persist_seq[dev];
    if (s == 0) s = ++g_persist_seq[dev];
    g_persist_stream[dev] = stream;
    g_persist_any[dev] = true;
    return (long long)s;
}

}  // namespace
