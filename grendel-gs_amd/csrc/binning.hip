// binning.hip -- K3..K7 for gfx950: which locally computed tiles each Gaussian touches, and the
// per-tile front-to-back lists.
//
// The reference stage list (analyze_statistic.py:1972-1991 of the reference) is a device-wide
// 64-bit (tile<<32 | depth) CUB sort over all D (tile, Gaussian) pairs.  Here the same ORDER is
// produced with far less HBM traffic by splitting the key:
//   1. stable radix sort of the P Gaussians by depth bits (4 x 8-bit passes over P pairs);
//   2. emit the D pairs in that depth order, (ty, tx)-major inside a Gaussian;
//   3. stable radix sort of the D pairs by tile only: keys (row << xbits | column), pass 0 by column, pass 1 by row
//      (frames of <= 256 x 256 tiles; larger frames sort tile ids in ceil(log2(tiles)/8) passes).
// A stable sort by tile of a depth-ordered sequence is exactly the (tile, depth, arrival)
// order of SURVEY.md A.3.  Traffic: ~64 B per Gaussian + ~32 B per pair instead of ~200 B per pair.
//
// At the sizes of one camera (1e6 Gaussians, 1e7 pairs) every kernel of this stage lasts 10-70 us, so
// the stage is bound by the NUMBER of dependent launches as much as by bytes.  The sort is therefore a
// "one sweep" radix sort: the digit histograms of all passes are known BEFORE the keys are sorted -- the depth
// histograms are accumulated by K3 while it writes the keys, and the tile-sort histograms follow from the rects
// alone (a w x h rect adds h to every column digit it spans and w to every row digit) -- and each pass is a single
// kernel that obtains its workgroup's global digit offsets by decoupled look-back over the preceding
// workgroups' per-digit counts (no histogram / scan launches between passes).  Step 2 is fused into the first
// pass of step 3 (emit_scatter_kernel): the pairs are decoded into registers and scattered, never stored
// unsorted.  The offsets scan of K4 is a single-pass look-back scan that gathers its input through the sorted
// order.  Workgroups take their tile from a ticket counter, so a workgroup only ever waits for workgroups that
// are already running (forward progress without co-residency assumptions).
//
// All primitives are hand-written for wave64: ballot-based stable multisplit inside a wave,
// LDS per-wave digit tables, LDS digit-ordered staging for coalesced stores.
#include "common.h"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "radix.h"
#ifndef GSR_EMIT_DECODE_BATCH
#define GSR_EMIT_DECODE_BATCH 2
#endif
#include "binning_persist.h"
#include "binning_rows.h"

namespace {

// ------------------------------------------------------------ K4: offsets = exclusive scan of tt[ids[.]]
// Single pass with decoupled look-back (wave 0 inspects 64 predecessors per round).  out[n] = total.
__global__ void __launch_bounds__(SCAN_THREADS)
scan_gather_lookback_kernel(const uint32_t *__restrict__ src, int packed,
                            const uint32_t *__restrict__ idx, uint32_t *__restrict__ out, uint32_t *__restrict__ out2,
                            long long n, unsigned long long *__restrict__ state, uint32_t *__restrict__ ticket, int nb,
                            uint32_t *__restrict__ host_total, uint32_t seq,
                            const unsigned long long *__restrict__ early) {
    // Two scans in one sweep (round 6): out = exclusive scan of the pairs per Gaussian (-> the offsets of the Gaussian-major
    // emission) and out2 = exclusive scan of the rows of the Gaussian's rect (-> its row segments, binning_rows.h).  K3
    // leaves both in ONE word per Gaussian (`packed`: pairs | rows << 20 on frames of <= 256 x 256 tiles; else the pairs
    // alone): one random gather per element, as before the second scan existed.  Both running sums stay below 2^31
    // (RADIX_MAX_N) and share the 62 value bits of the look-back word: low 31 bits pairs, high 31 bits segments.
    __shared__ uint32_t smem[4];
    __shared__ uint32_t s_bid;
    __shared__ unsigned long long s_excl;
    if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t bid = s_bid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each thread owns SCAN_ITEMS CONSECUTIVE elements so that one workgroup scan suffices
    const long long base = (long long)bid * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], v2[SCAN_ITEMS];
    uint32_t s = 0, s2 = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const uint32_t x = (base + k < n) ? src[idx[base + k]] : 0u;
        v[k] = packed ? (x & TT_MASK) : x;
        v2[k] = packed ? (x >> TT_SHIFT) : 0u;
        s += v[k];
        s2 += v2[k];
    }
    uint32_t tot, tot2;
    const uint32_t local = block_exclusive_scan(s, smem, &tot);
    const uint32_t local2 = block_exclusive_scan(s2, smem, &tot2);
    const unsigned long long tot64 = (unsigned long long)tot | ((unsigned long long)tot2 << 31);
    if (wave == 0) {
        if (lane == 0) st_agent64(&state[bid], tot64 | (bid == 0 ? LB64_PRE : LB64_AGG));
        unsigned long long excl = 0;
        if (bid > 0) {
            long long top = (long long)bid - 1;
            while (true) {
                const long long j = top - lane;
                unsigned long long x = j >= 0 ? ld_agent64(&state[j]) : LB64_PRE;  // virtual zero prefix before 0
                int fp;
                while (true) {
                    const unsigned long long empty = __ballot((x >> 62) == 0ull);
                    const unsigned long long pre = __ballot((x >> 62) >= 2ull);
                    fp = pre ? __ffsll((long long)pre) - 1 : 64;
                    const unsigned long long need = fp >= 63 ? ~0ull : ((2ull << fp) - 1ull);
                    if ((empty & need) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((x >> 62) == 0ull) x = ld_agent64(&state[j]);
                }
                unsigned long long part = lane <= fp ? (x & LB64_VAL) : 0ull;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
                excl += part;
                if (fp < 64) break;
                top -= 64;
            }
        }
        if (lane == 0) {
            if (bid > 0) st_agent64(&state[bid], ((excl + tot64) & LB64_VAL) | LB64_PRE);
            s_excl = excl;
            if ((int)bid == nb - 1) {
                // (the pair count on the device comes from K3's clean 64-bit sum when given: the packed low half would
                // wrap at 2^31 pairs, which the callers must see as "too many"; the host gets it from K3 as well)
                const unsigned long long dsum = early ? *early : ((excl + tot64) & 0x7FFFFFFFull);
                out[n] = clamp_pair_count(dsum);
                out2[n] = (uint32_t)((excl + tot64) >> 31) & 0x7FFFFFFFu;
                if (host_total) {
                    __hip_atomic_store(host_total, clamp_pair_count(dsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(host_total + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
    __syncthreads();
    uint32_t run = (uint32_t)(s_excl & 0x7FFFFFFFull) + local;
    uint32_t run2 = (uint32_t)((s_excl >> 31) & 0x7FFFFFFFull) + local2;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) {
            out[base + k] = run;
            out2[base + k] = run2;
        }
        run += v[k];
        run2 += v2[k];
    }
}

// ---------------------------------------------------------------------------------- K3 and friends
// K3: per Gaussian, the tile rect it can CONTRIBUTE to and the number of its tiles; depth sort keys and
// the digit histograms of the four depth-sort passes.
// The rect is the reference's 3-sigma-radius rect intersected with the
// bounding box of the alpha >= 1/255 ellipse (gsr_alpha_extent) and with the mask's row hull: tiles
// dropped by the intersection cannot receive a contribution under the alpha < 1/255 rule of A.4, so
// the image is unchanged while D (pairs to sort and to walk) shrinks.  Gaussians touching nothing get
// key 0xFFFFFFFF (depths are > 0.2, so real keys are < 0x7F800000) and sort to the end.
// Row hull of the locally computed tiles (computed by every workgroup from the 8-32 KB mask):
// [first tile row with a local tile, one past the last).  Grendel's final mode always passes whole-row
// bands, for which the hull IS the mask; for a general mask the tiles inside the hull that are not
// local are emitted with a sentinel key.
constexpr int TC_THREADS = 1024, TC_BLOCKS = 256;  // 16-wave persistent workgroups, one per CU (measured best)
__global__ void __launch_bounds__(TC_THREADS)
touch_count_kernel(int P, int gx, int gy, const float2 *__restrict__ means2D, const float *__restrict__ depths,
                   const int32_t *__restrict__ radii, const float4 *__restrict__ conic_opacity,
                   const uint8_t *__restrict__ mask, RadixPlan plan, uint32_t *__restrict__ tt,
                   uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, TileRect *__restrict__ rects,
                   uint32_t *__restrict__ ghist, uint32_t *__restrict__ tile_hist, int32_t *__restrict__ hull_out,
                   int cull, unsigned long long *__restrict__ early, uint32_t *__restrict__ host_total, uint32_t seq,
                   uint4 *__restrict__ zero16, size_t zero16_n) {
    // (round 6) the control block of the tile sort that follows on this stream is cleared here instead of by a fill
    // launch between the two steps (gsr_bin_speculative_async)
    for (size_t i = (size_t)blockIdx.x * TC_THREADS + threadIdx.x; i < zero16_n; i += (size_t)gridDim.x * TC_THREADS)
        zero16[i] = make_uint4(0u, 0u, 0u, 0u);
    __shared__ int s_lo, s_hi;
    __shared__ unsigned long long s_nsum[TC_THREADS / 64];
    unsigned long long nsum = 0;  // this thread's share of the pair count D = sum of tiles_touched
    __shared__ uint32_t mh[RADIX_MAX_PASSES][RADIX_DIGITS];
    // digit histograms of the TILE sort (keys (y, x): pass 0 = column, pass 1 = row), known here without looking at
    // a single pair: a rect of w x h tiles adds h to every column digit in [minx, maxx) and w to every row digit in
    // [miny, maxy) -- two +/- entries in a difference array each, prefix-summed once per workgroup
    __shared__ int32_t dxy[3][RADIX_DIGITS + 1];
    __shared__ uint32_t tc_scan[TC_THREADS / 64];
    if (threadIdx.x <= RADIX_DIGITS) dxy[0][threadIdx.x] = dxy[1][threadIdx.x] = dxy[2][threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_lo = gy; s_hi = 0; }
    if (threadIdx.x < RADIX_DIGITS)
        for (int p = 0; p < RADIX_MAX_PASSES; p++) mh[p][threadIdx.x] = 0;
    __syncthreads();
    {  // every thread inspects a contiguous slice of the mask; rows are monotone in the byte index
        const int total = gx * gy;
        const int per = (total + TC_THREADS - 1) / TC_THREADS;
        const int b0 = threadIdx.x * per, b1 = min(b0 + per, total);
        int first = -1, last = -1;
        for (int b = b0; b < b1; b++)
            if (mask[b]) {
                if (first < 0) first = b;
                last = b;
            }
        if (first >= 0) {
            atomicMin(&s_lo, first / gx);
            atomicMax(&s_hi, last / gx + 1);
        }
    }
    __syncthreads();
    const int hull0 = s_lo, hull1 = s_hi;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // K8 / K10 spread THE BAND over the XCDs (gsr_xcd_span_of_block_band)
        hull_out[0] = hull1 > hull0 ? hull0 : 0;
        hull_out[1] = hull1 > hull0 ? hull1 : 0;
    }
    for (long long base = (long long)blockIdx.x * TC_THREADS; base < P; base += (long long)gridDim.x * TC_THREADS) {
        const long long i = base + threadIdx.x;
        const bool valid = i < P;
        uint32_t key = 0xFFFFFFFFu;
        if (valid) {
            uint32_t n = 0;
            TileRect rect{0u, 0u, 0ull};
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
                    miny = max(max(miny, hull0), (int)ceilf((xy.y - ey - (GSR_BLOCK_Y - 1)) * (1.0f / GSR_BLOCK_Y)));
                    maxy = min(min(maxy, hull1), (int)floorf((xy.y + ey) * (1.0f / GSR_BLOCK_Y)) + 1);
                    // (the tile count, the exact tile mask when culling is on, the tile sort's digit histograms)
                    if (maxx > minx && maxy > miny)
                        n = gsr_rect_tiles(xy, co, minx, miny, maxx, maxy, cull != 0, tile_hist ? dxy : nullptr, rect);
                }
            }
            if (n) key = __float_as_uint(depths[i]);
            nsum += n;
            // (with the rows of the rect -- its row segments, binning_rows.h -- in the upper bits on the (row, column) path)
            tt[i] = (tile_hist && n) ? (n | (((rect.ys >> 16) - (rect.ys & 0xFFFFu)) << TT_SHIFT)) : n;
            rects[i] = rect;
            keys[i] = key;
            vals[i] = (uint32_t)i;
        }
        multihist_add(mh, plan, key, valid);
    }
    nsum = wave_sum64(nsum);
    if ((threadIdx.x & 63) == 0) s_nsum[threadIdx.x >> 6] = nsum;
    __syncthreads();
    if (threadIdx.x == 0) {  // the pair count reaches the host HERE, not four sort passes and a scan later
        unsigned long long tot = 0;
        for (int wv = 0; wv < TC_THREADS / 64; wv++) tot += s_nsum[wv];
        publish_pair_count(early, tot, gridDim.x, host_total, seq);
    }
    multihist_flush(mh, plan, ghist);
    if (tile_hist) {  // kernel-uniform
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;  // HW_REG_XCC_ID[3:0]
        for (int p = 0; p < 3; p++) {
            const uint32_t v = threadIdx.x < RADIX_DIGITS ? (uint32_t)dxy[p][threadIdx.x] : 0u;
            uint32_t all;
            const uint32_t c = block_exclusive_scan_n<TC_THREADS / 64>(v, tc_scan, &all) + v;  // inclusive: the count
            if (threadIdx.x < RADIX_DIGITS && c)
                __hip_atomic_fetch_add(&tile_hist[(xcc * RADIX_MAX_PASSES + p) * RADIX_DIGITS + threadIdx.x], c,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// K5: the D output pairs are cut into chunks of EMIT_CHUNK slots, one wave per chunk, so the work is
// balanced by OUTPUT (a near-camera splat covering thousands of tiles no longer serialises one wave).
// The wave finds the Gaussian that owns its first slot with a 64-ary search over the offsets (one
// coalesced probe per round), then walks windows of 64 depth-consecutive Gaussians staged in LDS; every
// slot locates its owner by a 6-step binary search in the window and is written with coalesced stores.
// The digit histograms of the tile-sort passes are accumulated on the way.
constexpr int EMIT_CHUNK = 1024;  // slots per wave (measured: 512-1024 best, 4096 8 % slower)
__global__ void __launch_bounds__(256)
emit_pairs_kernel(int P, long long D, int gx, int tiles, const TileRect *__restrict__ rects,
                  const uint8_t *__restrict__ mask, const uint32_t *__restrict__ sorted_ids,
                  const uint32_t *__restrict__ offsets, RadixPlan plan, uint32_t *__restrict__ keys,
                  uint32_t *__restrict__ vals, uint32_t *__restrict__ ghist) {
    __shared__ uint32_t s_off[4][65];
    __shared__ uint32_t s_g[4][64];
    __shared__ uint2 s_rect[4][64];
    __shared__ uint32_t mh[RADIX_MAX_PASSES][RADIX_DIGITS];
    for (int p = 0; p < RADIX_MAX_PASSES; p++) mh[p][threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long sb = ((long long)blockIdx.x * 4 + wave) * EMIT_CHUNK; sb < D;
         sb += (long long)gridDim.x * 4 * EMIT_CHUNK) {
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
            // (frames above 256 x 256 tiles: K3 keeps every tile of the rect, the mask is not looked at)
            s_rect[wave][lane] = (j < P && end > off) ? make_uint2(rects[g].xs, rects[g].ys) : make_uint2(0u, 0u);
            __builtin_amdgcn_wave_barrier();
            const uint32_t wend = min(__builtin_amdgcn_readlane(end, 63), s_end);
            while (__ballot(s < wend) != 0ull) {  // wave-uniform trip count (the histogram uses ballots)
                const bool valid = s < wend;
                uint32_t key = 0;
                if (valid) {
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
                    key = mask[tile] ? tile : (uint32_t)tiles;  // non-local tile in the hull: sentinel, sorts last
                    keys[s] = key;
                    vals[s] = s_g[wave][a];
                }
                multihist_add(mh, plan, key, valid);
                if (valid) s += 64;
            }
            if (wend >= s_end) break;
            g0 += 64;
        }
    }
    __syncthreads();
    multihist_flush(mh, plan, ghist);
}

// K5 + first pass of K6 in one kernel (frames of <= 256 x 256 tiles).  The tile sort uses keys (row << xbits | column):
// pass 0 sorts by column, pass 1 by row, and both digit histograms are already known (K3).  So the pairs need not
// be written unsorted and read back: the workgroup that owns slots [4096 b, 4096 b + 4096) of the emission order
// decodes them into registers (owner search as in emit_pairs_kernel, one window of 64 depth-consecutive Gaussians
// at a time) and runs the pass-0 scatter on them directly.  All tiles of the rect are emitted, computed locally
// or not: K7 leaves the ranges of tiles that are not computed locally empty, and K8 / K10 never look at them.
template <int ITEMS, int THREADS>
__global__ void __launch_bounds__(THREADS, 8)  // <= 64 VGPRs: four 8-wave workgroups per CU
emit_scatter_kernel(int P, long long D, int xbits, const TileRect *__restrict__ rects,
                    const uint32_t *__restrict__ sorted_ids, const uint32_t *__restrict__ offsets,
                    const uint32_t *__restrict__ ghist, uint32_t *__restrict__ state, uint32_t *__restrict__ ticket,
                    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, bool bounded,
                    int32_t *__restrict__ ranges_flat, int ranges_words, const int32_t *__restrict__ hull) {
    constexpr int WAVES = THREADS / 64;
    __shared__ OnesweepSmem<ITEMS, THREADS> sm;
    // K7 only writes the tiles that own pairs: clear the range table here (it runs two kernels later on this stream)
    // instead of in a memset launch of its own.  Every workgroup takes part, before any of the early exits below.
    for (int t = blockIdx.x * THREADS + threadIdx.x; t < ranges_words; t += gridDim.x * THREADS) ranges_flat[t] = 0;
    if (blockIdx.x == 0 && threadIdx.x < 2) ranges_flat[ranges_words + threadIdx.x] = hull[threadIdx.x];  // row `tiles`
    if (bounded) {  // the grid covers a capacity D; the true pair count is offsets[P] (K4's total)
        const long long d = offsets[P];
        if (d > D) return;
        D = d;
    }
    __shared__ uint32_t cflag[WAVES][ITEMS * 64 / 32];  // decode_chunk's 512 start bits per wave
    const uint32_t bid = onesweep_begin(sm, ticket);
    if ((long long)bid * (ITEMS * THREADS) >= D) return;
    const int wave = threadIdx.x >> 6;
    const long long wbase = (long long)bid * (ITEMS * THREADS) + (long long)wave * (ITEMS * 64);
    uint32_t key[ITEMS], val[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) { key[r] = 0xFFFFFFFFu; val[r] = 0u; }
    if (wbase < D) {  // wave-uniform
        // the Gaussian that owns the chunk's first slot (64-ary search over the offsets), then the flag-word decode of
        // binning_persist.h (round 5: replaces the per-slot binary search in LDS windows of 64 Gaussians -- ~30 VALU
        // instructions per pair in a kernel that is 60-76 % VALU-busy -- by one broadcast read and a population count)
        const int g0 = owner_search(offsets, 0, P, (uint32_t)wbase);
        static_assert(ITEMS == PS_ITEMS, "decode_chunk decodes 8 rounds of 64 slots");
        decode_chunk<GSR_EMIT_DECODE_BATCH>(offsets, sorted_ids, rects, P, D, wbase, g0, xbits, cflag[wave], key, val);
    }
    onesweep_scatter(sm, key, val, bid, D, 0, xbits, ghist, state, keys_out, vals_out);
}

// K7 for (row << xbits | column) keys: four consecutive sorted pairs per thread; tiles that are not computed
// locally keep the empty range
__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
tile_ranges_yx_kernel(long long D, int gx, int xbits, const uint32_t *__restrict__ keys,
                      const uint8_t *__restrict__ mask, int2 *__restrict__ ranges,
                      const uint32_t *__restrict__ D_dev) {
    if (D_dev) {
        const long long d = *D_dev;
        if (d > D) return;
        D = d;
    }
    const long long j = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (j >= D) return;
    uint32_t k[6];  // k[0] = predecessor, k[1..4] = own, k[5] = successor
    k[0] = j > 0 ? keys[j - 1] : 0xFFFFFFFFu;
    if (j + 4 <= D) {
        const uint4 q = *reinterpret_cast<const uint4 *>(keys + j);
        k[1] = q.x; k[2] = q.y; k[3] = q.z; k[4] = q.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) k[1 + i] = j + i < D ? keys[j + i] : 0xFFFFFFFFu;
    }
    k[5] = j + 4 < D ? keys[j + 4] : 0xFFFFFFFFu;
    const uint32_t xmask = (1u << xbits) - 1u;
#pragma unroll
    for (int i = 1; i <= 4; i++) {
        if (j + i - 1 >= D) break;
        const uint32_t kk = k[i];
        if (k[i - 1] != kk || k[i + 1] != kk) {
            const uint32_t t = (kk >> xbits) * (uint32_t)gx + (kk & xmask);
            if (!mask[t]) continue;
            if (k[i - 1] != kk) ranges[t].x = (int)(j + i - 1);
            if (k[i + 1] != kk) ranges[t].y = (int)(j + i);
        }
    }
}

// K7: four consecutive sorted pairs per thread
__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
tile_ranges_kernel(long long D, uint32_t tiles, const uint32_t *__restrict__ tile_of, int2 *__restrict__ ranges) {
    const long long j = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (j >= D) return;
    uint32_t k[6];  // k[0] = predecessor, k[1..4] = own, k[5] = successor
    k[0] = j > 0 ? tile_of[j - 1] : 0xFFFFFFFFu;
    if (j + 4 <= D) {
        const uint4 q = *reinterpret_cast<const uint4 *>(tile_of + j);
        k[1] = q.x; k[2] = q.y; k[3] = q.z; k[4] = q.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) k[1 + i] = j + i < D ? tile_of[j + i] : 0xFFFFFFFFu;
    }
    k[5] = j + 4 < D ? tile_of[j + 4] : 0xFFFFFFFFu;
#pragma unroll
    for (int i = 1; i <= 4; i++) {
        if (j + i - 1 >= D) break;
        const uint32_t t = k[i];
        if (t >= tiles) continue;
        if (k[i - 1] != t) ranges[t].x = (int)(j + i - 1);
        if (k[i + 1] != t) ranges[t].y = (int)(j + i);
    }
}

// Optional (gsr_set_depth_tie_order(1)): order Gaussians of EXACTLY equal depth by their screen position instead of by
// their arrival index.  The reference breaks depth ties by the index in the array the op is given -- at world size
// > 1 that is (source rank, index on the source), gaussian_renderer/__init__.py:624-640 -- so two runs that shard or
// order the same Gaussians differently composite tied pairs in different orders (measured on the bench scene, whose
// generator produces ~N^2 / 3.6e7 exact ties: 1e-4 .. 1e-3 relative on the gradients).  (x, y) of means2D does not depend
// on who computed it.  After the stable depth sort tied Gaussians are adjacent: the first thread of a run (runs are a
// handful of elements; culled Gaussians, key 0xFFFFFFFF, are not a run) insertion-sorts its ids in place.
__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
depth_tie_fixup_kernel(long long n, const uint32_t *__restrict__ keys, uint32_t *__restrict__ ids,
                       const float2 *__restrict__ means2D) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    if (k == 0xFFFFFFFFu || (i > 0 && keys[i - 1] == k) || i + 1 >= n || keys[i + 1] != k) return;
    long long e = i + 2;
    while (e < n && keys[e] == k) e++;
    auto before = [&](uint32_t a, uint32_t b) {
        const float2 pa = means2D[a], pb = means2D[b];
        if (pa.x != pb.x) return pa.x < pb.x;
        if (pa.y != pb.y) return pa.y < pb.y;
        return a < b;
    };
    for (long long a = i + 1; a < e; a++) {
        const uint32_t v = ids[a];
        long long b = a - 1;
        while (b >= i && before(v, ids[b])) {
            ids[b + 1] = ids[b];
            b--;
        }
        ids[b + 1] = v;
    }
}

__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
copy_u32_kernel(long long n, const uint32_t *__restrict__ src, uint32_t *__restrict__ dst) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[j] = src[j];
}

struct PrepLayout {
    size_t tt, kA, vA, kB, vB, offsets, segoff, rects, hull, early, thist, ctrl, total;
    CtrlLayout C;
};
PrepLayout prep_layout(int P, int W, int H) {
    (void)W; (void)H;
    PrepLayout L;
    size_t o = 0;
    const size_t np = align_up((size_t)(P + 1) * 4);
    L.tt = o; o += np;      // pairs of each Gaussian (| the rows of its rect << 20 on the (row, column) path)
    L.segoff = o; o += np;  // exclusive scan of hh[sorted id]: the row segments (round 6, binning_rows.h)
    L.kA = o; o += np;
    L.vA = o; o += np;
    L.kB = o; o += np;
    L.vB = o; o += np;
    L.offsets = o; o += np;
    L.rects = o; o += align_up((size_t)(P + 1) * sizeof(TileRect));
    L.hull = o; o += 256;  // int32 [2]: the row hull of the mask (written by K3, copied into the range table by the sort)
    L.early = o; o += 256;  // { u64 sum of tiles_touched (K3's atomics), u32 workgroups that added theirs }: zeroed with ctrl
    L.thist = o; o += align_up(sizeof(uint32_t) * RADIX_REPLICAS * RADIX_MAX_PASSES * RADIX_DIGITS);  // zeroed with ctrl
    L.ctrl = o;
    L.C = ctrl_layout(P, 4, true);
    o += L.C.total;
    L.total = o;
    return L;
}
// the persistent pipeline's control block shares the space of the look-back pipeline's (one of the two runs per call)
int persist_grid_bound_p(int P) {
    const long long nb = radix_blocks(P);
    return (int)(nb < PERSIST_MAX_GRID_P ? (nb > 0 ? nb : 1) : PERSIST_MAX_GRID_P);
}
int persist_grid_bound_s(int64_t D) {
    const long long nb = radix_blocks(D);
    return (int)(nb < PERSIST_MAX_GRID_S ? (nb > 0 ? nb : 1) : PERSIST_MAX_GRID_S);
}
PrepLayout prep_layout_full(int P, int W, int H) {
    PrepLayout L = prep_layout(P, W, H);
    const size_t need = persist_layout_p(persist_grid_bound_p(P)).total;
    if (need > L.C.total) L.total += need - L.C.total;
    return L;
}
// Pinned, device-mapped result slots (64 per device, two words each): K4's last workgroup stores the pair count and
// then the call's sequence tag; the host POLLS the tag (a few microseconds after the store lands) instead of paying a
// stream synchronise (interrupt + wake-up, ~30-40 us measured) -- everything the host launches after this point is on
// the critical path of the iteration.  If the tag does not show up within ~2 ms the host falls back to
// hipStreamSynchronize (which also surfaces a faulted kernel).  A call's ticket is its sequence number; slot =
// ticket % 64, so up to 64 counts can be outstanding per device (gsr_bin_prepare_async / gsr_bin_count_wait).
constexpr int TOTAL_SLOTS = 64;
std::mutex g_total_mutex;  // slot table set-up and ticket numbering only
uint32_t *g_total_word[64] = {};
uint32_t g_total_seq = 0;
int total_slot(uint32_t **host, uint32_t *seq_out) {
    int dev = 0;
    GSR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return GSR_EINVAL;
    std::lock_guard<std::mutex> guard(g_total_mutex);
    if (!g_total_word[dev]) {
        void *p = nullptr;
        GSR_HIP(hipHostMalloc(&p, sizeof(uint32_t) * 2 * TOTAL_SLOTS, hipHostMallocDefault));
        g_total_word[dev] = reinterpret_cast<uint32_t *>(p);
        memset(p, 0, sizeof(uint32_t) * 2 * TOTAL_SLOTS);
    }
    const uint32_t seq = ++g_total_seq ? g_total_seq : ++g_total_seq;  // never 0
    *seq_out = seq;
    *host = g_total_word[dev] + 2 * (seq % TOTAL_SLOTS);
    return 0;
}
int slot_of_ticket(uint32_t seq, uint32_t **host) {
    int dev = 0;
    GSR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !g_total_word[dev] || seq == 0) return GSR_EINVAL;
    *host = g_total_word[dev] + 2 * (seq % TOTAL_SLOTS);
    return 0;
}
int wait_total(hipStream_t stream, uint32_t *host_total, uint32_t seq, uint32_t *total) {
    volatile uint32_t *w = host_total;
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0;; spins++) {
        if (w[1] == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            *total = w[0];
            return 0;
        }
        if ((spins & 1023) == 1023 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
            break;
    }
    GSR_HIP(hipStreamSynchronize(stream));
    *total = w[0];
    return w[1] == seq ? 0 : GSR_EINVAL;
}
// (row, column) keys + fused emission (emit_scatter_kernel): frames of <= 256 x 256 tiles; GSR_BINNING=generic
// forces the tile-id path (kept for larger frames) for A/B measurements and tests
bool yx_path(int gx, int gy) {
    const char *e = getenv("GSR_BINNING");
    if (e && strcmp(e, "generic") == 0) return false;
    return gx <= RADIX_DIGITS && gy <= RADIX_DIGITS;
}
std::atomic<int> g_tie_order{0};  // 0: arrival index (the reference), 1: screen position
int bits_for(int n) {  // bits needed for the values 0 .. n-1 (at least 1)
    int b = 1;
    while ((1 << b) < n) b++;
    return b;
}
int tile_bits(int tiles) {  // bits of the largest key value, the sentinel `tiles`
    int b = 1;
    while ((1ll << b) <= tiles) b++;
    return b;
}
}  // namespace

// ----------------------------------------------------------------------------------- K3..K7 API
extern "C" size_t gsr_bin_prepare_bytes(int P, int width, int height) {
    if (P < 0 || width <= 0 || height <= 0) return 0;
    return prep_layout_full(P, width, height).total;
}

namespace {
// Exact tile culling (binning_persist.h: gsr_tile_mask).  Its cost is K3's -- per Gaussian and tile row of the rect,
// +27 us per 10^6 Gaussians at 1080p, +37 us at 4K -- its return the D-sized passes' and the composite kernels'.
// Measured (profiles/r05_tile_cull.txt): D -9..19 %; K3-K7 327 -> 330 us at c1, 713 -> 694 us at 4K; the training steps
// 1.206 -> 1.204 ms and 2.811 -> 2.799 ms: the composite kernels skip such pairs with one box test per quadrant anyway.
// Hence OFF by default; "auto" (frames above GSR_TILE_CULL_TILES tiles, default 16384) is a static rule for scenes whose
// splats cover many tiles -- not a function of earlier views.
std::atomic<int> g_tile_cull{-1};  // gsr_set_tile_cull: -1 = environment (GSR_TILE_CULL = 0 | 1 | auto, default 0)
int tile_cull_on(int tiles) {
    int mode = g_tile_cull.load(std::memory_order_relaxed);
    if (mode < 0) {
        static const int env_mode = [] {
            const char *e = getenv("GSR_TILE_CULL");
            if (!e || !*e || strcmp(e, "0") == 0) return 0;
            return strcmp(e, "auto") == 0 ? 2 : 1;
        }();
        mode = env_mode;
    }
    if (mode != 2) return mode;
    static const int min_tiles = [] {
        const char *e = getenv("GSR_TILE_CULL_TILES");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 16384;
    }();
    return tiles > min_tiles ? 1 : 0;
}

// what a gsr_bin_prepare_async call was given: kept per count slot so that gsr_bin_count_wait can repeat the call on
// the look-back pipeline when the persistent kernel gave up at its first barrier (binning_persist.h)
struct PrepCall {
    int P, width, height;
    const float *means2D, *depths;
    const int32_t *radii;
    const float *conic_opacity;
    const uint8_t *compute_locally;
    void *prep;
    hipStream_t stream;
    bool persistent;
    void *zero_ptr;     // the control block of the tile sort launched behind this prepare step (16-byte aligned), or null
    size_t zero_bytes;  // a multiple of 16
};
PrepCall g_prep_calls[64][TOTAL_SLOTS];

int persist_effective_mode() {
    const int o = g_persist_override.load(std::memory_order_relaxed);
    return o >= 0 ? o : persist_mode();
}

// K3-K4 on the look-back pipeline: K3, four one-sweep passes, the offsets scan (six launches)
int prepare_lookback(const PrepCall &c, uint32_t *ticket) {
    const int P = c.P;
    hipStream_t stream = c.stream;
    const PrepLayout L = prep_layout(P, c.width, c.height);
    const int gx = (c.width + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (c.height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    char *base = reinterpret_cast<char *>(c.prep);
    uint32_t *tt = reinterpret_cast<uint32_t *>(base + L.tt);
    uint32_t *kA = reinterpret_cast<uint32_t *>(base + L.kA), *vA = reinterpret_cast<uint32_t *>(base + L.vA);
    uint32_t *kB = reinterpret_cast<uint32_t *>(base + L.kB), *vB = reinterpret_cast<uint32_t *>(base + L.vB);
    uint32_t *offsets = reinterpret_cast<uint32_t *>(base + L.offsets);
    TileRect *rects = reinterpret_cast<TileRect *>(base + L.rects);
    char *ctrl = base + L.ctrl;

    GSR_HIP(hipMemsetAsync(base + L.early, 0, (L.ctrl - L.early) + L.C.total, stream));
    const RadixPlan plan = radix_plan(0, 32);
    uint32_t *host_total = nullptr;
    uint32_t seq = 0;
    int rc = total_slot(&host_total, &seq);
    if (rc) return rc;
    uint32_t *tile_hist = yx_path(gx, gy) ? reinterpret_cast<uint32_t *>(base + L.thist) : nullptr;
    // persistent workgroups (mask hull + LDS tables are per-workgroup set-up)
    const int blocks = gsr_div_up(P, TC_THREADS) < TC_BLOCKS ? gsr_div_up(P, TC_THREADS) : TC_BLOCKS;
    hipLaunchKernelGGL(touch_count_kernel, dim3(blocks), dim3(TC_THREADS), 0, stream, P, gx, gy,
                       reinterpret_cast<const float2 *>(c.means2D), c.depths, c.radii,
                       reinterpret_cast<const float4 *>(c.conic_opacity), c.compute_locally, plan, tt, kA, vA, rects,
                       reinterpret_cast<uint32_t *>(ctrl + L.C.ghist), tile_hist,
                       reinterpret_cast<int32_t *>(base + L.hull), tile_hist ? tile_cull_on(gx * gy) : 0,
                       reinterpret_cast<unsigned long long *>(base + L.early), host_total, seq,
                       reinterpret_cast<uint4 *>(c.zero_ptr), c.zero_bytes / 16);
    int in_first = 1;
    rc = radix_sort_pairs(kA, vA, kB, vB, P, plan, ctrl, L.C, &in_first, stream, nullptr);
    if (rc) return rc;
    // 4 passes -> back in (kA, vA); the sorted ids stay in vA for K5
    uint32_t *sorted_ids = in_first ? vA : vB;
    if (!in_first) {
        hipLaunchKernelGGL(copy_u32_kernel, dim3(gsr_div_up(P, GSR_ONE_DIM_BLOCK)), dim3(GSR_ONE_DIM_BLOCK), 0, stream,
                           (long long)P, vB, vA);
        sorted_ids = vA;
    }
    if (g_tie_order.load(std::memory_order_relaxed) == 1)
        hipLaunchKernelGGL(depth_tie_fixup_kernel, dim3(gsr_div_up(P, GSR_ONE_DIM_BLOCK)), dim3(GSR_ONE_DIM_BLOCK), 0,
                           stream, (long long)P, in_first ? kA : kB, sorted_ids,
                           reinterpret_cast<const float2 *>(c.means2D));
    const int nbs = gsr_div_up(P, SCAN_TILE);
    hipLaunchKernelGGL(scan_gather_lookback_kernel, dim3(nbs), dim3(SCAN_THREADS), 0, stream, tt,
                       tile_hist ? 1 : 0, sorted_ids, offsets,
                       reinterpret_cast<uint32_t *>(base + L.segoff), (long long)P,
                       reinterpret_cast<unsigned long long *>(ctrl + L.C.scan_state),
                       reinterpret_cast<uint32_t *>(ctrl + L.C.tickets), nbs, (uint32_t *)nullptr, seq,
                       reinterpret_cast<const unsigned long long *>(base + L.early));
    GSR_LAUNCH_CHECK();
    *ticket = seq;
    return 0;
}

// K3-K4 as ONE persistent launch (binning_persist.h).  -> 0 launched, 1 not applicable here (use the look-back
// pipeline), otherwise an error
int prepare_persistent(const PrepCall &c, uint32_t *ticket) {
    if (!(persist_effective_mode() & PERSIST_P) || g_tie_order.load(std::memory_order_relaxed) != 0) return 1;
    int dev = 0;
    GSR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return 1;
    PersistCaps caps;
    persist_caps(dev, &caps);
    // tiles of 4096 Gaussians, or of 8192 when that brings the launch down to one tile per workgroup
    const int max_tpw = env_cap("GSR_BIN_MAX_TPW", PP_MAX_TPW + 1) <= PP_MAX_TPW ? env_cap("GSR_BIN_MAX_TPW", PP_MAX_TPW + 1) : 1;
    long long nb = radix_blocks(c.P);
    int items = 4;
    if (nb > caps.grid_p && (nb + 1) / 2 <= caps.grid_p) {
        items = 8;
        nb = (nb + 1) / 2;
    }
    const int G = (int)(nb < caps.grid_p ? nb : caps.grid_p);
    if (G <= 0 || nb > (long long)G * max_tpw) return 1;  // (long sorts are bandwidth-bound: look-back pipeline)
    const long long admit = persist_admit(dev, c.stream);
    if (admit < 0) return 1;
    const int P = c.P;
    const PrepLayout L = prep_layout(P, c.width, c.height);
    const PersistLayoutP PL = persist_layout_p(G);
    const int gx = (c.width + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (c.height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    char *base = reinterpret_cast<char *>(c.prep);
    char *ctrl = base + L.ctrl;
    GSR_HIP(hipMemsetAsync(base + L.early, 0, (L.ctrl - L.early) + PL.zero_bytes, c.stream));
    uint32_t *host_total = nullptr;
    uint32_t seq = 0;
    int rc = total_slot(&host_total, &seq);
    if (rc) return rc;
    PrepPersistArgs a{};
    a.P = P; a.gx = gx; a.gy = gy;
    a.means2D = reinterpret_cast<const float2 *>(c.means2D);
    a.depths = c.depths;
    a.radii = c.radii;
    a.conic_opacity = reinterpret_cast<const float4 *>(c.conic_opacity);
    a.mask = c.compute_locally;
    a.tt = reinterpret_cast<uint32_t *>(base + L.tt);
    a.kA = reinterpret_cast<uint32_t *>(base + L.kA); a.vA = reinterpret_cast<uint32_t *>(base + L.vA);
    a.kB = reinterpret_cast<uint32_t *>(base + L.kB); a.vB = reinterpret_cast<uint32_t *>(base + L.vB);
    a.offsets = reinterpret_cast<uint32_t *>(base + L.offsets);
    a.segoff = reinterpret_cast<uint32_t *>(base + L.segoff);
    a.rects = reinterpret_cast<TileRect *>(base + L.rects);
    a.tile_hist = yx_path(gx, gy) ? reinterpret_cast<uint32_t *>(base + L.thist) : nullptr;
    a.cull = a.tile_hist ? tile_cull_on(gx * gy) : 0;
    a.hull_out = reinterpret_cast<int32_t *>(base + L.hull);
    a.early = reinterpret_cast<unsigned long long *>(base + L.early);
    a.zero16 = reinterpret_cast<uint4 *>(c.zero_ptr);
    a.zero16_n = c.zero_bytes / 16;
    const int ngroups = (G + GB_FAN - 1) / GB_FAN;
    a.sync.leaf = reinterpret_cast<uint32_t *>(ctrl + PL.sync);
    a.sync.root = a.sync.leaf + (size_t)ngroups * GB_LEAF_STRIDE;
    a.sync.flags = a.sync.root + GB_LEAF_STRIDE;
    a.tstamp = timeline_buffer(0, G);
    a.grp = reinterpret_cast<uint32_t *>(ctrl + PL.grp);
    a.cnt = reinterpret_cast<uint32_t *>(ctrl + PL.cnt);
    a.wtot = reinterpret_cast<unsigned long long *>(ctrl + PL.wtot);
    a.host_total = host_total;
    a.seq = seq;
    a.done_word = g_persist_done[dev];
    a.done_seq = (uint32_t)admit;
    a.timeout_ticks = 5000000ull;  // 50 ms of the 100 MHz clock at the first barrier
    a.force_abort = force_abort_env('p') ? 1 : 0;
    if (items == 4) hipLaunchKernelGGL(bin_prepare_persist_kernel<4>, dim3(G), dim3(PP_THREADS), 0, c.stream, a);
    else hipLaunchKernelGGL(bin_prepare_persist_kernel<8>, dim3(G), dim3(PP_THREADS), 0, c.stream, a);
    GSR_LAUNCH_CHECK();
    *ticket = seq;
    return 0;
}
}  // namespace

namespace {
int prepare_async_impl(int P, int width, int height, const float *means2D, const float *depths, const int32_t *radii,
                       const float *conic_opacity, const uint8_t *compute_locally, void *prep, size_t prep_bytes,
                       uint32_t *ticket, gsr_stream_t stream_, void *zero_ptr, size_t zero_bytes);
}
extern "C" int gsr_bin_prepare_async(int P, int width, int height, const float *means2D, const float *depths,
                                     const int32_t *radii, const float *conic_opacity, const uint8_t *compute_locally,
                                     void *prep, size_t prep_bytes, uint32_t *ticket, gsr_stream_t stream_) {
    return prepare_async_impl(P, width, height, means2D, depths, radii, conic_opacity, compute_locally, prep, prep_bytes,
                              ticket, stream_, nullptr, 0);
}
namespace {
int prepare_async_impl(int P, int width, int height, const float *means2D, const float *depths, const int32_t *radii,
                       const float *conic_opacity, const uint8_t *compute_locally, void *prep, size_t prep_bytes,
                       uint32_t *ticket, gsr_stream_t stream_, void *zero_ptr, size_t zero_bytes) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || width <= 0 || height <= 0 || !ticket) return GSR_EINVAL;
    *ticket = 0;  // 0: nothing was launched, the count is 0
    if (P == 0) return 0;
    if (!means2D || !depths || !radii || !conic_opacity || !compute_locally || !prep) return GSR_EINVAL;
    if (P > RADIX_MAX_N) return GSR_EINVAL;
    if (prep_bytes < prep_layout_full(P, width, height).total) return GSR_ENOSPACE;
    const int gx = (width + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    if (gx > 0xFFFF || gy > 0xFFFF) return GSR_EINVAL;
    PrepCall c{P, width, height, means2D, depths, radii, conic_opacity, compute_locally, prep, stream, true,
               zero_ptr, zero_bytes};
    {
        int dev0 = 0;
        GSR_HIP(hipGetDevice(&dev0));
        const int fault = persist_fault_check(dev0);  // (a barrier fault of an earlier call: reported once)
        if (fault) return fault;
    }
    int rc = prepare_persistent(c, ticket);
    if (rc == 1) {
        c.persistent = false;
        rc = prepare_lookback(c, ticket);
    }
    if (rc) return rc;
    int dev = 0;
    GSR_HIP(hipGetDevice(&dev));
    c.zero_ptr = nullptr;  // (a repeat through gsr_bin_count_wait must not clear the control block of a sort in flight)
    c.zero_bytes = 0;
    if (dev >= 0 && dev < 64) g_prep_calls[dev][*ticket % TOTAL_SLOTS] = c;
    return 0;
}
}  // namespace

extern "C" int gsr_set_depth_tie_order(int mode) {
    if (mode != 0 && mode != 1) return GSR_EINVAL;
    g_tie_order.store(mode, std::memory_order_relaxed);
    return 0;
}

extern "C" int gsr_bin_count_wait(uint32_t ticket, int64_t *num_rendered_host, gsr_stream_t stream_) {
    if (!num_rendered_host) return GSR_EINVAL;
    *num_rendered_host = 0;
    if (ticket == 0) return 0;
    uint32_t *host_total = nullptr;
    int rc = slot_of_ticket(ticket, &host_total);
    if (rc) return rc;
    uint32_t total = 0;
    rc = wait_total(reinterpret_cast<hipStream_t>(stream_), host_total, ticket, &total);
    if (rc) return rc;
    if (total == PAIRS_ABORTED) {
        // the persistent kernel gave up at its first barrier (the device is shared with another barrier kernel): the
        // same call again on the look-back pipeline.  A gsr_bin_sort_bounded launched meanwhile has written nothing
        // (it saw a count above any capacity): GSR_ERETRY tells the caller to sort again
        int dev = 0;
        GSR_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64) return GSR_EINVAL;
        PrepCall c = g_prep_calls[dev][ticket % TOTAL_SLOTS];
        if (!c.persistent) return GSR_EINVAL;
        c.persistent = false;
        uint32_t again = 0;
        rc = prepare_lookback(c, &again);
        if (rc) return rc;
        rc = slot_of_ticket(again, &host_total);
        if (rc) return rc;
        rc = wait_total(c.stream, host_total, again, &total);
        if (rc) return rc;
        *num_rendered_host = (int64_t)total;
        return GSR_ERETRY;
    }
    *num_rendered_host = (int64_t)total;
    return 0;
}

extern "C" int gsr_set_tile_cull(int mode) {
    if (mode < -1 || mode > 2) return GSR_EINVAL;
    g_tile_cull.store(mode, std::memory_order_relaxed);
    return 0;
}

extern "C" int gsr_bin_timeline(int which, unsigned long long *out, int max_words, int *grid) {
    if (which < 0 || which > 1 || !out || !grid || max_words < 0) return GSR_EINVAL;
    *grid = g_timeline_grid[which];
    if (!g_timeline[which]) return 0;
    const size_t n = (size_t)g_timeline_grid[which] * 32;
    GSR_HIP(hipDeviceSynchronize());
    GSR_HIP(hipMemcpy(out, g_timeline[which], sizeof(unsigned long long) * (n < (size_t)max_words ? n : (size_t)max_words),
                      hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int gsr_bin_persist_status(uint32_t *out4) {
    if (!out4) return GSR_EINVAL;
    int dev = 0;
    GSR_HIP(hipGetDevice(&dev));
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    if (dev < 0 || dev >= 64 || !g_persist_done[dev]) return 0;
    const volatile uint32_t *w = g_persist_done[dev];
    out4[0] = w[0]; out4[1] = w[1]; out4[2] = w[2]; out4[3] = w[3];
    return 0;
}

extern "C" int gsr_set_bin_persistent(int mode) {
    if (mode < -1 || mode > 3) return GSR_EINVAL;
    g_persist_override.store(mode, std::memory_order_relaxed);
    {  // an explicit choice also ends the back-off that follows a one-workgroup recovery of the sort kernel
        std::lock_guard<std::mutex> guard(g_persist_mutex);
        for (int dev = 0; dev < 64; dev++) {
            if (!g_persist_done[dev]) continue;
            g_persist_solo_seen[dev] = reinterpret_cast<volatile uint32_t *>(g_persist_done[dev])[3];
            g_persist_backoff[dev] = 0;
        }
    }
    return 0;
}

extern "C" int gsr_bin_prepare(int P, int width, int height, const float *means2D, const float *depths,
                               const int32_t *radii, const float *conic_opacity, const uint8_t *compute_locally,
                               void *prep, size_t prep_bytes, int64_t *num_rendered_host, gsr_stream_t stream_) {
    if (!num_rendered_host) return GSR_EINVAL;
    *num_rendered_host = 0;
    uint32_t ticket = 0;
    const int rc = gsr_bin_prepare_async(P, width, height, means2D, depths, radii, conic_opacity, compute_locally, prep,
                                         prep_bytes, &ticket, stream_);
    if (rc) return rc;
    {
        const int rw = gsr_bin_count_wait(ticket, num_rendered_host, stream_);
        return rw == GSR_ERETRY ? 0 : rw;  // (no sort has been launched for this count yet)
    }
}

namespace {
struct SortLayout {
    size_t kA, vA, kB, vB, ctrl, total;
    CtrlLayout C;
};
SortLayout sort_layout(int64_t D, int passes, int gx = 0, int gy = 0) {
    SortLayout L;
    size_t o = 0;
    const size_t nd = align_up((size_t)(D + 1) * 4);
    L.kA = o; o += nd;
    L.vA = o; o += nd;
    L.kB = o; o += nd;
    L.vB = o; o += nd;
    L.ctrl = o;
    L.C = ctrl_layout(D, passes, false);
    const size_t need = persist_layout_s(persist_grid_bound_s(D)).total;  // (either pipeline's control block)
    o += L.C.total > need ? L.C.total : need;
    if (gx > 0 && gy > 0) {  // the row-major pipeline (binning_rows.h) lays the same scratch out its own way
        const size_t rows = rows_layout(D, gx, gy).total;
        if (rows > o) o = rows;
    }
    L.total = o;
    return L;
}

// K5-K7 with one D-sized pass (binning_rows.h): on by default where it applies -- frames of <= 256 x 256 tiles, uncut rects
std::atomic<int> g_rows_override{-1};  // gsr_set_bin_rowmajor: -1 = environment (GSR_BIN_ROWS = 0 | 1, default 1)
// ... from ~5 million pairs on: below, its four launches (three latency chains of one tile each) lose to the persistent sort
// kernel / the two look-back passes -- measured (tools/binbench.py, K5-K7 alone, exact sizes): 1.8 M pairs 71 against 54 us,
// 5.6 M 107 against 101 us, 13.6 M 157 against 188 us (profiles/r06_rows_pipeline.txt).  GSR_BIN_ROWS_MIN overrides
// (tests set 1 to run it on small scenes).
long long rows_min_pairs() {
    const char *e = getenv("GSR_BIN_ROWS_MIN");
    const long long v = e ? atoll(e) : 0;
    return v > 0 ? v : (5ll << 20);
}
bool rows_path(int gx, int gy, long long D = -1) {
    if (D >= 0 && D < rows_min_pairs()) return false;
    int mode = g_rows_override.load(std::memory_order_relaxed);
    if (mode < 0) {
        static const int env_mode = [] {
            const char *e = getenv("GSR_BIN_ROWS");
            return (e && *e == '0') ? 0 : 1;
        }();
        mode = env_mode;
    }
    return mode == 1 && yx_path(gx, gy) && !tile_cull_on(gx * gy);
}
}  // namespace

#ifdef GSR_ROWS_TS
extern "C" int gsr_debug_rows_ts(unsigned long long *out, int words) {
    GSR_HIP(hipDeviceSynchronize());
    GSR_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rows_ts), sizeof(unsigned long long) * (size_t)words));
    return 0;
}
extern "C" int gsr_debug_rows_ts2(unsigned long long *out, int words) {
    GSR_HIP(hipDeviceSynchronize());
    GSR_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rows_ts2), sizeof(unsigned long long) * (size_t)words));
    return 0;
}
#endif

extern "C" int gsr_set_bin_rowmajor(int mode) {
    if (mode < -1 || mode > 1) return GSR_EINVAL;
    g_rows_override.store(mode, std::memory_order_relaxed);
    return 0;
}

extern "C" size_t gsr_bin_sort_bytes(int P, int64_t num_rendered, int width, int height) {
    (void)P;
    if (num_rendered < 0 || width <= 0 || height <= 0) return 0;
    const int gx = (width + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    const int passes = radix_plan(0, tile_bits(gx * gy)).passes;
    // (the (row, column) path always runs 2 passes; the row-major pipeline's layout fits the same buffer)
    return sort_layout(num_rendered, passes < 2 ? 2 : passes, yx_path(gx, gy) ? gx : 0, gy).total;
}

namespace {
// `bounded`: D is a CAPACITY; the kernels read the pair count from the prep workspace (K4's total) and do nothing past it
int bin_sort_impl(int P, int width, int height, const uint8_t *compute_locally, const void *prep, int64_t D,
                  void *scratch, size_t scratch_bytes, uint32_t *point_list, int32_t *ranges, hipStream_t stream,
                  bool bounded, bool ctrl_zeroed = false) {
    if (P < 0 || width <= 0 || height <= 0 || D < 0 || !ranges) return GSR_EINVAL;
    const int gx = (width + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    if (D == 0 || P == 0 || !yx_path(gx, gy))  // (the (row, column) path clears the table inside its first kernel)
        GSR_HIP(hipMemsetAsync(ranges, 0, sizeof(int32_t) * 2 * ((size_t)gx * gy + 1), stream));  // + the hull row: (0, 0)
    if (D == 0 || P == 0) return 0;
    if (!compute_locally || !prep || !scratch || !point_list) return GSR_EINVAL;
    if (D > RADIX_MAX_N) return GSR_EINVAL;
    const RadixPlan plan = radix_plan(0, tile_bits(gx * gy));
    const SortLayout S = sort_layout(D, plan.passes < 2 ? 2 : plan.passes, yx_path(gx, gy) ? gx : 0, gy);
    if (scratch_bytes < S.total) return GSR_ENOSPACE;
    const PrepLayout L = prep_layout(P, width, height);
    const char *pbase = reinterpret_cast<const char *>(prep);
    const uint32_t *sorted_ids = reinterpret_cast<const uint32_t *>(pbase + L.vA);
    const uint32_t *offsets = reinterpret_cast<const uint32_t *>(pbase + L.offsets);
    const TileRect *rects = reinterpret_cast<const TileRect *>(pbase + L.rects);
    char *sbase = reinterpret_cast<char *>(scratch);
    uint32_t *kA = reinterpret_cast<uint32_t *>(sbase + S.kA), *vA = reinterpret_cast<uint32_t *>(sbase + S.vA);
    uint32_t *kB = reinterpret_cast<uint32_t *>(sbase + S.kB), *vB = reinterpret_cast<uint32_t *>(sbase + S.vB);
    char *ctrl = sbase + S.ctrl;

    if (bounded && !yx_path(gx, gy)) return GSR_EINVAL;
    {
        int dev0 = 0;
        GSR_HIP(hipGetDevice(&dev0));
        const int fault = persist_fault_check(dev0);
        if (fault) return fault;
    }
    if (rows_path(gx, gy, D)) {
        // ---- the row-major pipeline (binning_rows.h): segments by row, their scan, ONE pass over the pairs
        const RowsLayout R = rows_layout(D, gx, gy);
        if (scratch_bytes < R.total) return GSR_ENOSPACE;
        if (!ctrl_zeroed) GSR_HIP(hipMemsetAsync(sbase + R.ctrl, 0, R.ctrl_bytes, stream));
        const int xbits = bits_for(gx), ybits = bits_for(gy);
        const uint32_t *thist = reinterpret_cast<const uint32_t *>(pbase + L.thist);
        const uint32_t *segoff = reinterpret_cast<const uint32_t *>(pbase + L.segoff);
        uint32_t *tickets = reinterpret_cast<uint32_t *>(sbase + R.tickets);
        uint32_t *seg_key = reinterpret_cast<uint32_t *>(sbase + R.seg_key);
        uint32_t *seg_gid = reinterpret_cast<uint32_t *>(sbase + R.seg_gid);
        uint32_t *pairoff = reinterpret_cast<uint32_t *>(sbase + R.pairoff);
        int32_t *tdiff = reinterpret_cast<int32_t *>(sbase + R.tdiff);
        // grids: tiles are taken by ticket inside the kernels, so a grid only has to fill the device (the segment count
        // is not known on the host, and the pair count only as a capacity in a bounded launch)
        int dev = 0, cus = 256;
        GSR_HIP(hipGetDevice(&dev));
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        constexpr int TILE_B = 8 * ROWS_THREADS_B, TILE_D = 8 * ROWS_THREADS_D;
        auto tickets_of = [](long long tiles) { return (tiles + TICKET_TILES - 1) / TICKET_TILES; };
        const long long nbB = (D + TILE_B - 1) / TILE_B;
        const int fill_b = cus * 2, fill_d = cus * env_cap("GSR_ROWS_FILL_D", 4);  // (env: A/B measurements only)
        const int grid_b = (int)(tickets_of(nbB) < fill_b ? tickets_of(nbB) : fill_b);
        const long long nbs = (D + SEGSCAN_TILE - 1) / SEGSCAN_TILE;
        // (every workgroup costs a ticket -- an atomic on one word, ~15 ns each when hammered -- whether it finds work or not,
        // and the segment count is typically D / 3..6: a grid of two workgroups per CU loops over the tiles instead)
        const int grid_c = (int)(tickets_of(nbs) < cus * 2 ? tickets_of(nbs) : cus * 2);
        const long long nbt = (D + TILE_D - 1) / TILE_D + gy;
        const int grid_d = (int)(tickets_of(nbt) < fill_d ? tickets_of(nbt) : fill_d);
        uint32_t *tilebase = reinterpret_cast<uint32_t *>(sbase + R.tilebase);
        int32_t *chunk_owner = reinterpret_cast<int32_t *>(sbase + R.chunk_owner);
        hipLaunchKernelGGL((seg_scatter_kernel<8, ROWS_THREADS_B>), dim3(grid_b), dim3(ROWS_THREADS_B), 0, stream, P, (long long)D,
                           ybits, rects, sorted_ids, offsets, segoff, thist + 2 * RADIX_DIGITS,
                           reinterpret_cast<uint32_t *>(sbase + R.seg_state), tickets, seg_key, seg_gid, bounded, ranges,
                           2 * gx * gy, reinterpret_cast<const int32_t *>(pbase + L.hull), tdiff, RADIX_REPLICAS * gy * (gx + 1));
        hipLaunchKernelGGL(seg_scan_kernel, dim3(grid_c), dim3(SCAN_THREADS), 0, stream, P, (long long)D, gx, gy, offsets,
                           segoff, seg_key, pairoff, reinterpret_cast<unsigned long long *>(sbase + R.scan_state),
                           tickets + 1, tdiff, thist, chunk_owner, bounded);
        hipLaunchKernelGGL(tile_base_kernel, dim3(gy), dim3(GSR_ONE_DIM_BLOCK), 0, stream, P, (long long)D, gx, gy, offsets,
                           thist, tdiff, compute_locally, tilebase, reinterpret_cast<int2 *>(ranges), bounded);
        hipLaunchKernelGGL((pair_scatter_kernel<8, ROWS_THREADS_D>), dim3(grid_d), dim3(ROWS_THREADS_D), 0, stream, P,
                           (long long)D, gx, gy, xbits, offsets, segoff, seg_key, seg_gid, pairoff, thist, tilebase,
                           chunk_owner, reinterpret_cast<uint32_t *>(sbase + R.pair_state), tickets + 2, point_list,
                           bounded);
        GSR_LAUNCH_CHECK();
        return 0;
    }
    if (yx_path(gx, gy) && (persist_effective_mode() & PERSIST_S)) {
        // K5-K7 as ONE persistent launch (binning_persist.h) when the device can hold the grid and no barrier kernel of
        // another stream may still be waiting
        int dev = 0;
        GSR_HIP(hipGetDevice(&dev));
        PersistCaps caps;
        const bool dev_ok = dev >= 0 && dev < 64;
        if (dev_ok) persist_caps(dev, &caps);
        const long long nbD = radix_blocks(D);
        const int G = (int)(nbD < caps.grid_s ? nbD : caps.grid_s);
        // long sorts are throughput-bound (VALU: the ranking) and the look-back pipeline, which reads every pair once
        // less, wins: measured cross-over between 2 and 14 million pairs (profiles/r05_binning_persistent.txt)
        const long long max_pairs = env_cap("GSR_BIN_PERSIST_MAXD", 0x7fffffff) == 0x7fffffff ? PERSIST_SORT_MAX_PAIRS
                                                                                        : env_cap("GSR_BIN_PERSIST_MAXD", 0x7fffffff);
        const bool want = dev_ok && G > 0 && g_persist_done[dev] && D <= max_pairs && !persist_sort_backoff(dev);
        const long long admit = want ? persist_admit(dev, stream) : -1;
        if (admit >= 0) {
            const PersistLayoutS PS = persist_layout_s(G);
            if (!ctrl_zeroed) GSR_HIP(hipMemsetAsync(ctrl, 0, PS.zero_bytes, stream));
            SortPersistArgs a{};
            a.P = P; a.gx = gx; a.xbits = bits_for(gx); a.ybits = bits_for(gy);
            a.D = D; a.bounded = bounded ? 1 : 0;
            a.rects = rects; a.sorted_ids = sorted_ids; a.offsets = offsets; a.mask = compute_locally;
            a.kA = kA; a.kB = kB; a.vB = vB; a.point_list = point_list;
            a.ranges = ranges; a.ranges_words = 2 * gx * gy;
            a.hull = reinterpret_cast<const int32_t *>(pbase + L.hull);
            const int ngroups = (G + GB_FAN - 1) / GB_FAN;
            a.sync.leaf = reinterpret_cast<uint32_t *>(ctrl + PS.sync);
            a.sync.root = a.sync.leaf + (size_t)ngroups * GB_LEAF_STRIDE;
            a.sync.flags = a.sync.root + GB_LEAF_STRIDE;
            a.tstamp = timeline_buffer(1, G);
            a.grp = reinterpret_cast<uint32_t *>(ctrl + PS.grp);
            a.cnt = reinterpret_cast<uint32_t *>(ctrl + PS.cnt);
            a.grp_solo = reinterpret_cast<uint32_t *>(ctrl + PS.grp_solo);
            a.cnt_solo = reinterpret_cast<uint32_t *>(ctrl + PS.cnt_solo);
            a.done_word = g_persist_done[dev];
            a.done_seq = (uint32_t)admit;
            // a quarter of a second at the first barrier, then workgroup 0 sorts the view alone (no trap, no host round
            // trip: binning_persist.h); GSR_BIN_SORT_TIMEOUT_MS overrides (tests)
            a.timeout_ticks = 100000ull * (unsigned long long)env_cap("GSR_BIN_SORT_TIMEOUT_MS", 250);
            a.owners_cap = env_cap("GSR_BIN_OWNERS", PS_OWNERS);
            a.force_abort = force_abort_env('s') ? 1 : 0;
            hipLaunchKernelGGL(bin_sort_persist_kernel, dim3(G), dim3(PS_THREADS), 0, stream, a);
            GSR_LAUNCH_CHECK();
            return 0;
        }
    }
    if (!ctrl_zeroed) GSR_HIP(hipMemsetAsync(ctrl, 0, S.C.total, stream));
    if (yx_path(gx, gy)) {
        const int xbits = bits_for(gx), ybits = bits_for(gy);
        const uint32_t *thist = reinterpret_cast<const uint32_t *>(pbase + L.thist);
        const int nb = (int)radix_blocks(D);
        const uint32_t *D_dev = bounded ? offsets + P : nullptr;
        uint32_t *tickets = reinterpret_cast<uint32_t *>(ctrl + S.C.tickets);
        uint32_t *state = reinterpret_cast<uint32_t *>(ctrl + S.C.radix_state);
        // pass 0 (column digit) fused with the emission: pairs land in (kB, vB); pass 1 (row digit) -> (kA, point_list)
        hipLaunchKernelGGL((emit_scatter_kernel<RADIX_TILE / 512, 512>), dim3(nb), dim3(512), 0, stream, P, (long long)D,
                           xbits, rects, sorted_ids, offsets, thist, state, tickets + 1, kB, vB, bounded, ranges,
                           2 * gx * gy, reinterpret_cast<const int32_t *>(pbase + L.hull));
        hipLaunchKernelGGL((radix_onesweep_kernel<RADIX_TILE / 512, 512>), dim3(nb), dim3(512), 0, stream, kB, vB, kA,
                           point_list, (long long)D, xbits, ybits, thist + RADIX_DIGITS,
                           state + (size_t)nb * RADIX_DIGITS, tickets + 2, D_dev);
        hipLaunchKernelGGL(tile_ranges_yx_kernel, dim3(gsr_div_up(gsr_div_up(D, 4), GSR_ONE_DIM_BLOCK)),
                           dim3(GSR_ONE_DIM_BLOCK), 0, stream, (long long)D, gx, xbits, kA, compute_locally,
                           reinterpret_cast<int2 *>(ranges), D_dev);
        GSR_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(gsr_div_up(gsr_div_up(D, EMIT_CHUNK), 4)), dim3(256), 0, stream, P,
                       (long long)D, gx, gx * gy, rects, compute_locally, sorted_ids, offsets, plan, kA, vA,
                       reinterpret_cast<uint32_t *>(ctrl + S.C.ghist));
    int in_first = 1;
    // the last pass writes the Gaussian indices straight into point_list
    int rc = radix_sort_pairs(kA, vA, kB, vB, D, plan, ctrl, S.C, &in_first, stream, point_list);
    if (rc) return rc;
    const uint32_t *ks = in_first ? kA : kB;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(gsr_div_up(gsr_div_up(D, 4), GSR_ONE_DIM_BLOCK)),
                       dim3(GSR_ONE_DIM_BLOCK), 0, stream, (long long)D, (uint32_t)(gx * gy), ks,
                       reinterpret_cast<int2 *>(ranges));
    GSR_HIP(hipMemcpyAsync(ranges + 2 * (size_t)gx * gy, pbase + L.hull, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice,
                           stream));
    GSR_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int gsr_bin_sort(int P, int width, int height, const uint8_t *compute_locally, const void *prep,
                            int64_t D, void *scratch, size_t scratch_bytes, uint32_t *point_list, int32_t *ranges,
                            gsr_stream_t stream_) {
    return bin_sort_impl(P, width, height, compute_locally, prep, D, scratch, scratch_bytes, point_list, ranges,
                         reinterpret_cast<hipStream_t>(stream_), false);
}

extern "C" int gsr_bin_sort_bounded(int P, int width, int height, const uint8_t *compute_locally, const void *prep,
                                    int64_t capacity, void *scratch, size_t scratch_bytes, uint32_t *point_list,
                                    int32_t *ranges, gsr_stream_t stream_) {
    if (capacity <= 0) return GSR_EINVAL;
    return bin_sort_impl(P, width, height, compute_locally, prep, capacity, scratch, scratch_bytes, point_list, ranges,
                         reinterpret_cast<hipStream_t>(stream_), true);
}

// K3-K7 of one view in ONE call for callers that keep a grow-only scratch (the operator's path): gsr_bin_prepare_async,
// then -- when `capacity` > 0 -- gsr_bin_sort_bounded into (scratch, point_list).  Nothing waits: *ticket names the
// pair count for gsr_bin_count_wait, which the caller may call after it has launched the kernels that consume the
// lists (they read the ranges, never the count) -- round 6: the host no longer stands between K3-K7 and K8.
// *sorted = 1 when the bounded sort was launched.
extern "C" int gsr_bin_speculative_async(int P, int width, int height, const float *means2D, const float *depths,
                                         const int32_t *radii, const float *conic_opacity,
                                         const uint8_t *compute_locally, void *prep, size_t prep_bytes,
                                         int64_t capacity, void *scratch, size_t scratch_bytes, uint32_t *point_list,
                                         int32_t *ranges, uint32_t *ticket, int *sorted, gsr_stream_t stream_) {
    if (!ticket || !sorted) return GSR_EINVAL;
    *sorted = 0;
    // the tile sort's control block (either pipeline's: they share the space) is cleared by the prepare step's first
    // kernel instead of by a fill launch between the two steps
    void *zero_ptr = nullptr;
    size_t zero_bytes = 0;
    const bool spec = capacity > 0 && scratch && point_list && P > 0 && width > 0 && height > 0;
    if (spec) {
        const int gx = (width + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
        if (yx_path(gx, gy) && capacity <= RADIX_MAX_N) {
            const RadixPlan plan = radix_plan(0, tile_bits(gx * gy));
            const SortLayout S = sort_layout(capacity, plan.passes < 2 ? 2 : plan.passes, gx, gy);
            if (scratch_bytes >= S.total && rows_path(gx, gy, capacity)) {
                const RowsLayout RL = rows_layout(capacity, gx, gy);
                zero_ptr = reinterpret_cast<char *>(scratch) + RL.ctrl;
                zero_bytes = ((RL.ctrl_bytes + 15) / 16) * 16;
                if (RL.ctrl + zero_bytes > scratch_bytes || (RL.ctrl & 15)) { zero_ptr = nullptr; zero_bytes = 0; }
            } else if (scratch_bytes >= S.total) {
                const size_t zp = persist_layout_s(persist_grid_bound_s(capacity)).zero_bytes;
                zero_ptr = reinterpret_cast<char *>(scratch) + S.ctrl;
                zero_bytes = (((S.C.total > zp ? S.C.total : zp) + 15) / 16) * 16;
                if (S.ctrl + zero_bytes > scratch_bytes || (S.ctrl & 15)) { zero_ptr = nullptr; zero_bytes = 0; }
            }
        }
    }
    int rc = prepare_async_impl(P, width, height, means2D, depths, radii, conic_opacity, compute_locally, prep, prep_bytes,
                                ticket, stream_, zero_ptr, zero_bytes);
    if (rc) return rc;
    if (capacity > 0 && *ticket != 0 && scratch && point_list) {
        rc = bin_sort_impl(P, width, height, compute_locally, prep, capacity, scratch, scratch_bytes, point_list, ranges,
                           reinterpret_cast<hipStream_t>(stream_), true, zero_ptr != nullptr);
        if (rc) return rc;
        *sorted = 1;
    }
    return 0;
}

// The same followed by gsr_bin_count_wait.
// *status = 0: the lists are complete (the count fitted the capacity); 1: the caller must run gsr_bin_sort with buffers for
// *num_rendered_host pairs (no capacity yet, the count outgrew it, or the persistent prepare kernel repeated itself).
extern "C" int gsr_bin_speculative(int P, int width, int height, const float *means2D, const float *depths,
                                   const int32_t *radii, const float *conic_opacity, const uint8_t *compute_locally,
                                   void *prep, size_t prep_bytes, int64_t capacity, void *scratch, size_t scratch_bytes,
                                   uint32_t *point_list, int32_t *ranges, int64_t *num_rendered_host, int *status,
                                   gsr_stream_t stream_) {
    if (!num_rendered_host || !status) return GSR_EINVAL;
    *num_rendered_host = 0;
    *status = 1;
    uint32_t ticket = 0;
    int sorted = 0;
    int rc = gsr_bin_speculative_async(P, width, height, means2D, depths, radii, conic_opacity, compute_locally, prep,
                                       prep_bytes, capacity, scratch, scratch_bytes, point_list, ranges, &ticket, &sorted,
                                       stream_);
    if (rc) return rc;
    rc = gsr_bin_count_wait(ticket, num_rendered_host, stream_);
    if (rc == GSR_ERETRY) return 0;  // (count valid, the bounded sort wrote nothing: status 1)
    if (rc) return rc;
    if (sorted && *num_rendered_host <= capacity) *status = 0;
    return 0;
}

extern "C" size_t gsr_bin_total_offset(int P, int width, int height) {
    if (P < 0 || width <= 0 || height <= 0) return 0;
    return prep_layout(P, width, height).offsets + sizeof(uint32_t) * (size_t)P;  // offsets[P]: K4's total
}

extern "C" size_t gsr_bin_segments_offset(int P, int width, int height) {
    if (P < 0 || width <= 0 || height <= 0) return 0;
    return prep_layout(P, width, height).segoff + sizeof(uint32_t) * (size_t)P;  // segoff[P]: the row segments R
}

extern "C" int64_t gsr_bin_sort_capacity(int P, size_t scratch_bytes, int width, int height) {
    if (width <= 0 || height <= 0) return 0;
    const int gx = (width + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    if (!yx_path(gx, gy)) return 0;  // bounded launches exist for the (row, column) path only
    int64_t lo = 0, hi = RADIX_MAX_N;  // largest D with gsr_bin_sort_bytes(P, D, ...) <= scratch_bytes
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo + 1) / 2;
        if (gsr_bin_sort_bytes(P, mid, width, height) <= scratch_bytes) lo = mid; else hi = mid - 1;
    }
    return lo;
}
