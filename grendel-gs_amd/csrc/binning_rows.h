// binning_rows.h -- K5..K7 with ONE pass over the D (tile, Gaussian) pairs (round 6).  Included by binning.hip.
//
// The split-key pipeline of rounds 1-5 sorts the D pairs, emitted Gaussian by Gaussian in depth order, twice: stably by
// column, then stably by row (emit_scatter_kernel + radix_onesweep_kernel + tile_ranges_yx_kernel, or the persistent
// kernel's E1 / R0 / R1 / T phases).  Both passes are bound by the ranking's VALU work, not by bytes: 0.28-0.33 of the HBM
// peak at every size -- 61 % of the step at configs[4]'s shape (1.4e9 pairs), 36 % at configs[2]'s on one GPU.
//
// A Gaussian's rect is h ROW SEGMENTS of w tiles each.  Sorting the R = sum(h) ~ D / 3..6 SEGMENTS stably by tile row
// leaves, per row, the Gaussians that reach it in depth order; emitting the pairs from THAT order (row-major) makes the
// second D-sized pass unnecessary -- a stable scatter by column inside a row already yields (row, column, depth, arrival)
// order, i.e. the lists of SURVEY.md A.3, bit for bit:
//
//   seg_scatter_kernel   R-sized.  The segments are decoded from the depth order (segoff = exclusive scan of the rects'
//                        heights, written by the prepare step beside the pair offsets) and scattered by row in one
//                        one-sweep pass; a segment travels as one word (row, minx, maxx) + the Gaussian's index.
//   seg_scan_kernel      R-sized.  pairoff = exclusive scan of the segments' widths in the sorted order (the position of
//                        a segment's first pair in the row-major emission); the per-tile pair counts as per-row
//                        difference arrays (+1 at minx, -1 at maxx), pre-aggregated in LDS, one replica per XCD; the
//                        segment that owns the first pair of every 512-pair chunk.
//   tile_base_kernel     one workgroup per tile row: the start of every (row, column) list = the row's pair prefix + the
//                        prefix of the row's tile counts, and the ranges -- no K7 pass over sorted keys, no key array.
//   pair_scatter_kernel  D-sized, the only one.  Tiles of 4096 pairs never straddle a tile row; a tile decodes its pairs
//                        (flag-word decode over pairoff, as emit_scatter_kernel does over the Gaussian-major offsets),
//                        ranks them by column and writes ONLY the Gaussian index, straight into point_list; the
//                        look-back runs inside the row.
//
// Pair traffic: 4 B written per pair (was 8 + 8 + 8 + 4 read / written over two passes and K7); the R-sized steps move
// ~12 B per segment.  Lists, ranges and counts are identical to both older pipelines
// (tests/test_gpu_parity.py::test_persistent_binning_equals_the_lookback_pipeline compares all three).
//
// Frames of <= 256 x 256 tiles with uncut rects (gsr_set_tile_cull off): the exact tile masks of round 5 give a segment a
// span of its own per row and keep to the two-pass pipelines.  Stands where the reference runs duplicateWithKeys, its
// 64-bit SortPairs and identifyTileRanges (analyze_statistic.py:1972-1991, stages 40 / 50 / 60).
#pragma once

namespace {

#ifdef GSR_ROWS_TS  // diagnostics build (tools/rows_timeline.py): per-tile phase stamps of the last pair_scatter launch
__device__ unsigned long long g_rows_ts[16384 * 8];
#define ROWS_TS(k)                                                                                    \
    do {                                                                                              \
        if (threadIdx.x == 0 && bid < 16384u) g_rows_ts[bid * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
__device__ unsigned long long g_rows_ts2[16384 * 8];  // the same for seg_scatter_kernel (3 stamps) / seg_scan_kernel (3)
#define ROWS_TS2(base, k)                                                                                    \
    do {                                                                                                     \
        if (threadIdx.x == 0 && bid < 4096u) g_rows_ts2[bid * 8 + (base) + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define ROWS_TS(k) do {} while (0)
#define ROWS_TS2(base, k) do {} while (0)
#endif

constexpr int SEG_ROW_SHIFT = 18;  // segment word: row << 18 | minx << 9 | maxx  (minx < 256, maxx <= 256, row < 256)
constexpr int ROWS_LDS = 4;        // tile rows whose count differences a scan workgroup keeps in LDS (its 8192 sorted
                                   // segments rarely span more; the rest go to global atomics directly)

// Tiles are handed out by ticket (a workgroup only ever waits for tiles that running workgroups own).  A ticket may grant
// TICKET_TILES consecutive tiles (an experiment: fewer atomics on the one ticket word) -- measured a disaster: a workgroup
// walks its tiles one after the other while the next workgroup's first tile looks back at the last of them, so the
// decoupled look-back degenerates into a serial chain (c1: binning 0.31 -> 8.6 / 16.6 / 22.6 ms at 2 / 4 / 8 tiles per
// ticket, profiles/r06_rows_pipeline.txt).  One tile per ticket.
#ifndef GSR_TICKET_TILES
#define GSR_TICKET_TILES 1
#endif
constexpr uint32_t TICKET_TILES = GSR_TICKET_TILES;
template <int ITEMS, int THREADS>
__device__ __forceinline__ uint32_t rows_next_tile(OnesweepSmem<ITEMS, THREADS> &sm, uint32_t *__restrict__ ticket,
                                                   uint32_t &next, uint32_t &end) {
    const bool need = next == end;  // (uniform)
    if (need && threadIdx.x == 0) sm.bid = atomicAdd(ticket, 1u);
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++)
        if (threadIdx.x < RADIX_DIGITS) sm.wtab[w][threadIdx.x] = 0;
    __syncthreads();
    if (need) {
        next = sm.bid * TICKET_TILES;
        end = next + TICKET_TILES;
    }
    return next++;
}

__device__ __forceinline__ uint32_t seg_minx(uint32_t k) { return (k >> 9) & 511u; }
__device__ __forceinline__ uint32_t seg_maxx(uint32_t k) { return k & 511u; }
__device__ __forceinline__ uint32_t seg_row(uint32_t k) { return k >> SEG_ROW_SHIFT; }

// Per-row tables every workgroup derives from K3's row histograms (8 XCD replicas each): pairs per row -> the row's pair
// prefix and its tiles of 4096 pairs (tiles never straddle rows).  THREADS >= 256; thread r < 256 owns row r.
struct RowTables {
    uint32_t pair_prefix[RADIX_DIGITS + 1];  // pairs in the rows before r; [gy] = D
    uint32_t tile_start[RADIX_DIGITS + 1];   // tiles in the rows before r; [gy] = all tiles
};
template <int WAVES, int TILE>
__device__ __forceinline__ void row_tables(RowTables &rt, const uint32_t *__restrict__ thist, int gy, uint32_t *scan_tmp) {
    const uint32_t r = threadIdx.x;
    uint32_t d = 0;
    if (r < (uint32_t)gy)
        for (int x = 0; x < RADIX_REPLICAS; x++) d += thist[((size_t)x * RADIX_MAX_PASSES + 1) * RADIX_DIGITS + r];
    uint32_t all;
    const uint32_t pp = block_exclusive_scan_n<WAVES>(d, scan_tmp, &all);
    const uint32_t tiles = (d + TILE - 1) / TILE;
    uint32_t allt;
    const uint32_t ts = block_exclusive_scan_n<WAVES>(tiles, scan_tmp, &allt);
    if (r < RADIX_DIGITS) {
        rt.pair_prefix[r] = pp;
        rt.tile_start[r] = ts;
    }
    if (r == 0) {
        rt.pair_prefix[RADIX_DIGITS] = all;
        rt.tile_start[RADIX_DIGITS] = allt;
    }
    __syncthreads();
}

// The ROW SEGMENTS of one wave's chunk -- segment slots [wbase, wbase + 512) of the depth-ordered emission: Gaussian i
// (depth order) owns the slots [segoff[i], segoff[i + 1]), one per row of its rect, top row first -- decoded into
// registers with the flag-word technique of decode_chunk (binning_persist.h): starts marked as bits of a 512-bit LDS
// word set, a round's owners from one broadcast read and a population count.
template <int BATCH>
__device__ __forceinline__ void decode_segments(const uint32_t *__restrict__ segoff, const uint32_t *__restrict__ sorted_ids,
                                                const TileRect *__restrict__ rects, int P, long long R, long long wbase,
                                                int g0, uint32_t *__restrict__ cflag, uint32_t (&key)[PS_ITEMS],
                                                uint32_t (&val)[PS_ITEMS]) {
    const int lane = threadIdx.x & 63;
    const uint32_t cbeg = (uint32_t)wbase;
    const uint32_t cend = (uint32_t)(wbase + PS_CHUNK < R ? wbase + PS_CHUNK : R);
    __builtin_amdgcn_wave_barrier();
    if (lane < PS_CHUNK / 32) cflag[lane] = 0u;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int base = g0 + 1;; base += 64) {
        const int j = base + lane;
        const uint32_t o = segoff[j < P ? j : P];  // (segoff[P] = R >= cend: never marked)
        const bool in = o < cend;
        if (in) atomicOr(&cflag[(o - cbeg) >> 5], 1u << ((o - cbeg) & 31u));
        if (__ballot(in) != ~0ull) break;
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
            const bool live = wbase + (r0 + i) * 64 + lane < R;
            const int j = live ? jr[r0 + i] : g0;
            off[i] = segoff[j];
            gid[i] = sorted_ids[j];
        }
        uint2 rc[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; i++) rc[i] = make_uint2(rects[gid[i]].xs, rects[gid[i]].ys);
#pragma unroll
        for (int i = 0; i < BATCH; i++) {
            const int r = r0 + i;
            if (wbase + r * 64 + lane < R) {
                const uint32_t s = cbeg + (uint32_t)(r * 64 + lane);
                const uint32_t minx = rc[i].x & 0xFFFFu, maxx = rc[i].x >> 16, miny = rc[i].y & 0xFFFFu;
                key[r] = ((miny + (s - off[i])) << SEG_ROW_SHIFT) | (minx << 9) | maxx;
                val[r] = gid[i];
            }
        }
    }
}

// ---- R-sized: decode the segments in depth order, scatter them by tile row (one one-sweep pass, radix.h)
#ifndef GSR_ROWS_THREADS_B
#define GSR_ROWS_THREADS_B 512
#endif
#ifndef GSR_SEG_DECODE_BATCH
#define GSR_SEG_DECODE_BATCH 8
#endif
#ifndef GSR_SEG_WAVES
#define GSR_SEG_WAVES 4  // (128 registers: the eight rounds' gathers of the decode stay in flight together; 64 spills them)
#endif
#ifndef GSR_ROWS_THREADS_D
#define GSR_ROWS_THREADS_D 512
#endif
constexpr int ROWS_THREADS_B = GSR_ROWS_THREADS_B, ROWS_THREADS_D = GSR_ROWS_THREADS_D;
template <int ITEMS, int THREADS>
__global__ void __launch_bounds__(THREADS, GSR_SEG_WAVES)
seg_scatter_kernel(int P, long long Dcap, int ybits, const TileRect *__restrict__ rects,
                   const uint32_t *__restrict__ sorted_ids, const uint32_t *__restrict__ offsets,
                   const uint32_t *__restrict__ segoff, const uint32_t *__restrict__ ghist, uint32_t *__restrict__ state,
                   uint32_t *__restrict__ ticket, uint32_t *__restrict__ seg_key, uint32_t *__restrict__ seg_gid,
                   bool bounded, int32_t *__restrict__ ranges_flat, int ranges_words, const int32_t *__restrict__ hull,
                   int32_t *__restrict__ tdiff, int tdiff_words) {
    constexpr int WAVES = THREADS / 64;
    __shared__ OnesweepSmem<ITEMS, THREADS> sm;
    __shared__ uint32_t cflag[WAVES][ITEMS * 64 / 32];
    // the range table and the per-tile count differences start at zero: cleared here, by every workgroup of the launch,
    // before any early exit (both are written two kernels later on this stream)
    for (int t = blockIdx.x * THREADS + threadIdx.x; t < ranges_words; t += gridDim.x * THREADS) ranges_flat[t] = 0;
    for (int t = blockIdx.x * THREADS + threadIdx.x; t < tdiff_words; t += gridDim.x * THREADS) tdiff[t] = 0;
    if (blockIdx.x == 0 && threadIdx.x < 2) ranges_flat[ranges_words + threadIdx.x] = hull[threadIdx.x];  // row `tiles`
    if (bounded && (long long)offsets[P] > Dcap) return;  // the pair count does not fit: nothing is sorted
    const long long R = segoff[P];
    static_assert(ITEMS == PS_ITEMS, "decode_segments decodes 8 rounds of 64 slots");
    uint32_t t_next = 0, t_end = 0;
    for (;;) {  // tiles by ticket: the grid need not cover R (unknown on the host)
        const uint32_t bid = rows_next_tile(sm, ticket, t_next, t_end);
        if ((long long)bid * (ITEMS * THREADS) >= R) return;
        ROWS_TS2(0, 0);
        const int wave = threadIdx.x >> 6;
        const long long wbase = (long long)bid * (ITEMS * THREADS) + (long long)wave * (ITEMS * 64);
        uint32_t key[ITEMS], val[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; r++) { key[r] = 0xFFFFFFFFu; val[r] = 0u; }
        if (wbase < R) {  // wave-uniform
            const int g0 = owner_search(segoff, 0, P, (uint32_t)wbase);
            // (all eight rounds' gathers in flight together: offset / id, then the rect -- two dependent trips to memory
            // instead of eight; measured 23 -> see profiles/r06_rows_pipeline.txt)
            decode_segments<GSR_SEG_DECODE_BATCH>(segoff, sorted_ids, rects, P, R, wbase, g0, cflag[wave], key, val);
        }
        ROWS_TS2(0, 1);  // decoded
        onesweep_scatter(sm, key, val, bid, R, SEG_ROW_SHIFT, ybits, ghist, state, seg_key, seg_gid);
        ROWS_TS2(0, 2);  // scattered
        __syncthreads();  // (the staging area is read by the scatter: the next tile clears the digit tables first)
    }
}

// ---- R-sized: pairoff = exclusive scan of the sorted segments' widths; per-tile pair counts as row difference arrays;
// the segment that owns the first pair of every 512-pair chunk of the row-major emission (so that pair_scatter_kernel
// never searches).  8192 segments per workgroup: all workgroups of this launch are resident at once and start together,
// so a tile's look-back walks back over half of the tiles in front of it -- fewer, larger tiles keep that walk short.
constexpr int SEGSCAN_ITEMS = 32, SEGSCAN_TILE = SCAN_THREADS * SEGSCAN_ITEMS;
__global__ void __launch_bounds__(SCAN_THREADS)
seg_scan_kernel(int P, long long Dcap, int gx, int gy, const uint32_t *__restrict__ offsets,
                const uint32_t *__restrict__ segoff, const uint32_t *__restrict__ seg_key,
                uint32_t *__restrict__ pairoff, unsigned long long *__restrict__ state, uint32_t *__restrict__ ticket,
                int32_t *__restrict__ tdiff, const uint32_t *__restrict__ thist, int32_t *__restrict__ chunk_owner,
                bool bounded) {
    __shared__ uint32_t smem[4];
    __shared__ uint32_t s_bid;
    __shared__ unsigned long long s_excl;
    __shared__ int32_t ldiff[ROWS_LDS][RADIX_DIGITS + 1];
    __shared__ RowTables rt;
    if (bounded && (long long)offsets[P] > Dcap) return;
    const long long R = segoff[P];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stride = gx + 1;
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;  // HW_REG_XCC_ID[3:0]
    int32_t *const trep = tdiff + (size_t)xcc * gy * stride;
    bool have_tables = false;
    for (;;) {
        __syncthreads();  // (s_bid, smem, ldiff of the previous tile are no longer read)
        if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u);
        for (int i = threadIdx.x; i < ROWS_LDS * (RADIX_DIGITS + 1); i += SCAN_THREADS) (&ldiff[0][0])[i] = 0;
        __syncthreads();
        const uint32_t bid = s_bid;
        const long long tbase = (long long)bid * SEGSCAN_TILE;
        if (tbase >= R) return;  // (the grid is sized for R = D: most workgroups leave here, before the row tables)
        if (!have_tables) {
            row_tables<SCAN_THREADS / 64, RADIX_TILE>(rt, thist, gy, smem);
            have_tables = true;
        }
        ROWS_TS2(4, 0);
        const long long base = tbase + (long long)threadIdx.x * SEGSCAN_ITEMS;  // consecutive segments per thread
        const uint32_t row0 = seg_row(seg_key[tbase]);                         // (sorted by row: the tile's first row)
        uint32_t kk[SEGSCAN_ITEMS];
        if (base + SEGSCAN_ITEMS <= R) {
#pragma unroll
            for (int q = 0; q < SEGSCAN_ITEMS / 4; q++) {
                const uint4 u = *reinterpret_cast<const uint4 *>(seg_key + base + 4 * q);
                kk[4 * q] = u.x; kk[4 * q + 1] = u.y; kk[4 * q + 2] = u.z; kk[4 * q + 3] = u.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < SEGSCAN_ITEMS; k++) kk[k] = base + k < R ? seg_key[base + k] : 0u;  // (0: width 0)
        }
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < SEGSCAN_ITEMS; k++) {
            if (base + k < R) {
                const uint32_t minx = seg_minx(kk[k]), maxx = seg_maxx(kk[k]), dr = seg_row(kk[k]) - row0;
                if (dr < (uint32_t)ROWS_LDS) {
                    atomicAdd(&ldiff[dr][minx], 1);
                    atomicAdd(&ldiff[dr][maxx], -1);
                } else {
                    int32_t *t = trep + (size_t)seg_row(kk[k]) * stride;
                    __hip_atomic_fetch_add(t + minx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(t + maxx, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                s += maxx - minx;
            }
        }
        uint32_t tot;
        const uint32_t local = block_exclusive_scan(s, smem, &tot);
        ROWS_TS2(4, 1);  // loaded, counted, scanned
        if (wave == 0) {
            if (lane == 0) st_agent64(&state[bid], (unsigned long long)tot | (bid == 0 ? LB64_PRE : LB64_AGG));
            unsigned long long excl = 0;
            if (bid > 0) {
                long long top = (long long)bid - 1;
                while (true) {
                    const long long j = top - lane;
                    unsigned long long x = j >= 0 ? ld_agent64(&state[j]) : LB64_PRE;
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
                if (bid > 0) st_agent64(&state[bid], (excl + tot) | LB64_PRE);
                s_excl = excl;
                if (tbase + SEGSCAN_TILE >= R) pairoff[R] = (uint32_t)(excl + tot);  // = D
            }
        }
        __syncthreads();  // (also: every thread's LDS differences are in)
        ROWS_TS2(4, 2);  // look-back done
        uint32_t run = (uint32_t)s_excl + local;
#pragma unroll
        for (int k = 0; k < SEGSCAN_ITEMS; k++) {
            if (base + k < R) {
                pairoff[base + k] = run;
                // the 512-pair chunks of the row-major emission start at rbeg + 512 c (tiles never straddle rows): the
                // chunk boundary inside this segment's pairs, if any (a segment has <= 256 pairs: at most one)
                const uint32_t row = seg_row(kk[k]), w = seg_maxx(kk[k]) - seg_minx(kk[k]);
                const uint32_t rel = run - rt.pair_prefix[row];
                const uint32_t c = (rel + PS_CHUNK - 1) / PS_CHUNK;
                if (c * PS_CHUNK < rel + w)
                    chunk_owner[(size_t)rt.tile_start[row] * (RADIX_TILE / PS_CHUNK) + c] = (int32_t)(base + k);
                run += w;
            }
        }
        // the tile's differences of its first ROWS_LDS rows -> this XCD's replica of the per-tile table (workgroup scope:
        // the adds stay in the XCD's L2, no memory-side atomic; the replicas are summed by tile_base_kernel, which
        // sees them complete at the kernel boundary -- the histogram idiom of radix.h)
        for (int i = threadIdx.x; i < ROWS_LDS * stride; i += SCAN_THREADS) {
            const int dr = i / stride, x = i - dr * stride;
            const int32_t dv = ldiff[dr][x];
            if (dv != 0 && (int)row0 + dr < gy)
                __hip_atomic_fetch_add(trep + (size_t)(row0 + dr) * stride + x, dv, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// ---- tiles-sized: the start of every tile's list (row pair prefix + prefix of the row's tile counts) and the ranges.
// One workgroup per tile row.  (Inside pair_scatter_kernel this was eight replica loads and two workgroup scans PER TILE of
// 4096 pairs: 4 of the ~20 us a tile lives, profiles/r06_rows_pipeline.txt.)
__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
tile_base_kernel(int P, long long Dcap, int gx, int gy, const uint32_t *__restrict__ offsets,
                 const uint32_t *__restrict__ thist, const int32_t *__restrict__ tdiff,
                 const uint8_t *__restrict__ mask, uint32_t *__restrict__ tilebase, int2 *__restrict__ ranges,
                 bool bounded) {
    __shared__ RowTables rt;
    __shared__ uint32_t scan_tmp[GSR_ONE_DIM_BLOCK / 64];
    if (bounded && (long long)offsets[P] > Dcap) return;
    row_tables<GSR_ONE_DIM_BLOCK / 64, RADIX_TILE>(rt, thist, gy, scan_tmp);
    const uint32_t row = blockIdx.x, d = threadIdx.x;
    const uint32_t rbeg = rt.pair_prefix[row];
    // thread d owns column d (gx <= 256); the closing difference at x = gx is never needed (no count behind it)
    int32_t dv = 0;
    if (d < (uint32_t)gx)
        for (int x = 0; x < RADIX_REPLICAS; x++) dv += tdiff[((size_t)x * gy + row) * (gx + 1) + d];
    uint32_t all;
    const uint32_t cx = block_exclusive_scan_n<GSR_ONE_DIM_BLOCK / 64>((uint32_t)dv, scan_tmp, &all) + (uint32_t)dv;
    const uint32_t cnt = d < (uint32_t)gx ? cx : 0u;
    const uint32_t cstart = block_exclusive_scan_n<GSR_ONE_DIM_BLOCK / 64>(cnt, scan_tmp, &all);
    if (d < (uint32_t)gx) {
        const uint32_t t = row * (uint32_t)gx + d;
        tilebase[t] = rbeg + cstart;
        if (cnt && mask[t]) ranges[t] = make_int2((int)(rbeg + cstart), (int)(rbeg + cstart + cnt));
    }
}

// ---- D-sized, the only one: decode a tile's pairs from the row-major order, rank by column, write point_list
template <int ITEMS, int THREADS>
__global__ void __launch_bounds__(THREADS, 8)
pair_scatter_kernel(int P, long long Dcap, int gx, int gy, int xbits, const uint32_t *__restrict__ offsets,
                    const uint32_t *__restrict__ segoff, const uint32_t *__restrict__ seg_key,
                    const uint32_t *__restrict__ seg_gid, const uint32_t *__restrict__ pairoff,
                    const uint32_t *__restrict__ thist, const uint32_t *__restrict__ tilebase,
                    const int32_t *__restrict__ chunk_owner, uint32_t *__restrict__ state,
                    uint32_t *__restrict__ ticket, uint32_t *__restrict__ point_list, bool bounded) {
    constexpr int WAVES = THREADS / 64;
    constexpr int TILE = ITEMS * THREADS;
    static_assert(TILE == RADIX_TILE, "seg_scan_kernel's chunk owners and row tables count tiles of RADIX_TILE pairs");
    __shared__ OnesweepSmem<ITEMS, THREADS> sm;
    __shared__ RowTables rt;
    __shared__ uint32_t cflag[WAVES][ITEMS * 64 / 32];
    __shared__ uint32_t s_row;
    if (bounded && (long long)offsets[P] > Dcap) return;
    const long long R = segoff[P];
    row_tables<WAVES, TILE>(rt, thist, gy, sm.scan_tmp);
    const uint32_t NT = rt.tile_start[RADIX_DIGITS];
    const uint32_t xmask = (1u << xbits) - 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t d = threadIdx.x;
    uint32_t t_next = 0, t_end = 0;
    for (;;) {
        const uint32_t bid = rows_next_tile(sm, ticket, t_next, t_end);  // (a barrier inside: the previous tile's LDS is free)
        if (bid >= NT) return;
        ROWS_TS(0);
        if (d < (uint32_t)gy && rt.tile_start[d] <= bid && bid < (d + 1 < RADIX_DIGITS ? rt.tile_start[d + 1]
                                                                                       : rt.tile_start[RADIX_DIGITS]))
            s_row = d;  // (exactly one row owns the tile: rows without pairs own none)
        __syncthreads();
        const uint32_t row = s_row;
        const uint32_t lt = bid - rt.tile_start[row];
        const long long rbeg = rt.pair_prefix[row];
        const long long rend = row + 1 < RADIX_DIGITS ? rt.pair_prefix[row + 1] : rt.pair_prefix[RADIX_DIGITS];
        const long long tbase = rbeg + (long long)lt * TILE;
        const long long tend = tbase + TILE < rend ? tbase + TILE : rend;  // (tiles never straddle rows)
        const long long wbase = tbase + (long long)wave * (ITEMS * 64);
        // the start of this row's (row, column) lists: requested now, used after the ranking
        const uint32_t tb = d < (uint32_t)gx ? tilebase[row * (uint32_t)gx + d] : 0u;
        uint32_t key[ITEMS], val[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; r++) { key[r] = 0u; val[r] = 0u; }
        if (wbase < tend) {  // wave-uniform
            // the segment that owns the chunk's first pair (seg_scan_kernel left it), then the flag-word decode over
            // pairoff (strictly increasing: every segment has at least one tile)
            const int s0 = __builtin_amdgcn_readfirstlane(chunk_owner[(size_t)bid * (TILE / (ITEMS * 64)) + wave]);
            const uint32_t cbeg = (uint32_t)wbase;
            const uint32_t cend = (uint32_t)(wbase + ITEMS * 64 < tend ? wbase + ITEMS * 64 : tend);
            uint32_t *cf = cflag[wave];
            __builtin_amdgcn_wave_barrier();
            if (lane < ITEMS * 64 / 32) cf[lane] = 0u;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            for (int base = s0 + 1;; base += 64) {
                const long long j = (long long)base + lane;
                const uint32_t o = pairoff[j < R ? j : R];  // (pairoff[R] = D >= cend: never marked)
                const bool in = o < cend;
                if (in) atomicOr(&cf[(o - cbeg) >> 5], 1u << ((o - cbeg) & 31u));
                if (__ballot(in) != ~0ull) break;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            int carry = 0;
            const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
            int jr[ITEMS];
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                const unsigned long long m = (unsigned long long)cf[2 * r] | ((unsigned long long)cf[2 * r + 1] << 32);
                jr[r] = s0 + carry + __popcll(m & le);
                carry += __popcll(m);
            }
            uint32_t off[ITEMS], sk[ITEMS];
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {  // (all gathers in flight together: consecutive, mostly shared addresses)
                const bool live = wbase + r * 64 + lane < tend;
                const int j = live ? jr[r] : s0;
                off[r] = pairoff[j];
                sk[r] = seg_key[j];
                val[r] = seg_gid[j];
            }
#pragma unroll
            for (int r = 0; r < ITEMS; r++)
                key[r] = seg_minx(sk[r]) + ((cbeg + (uint32_t)(r * 64 + lane)) - off[r]);  // the pair's column
        }
        ROWS_TS(1);  // decoded
        // ---- per-wave digit counts -> workgroup counts, published for the tiles behind this one IN THE SAME ROW
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            if (wbase + r * 64 + lane < tend) {
                const uint32_t dg = key[r] & xmask;
                atomicAdd(reinterpret_cast<uint32_t *>(sm.wtab[wave]) + (dg >> 1), 1u << (16 * (dg & 1u)));
            }
        }
        __syncthreads();
        ROWS_TS(2);  // counted
        const bool live = d <= xmask;
        uint32_t tot = 0, run, cstart;
        uint32_t *srow = state + (size_t)bid * RADIX_DIGITS;
        {
            uint32_t cnt[WAVES];
#pragma unroll
            for (int w = 0; w < WAVES; w++) {
                cnt[w] = d < RADIX_DIGITS ? sm.wtab[w][d] : 0u;
                tot += cnt[w];
            }
            if (live) st_agent(&srow[d], lt == 0 ? (tot | LB_PRE) : (tot + 1u));
            uint32_t all;
            run = block_exclusive_scan_n<WAVES>(tot, sm.scan_tmp, &all);  // local start of column d
            cstart = tb;
            if (d < RADIX_DIGITS) {
                uint32_t c = run;
#pragma unroll
                for (int w = 0; w < WAVES; w++) {
                    sm.wtab[w][d] = (uint16_t)c;
                    c += cnt[w];
                }
            }
        }
        __syncthreads();
        ROWS_TS(3);  // published, scans done
        const unsigned long long ltm = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const bool valid = wbase + r * 64 + lane < tend;
            const uint32_t dg = key[r] & xmask;
            const unsigned long long m = match_digit(dg, valid, xbits);
            const uint32_t rank = __popcll(m & ltm);
            uint16_t *cursor = sm.wtab[wave];
            uint32_t pos = 0;
            if (valid) pos = cursor[dg] + rank;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) cursor[dg] = (uint16_t)(pos + (uint32_t)__popcll(m));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                sm.skey[pos] = dg;
                sm.sval[pos] = val[r];
            }
        }
        ROWS_TS(4);  // ranked
        {
            uint32_t excl = 0;
            if (live && lt > 0) {  // look back over the earlier tiles of THIS row (its first tile publishes a prefix)
                long long j = (long long)bid - 1;
                bool done = false;
                while (!done) {
                    uint32_t v[LB_WINDOW];
#pragma unroll
                    for (int k = 0; k < LB_WINDOW; k++)
                        v[k] = (j - k >= 0) ? ld_agent(&state[(size_t)(j - k) * RADIX_DIGITS + d]) : LB_PRE;
#pragma unroll
                    for (int k = 0; k < LB_WINDOW; k++) {
                        if (!done) {
                            uint32_t x = v[k];
                            while (x == 0u) {
                                __builtin_amdgcn_s_sleep(1);
                                x = ld_agent(&state[(size_t)(j - k) * RADIX_DIGITS + d]);
                            }
                            done = (x & LB_PRE) != 0u;
                            excl += done ? (x & LB_VAL) : (x - 1u);
                        }
                    }
                    j -= LB_WINDOW;
                }
                st_agent(&srow[d], ((excl + tot) & LB_VAL) | LB_PRE);
            }
            if (d < RADIX_DIGITS) sm.gbase[d] = cstart + excl - run;
        }
        __syncthreads();
        ROWS_TS(5);  // look-back done (thread 0's digit; the barrier waits for all)
        const int count = (int)(tend - tbase);
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const int i = r * THREADS + threadIdx.x;
            if (i < count) point_list[sm.gbase[sm.skey[i]] + (uint32_t)i] = sm.sval[i];
        }
        ROWS_TS(6);  // stores issued
    }
}

// ------------------------------------------------------------------------------------------------- host side
struct RowsLayout {
    size_t seg_key, seg_gid, pairoff, tdiff, tilebase, chunk_owner, ctrl, tickets, scan_state, seg_state, pair_state,
        ctrl_bytes, total;
};
inline RowsLayout rows_layout(int64_t D, int gx, int gy) {
    RowsLayout L;
    size_t o = 0;
    const size_t nr = align_up((size_t)(D + 1) * 4);  // R <= D segments
    L.seg_key = o; o += nr;
    L.seg_gid = o; o += nr;
    L.pairoff = o; o += nr;
    // (one replica per XCD, cleared by seg_scatter_kernel)
    L.tdiff = o; o += align_up(sizeof(int32_t) * RADIX_REPLICAS * (size_t)gy * (size_t)(gx + 1));
    L.tilebase = o; o += align_up(sizeof(uint32_t) * (size_t)gx * gy);
    L.chunk_owner = o; o += align_up(sizeof(int32_t) * (size_t)(radix_blocks(D) + gy + 1) * (RADIX_TILE / PS_CHUNK));
    L.ctrl = o;  // [ctrl, ctrl + ctrl_bytes) is zero before the first kernel
    L.tickets = o; o += 256;
    L.scan_state = o; o += align_up(sizeof(unsigned long long) * (size_t)((D + SEGSCAN_TILE - 1) / SEGSCAN_TILE + 2));
    L.seg_state = o; o += align_up(sizeof(uint32_t) * (size_t)(radix_blocks(D) + 1) * RADIX_DIGITS);
    L.pair_state = o; o += align_up(sizeof(uint32_t) * (size_t)(radix_blocks(D) + gy + 1) * RADIX_DIGITS);
    L.ctrl_bytes = o - L.ctrl;
    L.total = o;
    return L;
}

}  // namespace
