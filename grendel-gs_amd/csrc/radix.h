// radix.h -- the one-sweep LSD radix sort of (key,value) u32 pairs shared by binning.hip (depth / tile sorts)
// and knn.hip (Morton sort).  Include inside the translation unit; everything lives in an anonymous namespace.
//
// The digit histograms of ALL passes are accumulated by the kernel that produces the keys (multihist_add /
// multihist_flush, one replica per XCD so that the adds are L2-local atomics); each pass is then one kernel
// that gets its workgroup's global digit offsets by decoupled look-back.  Workgroups take their tile from
// a ticket counter, so a workgroup only ever waits for workgroups that are already running.
#pragma once
#include "common.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048 elements per workgroup

constexpr int RADIX_THREADS = 256;  // block size of the key PRODUCERS (their LDS histograms: one digit per thread)
constexpr int RADIX_DIGITS = 256;
static_assert(GSR_ONE_DIM_BLOCK == RADIX_DIGITS, "histogram tables are initialised one digit per thread");
constexpr int RADIX_MAX_PASSES = 4;
constexpr int RADIX_REPLICAS = 8;  // histogram replicas, one per XCD
constexpr int RADIX_TILE = 4096;           // pairs per workgroup of a radix pass
constexpr long long RADIX_SHORT_N = 1ll << 22;  // at most this many pairs: the 1024-thread configuration

constexpr long long RADIX_MAX_N = (1ll << 31) - 1;  // prefixes share a word with one flag bit; tile ranges are int32

// look-back state word: 0 = nothing yet; bit 31 set = inclusive prefix in [30:0]; otherwise the workgroup's aggregate + 1
// (an aggregate is at most one tile, 4096, so "+ 1" can never reach bit 31).  31 value bits: up to 2^31 - 1 pairs --
// one MI355X holds the 1.2e9 pairs of the 40 M-Gaussian 4K frame (configs[4]) in ~25 GB of its 288 GB.
constexpr uint32_t LB_PRE = 0x80000000u, LB_VAL = 0x7FFFFFFFu;
#ifndef GSR_LB_WINDOW
#define GSR_LB_WINDOW 4
#endif
constexpr int LB_WINDOW = GSR_LB_WINDOW;  // independent state loads in flight per thread and look-back round
// 64-bit variant for the offsets scan (values up to 2^32)
constexpr unsigned long long LB64_PRE = 2ull << 62, LB64_AGG = 1ull << 62, LB64_VAL = (1ull << 62) - 1ull;

struct RadixPlan {
    int passes;
    int shift[RADIX_MAX_PASSES];
    int nbits[RADIX_MAX_PASSES];
};

// split key bits [bit_lo, bit_hi) evenly over the minimum number of <= 8-bit passes (13 tile bits -> 7 + 6)
RadixPlan radix_plan(int bit_lo, int bit_hi) {
    RadixPlan pl{};
    const int total = bit_hi - bit_lo;
    pl.passes = (total + 7) / 8;
    int shift = bit_lo;
    for (int p = 0; p < pl.passes; p++) {
        const int nb = (total - (shift - bit_lo) + (pl.passes - p) - 1) / (pl.passes - p);
        pl.shift[p] = shift;
        pl.nbits[p] = nb;
        shift += nb;
    }
    return pl;
}

__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) {
    return __hip_atomic_load(const_cast<uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent64(const unsigned long long *p) {
    return __hip_atomic_load(const_cast<unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent64(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

// ------------------------------------------------------------- digit histograms of all passes at once
// Called by the kernels that produce sort keys.  mh = LDS [passes][256]; one call per key per thread.
// When every valid lane of the wave holds the same digit (top depth bytes, upper tile bits of one
// splat) a single lane adds the population count instead of 64 serialised same-address LDS atomics.
__device__ __forceinline__ void multihist_add(uint32_t (*mh)[RADIX_DIGITS], const RadixPlan &pl, uint32_t key,
                                              bool valid) {
    const unsigned long long vm = __ballot(valid);
    if (vm == 0ull) return;
    const int src = __ffsll((long long)vm) - 1;
    const int lane = threadIdx.x & 63;
    for (int p = 0; p < pl.passes; p++) {
        const uint32_t d = (key >> pl.shift[p]) & ((1u << pl.nbits[p]) - 1u);
        const uint32_t d0 = __shfl(d, src, 64);
        if (__ballot(valid && d != d0) == 0ull) {
            if (lane == src) atomicAdd(&mh[p][d0], (uint32_t)__popcll(vm));
        } else if (valid) {
            atomicAdd(&mh[p][d], 1u);
        }
    }
}
__device__ __forceinline__ void multihist_flush(uint32_t (*mh)[RADIX_DIGITS], const RadixPlan &pl,
                                                uint32_t *__restrict__ ghist) {
    for (int p = 0; p < pl.passes; p++)
        for (int d = threadIdx.x; d < (1 << pl.nbits[p]); d += blockDim.x) {
            const uint32_t c = mh[p][d];
            // one histogram replica per XCD: the adds stay in that XCD's L2 (workgroup scope, no memory-side
            // atomic); replicas become visible to the consuming kernels at the kernel boundary
            const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;  // HW_REG_XCC_ID[3:0]
            if (c)
                __hip_atomic_fetch_add(&ghist[(xcc * RADIX_MAX_PASSES + p) * RADIX_DIGITS + d], c, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        }
}

// lanes holding the same digit (restricted to `valid` lanes)
__device__ __forceinline__ unsigned long long match_digit(uint32_t d, bool valid, int nbits) {
    const unsigned long long v = __ballot(valid);
    uint32_t mlo = (uint32_t)v, mhi = (uint32_t)(v >> 32);
    for (int b = 0; b < nbits; b++) {
        // s = all ones where this lane's bit b is set; a peer keeps its place in the mask when its bit equals ours:
        // m &= ~(ballot ^ s)  -- bfe, compare, and per half one xnor + one and
        const int s = ((int)(d << (31 - b))) >> 31;
        const unsigned long long bm = __ballot(s != 0);
        mlo &= ~((uint32_t)bm ^ (uint32_t)s);
        mhi &= ~((uint32_t)(bm >> 32) ^ (uint32_t)s);
    }
    return ((unsigned long long)mhi << 32) | mlo;
}

// ------------------------------------------------------------------------- one radix pass, one kernel
// Workgroup `bid` (ticket order) owns ITEMS*256 consecutive pairs.
//   1. per-wave LDS digit counts -> workgroup count per digit, published as AGGREGATE in state[bid][d];
//   2. thread d looks back over state[bid-1 .. ][d] (LB_WINDOW independent loads in flight), adding
//      aggregates until it meets an inclusive PREFIX, then publishes its own inclusive prefix;
//   3. global start of digit d = exclusive scan of the pass histogram (256 values, done by every
//      workgroup) + look-back sum; stable ranks from wave64 ballots + per-wave LDS cursors; the pairs
//      are placed in digit order INSIDE LDS and streamed out so that every digit's run leaves the
//      workgroup as contiguous, coalesced stores.
// the same for a workgroup of WAVES wavefronts
template <int WAVES>
__device__ __forceinline__ uint32_t block_exclusive_scan_n(uint32_t v, uint32_t *smem, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) {
        const uint32_t s = smem[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// LDS of one radix pass workgroup
template <int ITEMS, int THREADS>
struct OnesweepSmem {
    // per-wave digit counts, then per-wave cursors: 16-bit (a wave owns <= 512 pairs, a cursor is < 4096), so that the
    // workgroup's LDS stays under 40 KB and FOUR workgroups fit a CU (the passes are latency chains: throughput is
    // the number of workgroups in flight).  The histogram adds go through the 32-bit view (two digits per word).
    uint16_t wtab[THREADS / 64][RADIX_DIGITS];
    uint32_t gbase[RADIX_DIGITS];  // global start of the digit's run minus its local start
    uint32_t skey[ITEMS * THREADS], sval[ITEMS * THREADS];
    uint32_t scan_tmp[THREADS / 64];
    uint32_t bid;
};

// ticket of this workgroup (its tile index) + cleared per-wave digit tables; ends with a barrier
template <int ITEMS, int THREADS>
__device__ __forceinline__ uint32_t onesweep_begin(OnesweepSmem<ITEMS, THREADS> &sm, uint32_t *__restrict__ ticket) {
    if (threadIdx.x == 0) sm.bid = atomicAdd(ticket, 1u);
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++)
        if (threadIdx.x < RADIX_DIGITS) sm.wtab[w][threadIdx.x] = 0;
    __syncthreads();
    return sm.bid;
}

// steps 1-3 above for the ITEMS pairs per thread held in registers: pair r of a lane is element
// bid * TILE + wave * ITEMS * 64 + r * 64 + lane of the pass input
template <int ITEMS, int THREADS>
__device__ __forceinline__ void onesweep_scatter(OnesweepSmem<ITEMS, THREADS> &sm, const uint32_t (&key)[ITEMS],
                                                 const uint32_t (&val)[ITEMS], uint32_t bid, long long n, int shift,
                                                 int nbits, const uint32_t *__restrict__ ghist,
                                                 uint32_t *__restrict__ state, uint32_t *__restrict__ keys_out,
                                                 uint32_t *__restrict__ vals_out) {
    constexpr int TILE = ITEMS * THREADS;
    constexpr int WAVES = THREADS / 64;
    const uint32_t mask = (1u << nbits) - 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long bbase = (long long)bid * TILE;
    const long long wbase = bbase + (long long)wave * (ITEMS * 64);
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const long long j = wbase + r * 64 + lane;
        if (j < n) {
            const uint32_t dg = (key[r] >> shift) & mask;
            atomicAdd(reinterpret_cast<uint32_t *>(sm.wtab[wave]) + (dg >> 1), 1u << (16 * (dg & 1u)));  // no carry: <= 512
        }
    }
    __syncthreads();
    // thread d: digit d.  Order (round 5): the aggregate is published and the per-wave cursors are set up first, then ALL
    // threads rank the tile into the LDS staging area, and only then do the digit threads look back -- the ranking
    // (~2.8 us) used to sit BEHIND the look-back, whose wait (for the slowest aggregate of the ~1024 resident
    // workgroups, profiles/r04_radix_timeline.txt) is idle time that the ranking now fills.
    const uint32_t d = threadIdx.x;  // threads >= 256 (8-wave workgroups) only take part in the scans
    const bool live = d <= mask;
    uint32_t tot = 0, run, dstart;
    uint32_t *row = state + (size_t)bid * RADIX_DIGITS;
    {
        uint32_t cnt[WAVES];
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            cnt[w] = d < RADIX_DIGITS ? sm.wtab[w][d] : 0u;
            tot += cnt[w];
        }
        if (live) st_agent(&row[d], bid == 0 ? (tot | LB_PRE) : (tot + 1u));
        uint32_t all;
        run = block_exclusive_scan_n<WAVES>(tot, sm.scan_tmp, &all);  // local start of digit d
        uint32_t gh = 0;  // pass histogram = sum of the per-XCD replicas
        if (live)
            for (int x = 0; x < RADIX_REPLICAS; x++) gh += ghist[(size_t)x * RADIX_MAX_PASSES * RADIX_DIGITS + d];
        dstart = block_exclusive_scan_n<WAVES>(gh, sm.scan_tmp, &all);  // global start
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
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const long long j = wbase + r * 64 + lane;
        const bool valid = j < n;
        const uint32_t dg = (key[r] >> shift) & mask;
        const unsigned long long m = match_digit(dg, valid, nbits);
        const uint32_t rank = __popcll(m & lt);
        // (plain LDS accesses fenced for the compiler: a `volatile` pointer here turned them into FLAT loads / stores with
        // system-scope bits and a vmcnt(0) wait each -- ~1 us per round; LDS operations of one wave execute in order)
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
    {
        uint32_t excl = 0;
        if (live && bid > 0) {
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
            st_agent(&row[d], ((excl + tot) & LB_VAL) | LB_PRE);
        }
        if (d < RADIX_DIGITS) sm.gbase[d] = dstart + excl - run;
    }
    __syncthreads();
    const long long rem = n - bbase;
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
}

template <int ITEMS, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS >= 1024 ? 4 : 8)  // 8-wave workgroups: <= 64 VGPRs, four per CU
radix_onesweep_kernel(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                      uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, long long n, int shift,
                      int nbits, const uint32_t *__restrict__ ghist, uint32_t *__restrict__ state,
                      uint32_t *__restrict__ ticket, const uint32_t *__restrict__ n_dev = nullptr) {
    __shared__ OnesweepSmem<ITEMS, THREADS> sm;
    if (n_dev) {  // bounded launch: the grid covers a capacity `n`, the true count is on the device
        const long long d = *n_dev;
        if (d > n) return;  // does not fit: the caller re-runs with the exact count
        n = d;
    }
    const uint32_t bid = onesweep_begin(sm, ticket);
    if ((long long)bid * (ITEMS * THREADS) >= n) return;  // tiles past the end (bounded launches only)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave w owns the contiguous sub-chunk [base + w*ITEMS*64, +ITEMS*64), walked in ITEMS rounds of 64
    const long long wbase = (long long)bid * (ITEMS * THREADS) + (long long)wave * (ITEMS * 64);
    uint32_t key[ITEMS], val[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const long long j = wbase + r * 64 + lane;
        key[r] = j < n ? keys_in[j] : 0xFFFFFFFFu;
        val[r] = j < n ? vals_in[j] : 0u;
    }
    onesweep_scatter(sm, key, val, bid, n, shift, nbits, ghist, state, keys_out, vals_out);
}


inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

inline long long radix_blocks(long long n) {
    const long long tile = RADIX_TILE;
    return (n + tile - 1) / tile;
}

// control block of one sort (zeroed by ONE memset before the producer kernel runs):
//   ghist[4][256] | tickets[8] | scan_state u64[scan_blocks] | radix state u32[passes][blocks][256]
struct CtrlLayout {
    size_t ghist, tickets, scan_state, radix_state, total;
};
CtrlLayout ctrl_layout(long long n, int passes, bool with_scan) {
    CtrlLayout C;
    size_t o = 0;
    C.ghist = o; o += sizeof(uint32_t) * RADIX_REPLICAS * RADIX_MAX_PASSES * RADIX_DIGITS;
    C.tickets = o; o += 256;
    C.scan_state = o;
    if (with_scan) o += align_up(sizeof(unsigned long long) * (size_t)((n + SCAN_TILE - 1) / SCAN_TILE + 1));
    C.radix_state = o;
    o += align_up(sizeof(uint32_t) * (size_t)passes * (size_t)(radix_blocks(n) + 1) * RADIX_DIGITS);
    C.total = o;
    return C;
}

// stable LSD radix sort of (key,value) u32 pairs following `plan`; the pass histograms in ctrl->ghist were
// accumulated by the producer of the keys.  Ping-pongs between (k0,v0) and (k1,v1); *result_in_first tells
// where the sorted keys ended up; the last pass writes the values to final_vals when given.
int radix_sort_pairs(uint32_t *k0, uint32_t *v0, uint32_t *k1, uint32_t *v1, long long n, const RadixPlan &plan,
                     char *ctrl, const CtrlLayout &C, int *result_in_first, hipStream_t stream,
                     uint32_t *final_vals) {
    *result_in_first = 1;
    if (n <= 0) return 0;
    if (n > RADIX_MAX_N) return GSR_EINVAL;
    const int nb = (int)radix_blocks(n);
    uint32_t *ghist = reinterpret_cast<uint32_t *>(ctrl + C.ghist);
    uint32_t *tickets = reinterpret_cast<uint32_t *>(ctrl + C.tickets);
    uint32_t *state = reinterpret_cast<uint32_t *>(ctrl + C.radix_state);
    uint32_t *ki = k0, *vi = v0, *ko = k1, *vo = v1;
    for (int p = 0; p < plan.passes; p++) {
        uint32_t *vdst = (p == plan.passes - 1 && final_vals) ? final_vals : vo;  // last pass can land the values
        // 4096 pairs per workgroup either way; many small wavefront-rich workgroups hide the latency chain of the
        // short sorts (P ~ 1e6: 1024 threads x 4 pairs), 512 x 8 is the measured optimum of the long ones
        if (n <= RADIX_SHORT_N)
            hipLaunchKernelGGL((radix_onesweep_kernel<RADIX_TILE / 1024, 1024>), dim3(nb), dim3(1024), 0, stream, ki, vi,
                               ko,
                           vdst, n, plan.shift[p], plan.nbits[p], ghist + p * RADIX_DIGITS,
                           state + (size_t)p * nb * RADIX_DIGITS, tickets + 1 + p);
        else
            hipLaunchKernelGGL((radix_onesweep_kernel<RADIX_TILE / 512, 512>), dim3(nb), dim3(512), 0, stream, ki, vi, ko,
                           vdst, n, plan.shift[p], plan.nbits[p], ghist + p * RADIX_DIGITS,
                           state + (size_t)p * nb * RADIX_DIGITS, tickets + 1 + p);
        uint32_t *t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
        *result_in_first ^= 1;
    }
    GSR_LAUNCH_CHECK();
    return 0;
}

}  // namespace
