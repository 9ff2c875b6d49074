// Probe for DESIGN.md section 8 item 3 (c): can the D (tile, Gaussian) pairs be placed into per-tile segments with one
// RETURNING atomic per pair (counting scatter, order fixed later by a per-tile sort)?  Measures 11 M atomicAdd-with-return
// on 8 160 cursors + the 8-byte store it addresses, at agent scope (coherent across the 8 XCDs) and at workgroup scope
// (XCD-local L2: not usable as is, the upper bound of what a per-XCD tile partition could reach).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_probe tools/probes/atomic_scatter_probe.hip && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SCOPE, bool RET>
__global__ void scatter(long long D, int tiles, const uint32_t *__restrict__ tile_of, uint32_t *__restrict__ cursor,
                        const uint32_t *__restrict__ base, uint2 *__restrict__ list) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    const uint32_t t = tile_of[i];
    if (RET) {
        const uint32_t pos = __hip_atomic_fetch_add(&cursor[t], 1u, __ATOMIC_RELAXED, SCOPE);
        list[base[t] + pos] = make_uint2((uint32_t)i, t);
    } else {
        __hip_atomic_fetch_add(&cursor[t], 1u, __ATOMIC_RELAXED, SCOPE);
    }
}

int main() {
    const long long D = 11'000'000;
    const int tiles = 8160, gx = 120;
    std::vector<uint32_t> h(D), cnt(tiles, 0), base(tiles + 1, 0);
    // emission-like order: consecutive slots = the w x h rect of one Gaussian (4 x 4 tiles around a random centre)
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (long long i = 0; i < D;) {
        const int cx = rnd() % (gx - 4), cy = rnd() % (68 - 4);
        for (int k = 0; k < 16 && i < D; k++, i++) h[i] = (cy + k / 4) * gx + cx + k % 4;
    }
    for (long long i = 0; i < D; i++) cnt[h[i]]++;
    for (int t = 0; t < tiles; t++) base[t + 1] = base[t] + cnt[t];
    uint32_t *d_tile, *d_cur, *d_base; uint2 *d_list;
    CK(hipMalloc(&d_tile, D * 4)); CK(hipMalloc(&d_cur, tiles * 4)); CK(hipMalloc(&d_base, (tiles + 1) * 4)); CK(hipMalloc(&d_list, D * 8));
    CK(hipMemcpy(d_tile, h.data(), D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_base, base.data(), (tiles + 1) * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int B = 256; const int G = (int)((D + B - 1) / B);
    for (int variant = 0; variant < 4; variant++) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipMemset(d_cur, 0, tiles * 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (variant == 0) scatter<__HIP_MEMORY_SCOPE_AGENT, true><<<G, B>>>(D, tiles, d_tile, d_cur, d_base, d_list);
            if (variant == 1) scatter<__HIP_MEMORY_SCOPE_WORKGROUP, true><<<G, B>>>(D, tiles, d_tile, d_cur, d_base, d_list);
            if (variant == 2) scatter<__HIP_MEMORY_SCOPE_AGENT, false><<<G, B>>>(D, tiles, d_tile, d_cur, d_base, d_list);
            if (variant == 3) scatter<__HIP_MEMORY_SCOPE_WORKGROUP, false><<<G, B>>>(D, tiles, d_tile, d_cur, d_base, d_list);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const char *names[4] = {"agent scope, returning + 8-byte store", "workgroup scope (XCD-local), returning + store",
                                "agent scope, no return", "workgroup scope, no return"};
        printf("%-50s %8.1f us for %lld pairs on %d cursors\n", names[variant], best * 1e3f, D, tiles);
    }
    return 0;
}
