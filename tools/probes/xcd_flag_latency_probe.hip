// How long does a look-back state word take from one workgroup to another?  Two workgroups ping-pong a counter through
// two words with the accesses csrc/radix.h uses (relaxed agent-scope atomic store / load): once with both workgroups on
// the SAME XCD (blockIdx 0 and 8 of a 16-workgroup grid: dispatch is round-robin over the 8 XCDs), once on different XCDs
// (blockIdx 0 and 1).  Half a round trip = the latency one hop of the prefix ripple pays.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_flag_probe tools/probes/xcd_flag_latency_probe.hip && ./xcd_flag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void pingpong(uint32_t *f, int a, int b, int n, unsigned long long *out, uint32_t *xcc) {
    if (threadIdx.x != 0) return;
    const uint32_t me = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;
    if ((int)blockIdx.x == a) {
        xcc[0] = me;
        const unsigned long long t0 = wall_clock64();
        for (int i = 1; i <= n; i++) {
            __hip_atomic_store(&f[0], (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(&f[64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)i) __builtin_amdgcn_s_sleep(1);
        }
        out[0] = wall_clock64() - t0;
    } else if ((int)blockIdx.x == b) {
        xcc[1] = me;
        for (int i = 1; i <= n; i++) {
            while (__hip_atomic_load(&f[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)i) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(&f[64], (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the same through the XCD's own L2: workgroup-scope read-modify-write atomics execute in L2, which all CUs of ONE XCD
// share (csrc/binning.hip uses them for the per-XCD histogram replicas).  Only valid between workgroups of one XCD;
// polls are bounded so that a pair that cannot see each other ends instead of hanging the device.
__global__ void pingpong_l2(uint32_t *f, int a, int b, int n, unsigned long long *out, uint32_t *xcc) {
    if (threadIdx.x != 0) return;
    const uint32_t me = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;
    // (a fetch_add of 0 is folded into a plain load, which the CU's vector cache answers: poll with a compare-exchange)
    auto ld = [](uint32_t *p) {
        uint32_t expected = 0xFFFFFFFFu;
        __hip_atomic_compare_exchange_strong(p, &expected, 0xFFFFFFFFu, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_WORKGROUP);
        return expected;
    };
    auto st = [](uint32_t *p, uint32_t v) { __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    if ((int)blockIdx.x == a) {
        xcc[0] = me;
        const unsigned long long t0 = wall_clock64();
        bool ok = true;
        for (int i = 1; i <= n && ok; i++) {
            st(&f[0], (uint32_t)i);
            long polls = 0;
            while (ld(&f[64]) != (uint32_t)i) { if (++polls > 2000000) { ok = false; break; } }
        }
        out[0] = ok ? wall_clock64() - t0 : 0ull;
    } else if ((int)blockIdx.x == b) {
        xcc[1] = me;
        bool ok = true;
        for (int i = 1; i <= n && ok; i++) {
            long polls = 0;
            while (ld(&f[0]) != (uint32_t)i) { if (++polls > 2000000) { ok = false; break; } }
            st(&f[64], (uint32_t)i);
        }
    }
}

int main() {
    uint32_t *f, *xcc; unsigned long long *out;
    CK(hipMalloc(&f, 1024)); CK(hipMalloc(&xcc, 8)); CK(hipMalloc(&out, 8));
    const int n = 2000;
    const int pairs[4][2] = {{0, 8}, {0, 1}, {0, 4}, {3, 11}};
    for (int rep = 0; rep < 2; rep++)
        for (auto &p : pairs) {
            CK(hipMemset(f, 0, 1024));
            pingpong<<<16, 64>>>(f, p[0], p[1], n, out, xcc);
            CK(hipDeviceSynchronize());
            unsigned long long t; uint32_t x[2];
            CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
            printf("workgroups %2d (XCD %u) <-> %2d (XCD %u): %.0f ns per round trip, %.0f ns one way\n", p[0], x[0], p[1], x[1],
                   t * 10.0 / n, t * 5.0 / n);
        }
    const int same[2][2] = {{0, 8}, {3, 11}};
    for (auto &p : same) {
        CK(hipMemset(f, 0, 1024));
        pingpong_l2<<<16, 64>>>(f, p[0], p[1], n, out, xcc);
        CK(hipDeviceSynchronize());
        unsigned long long t; uint32_t x[2];
        CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
        if (x[0] != x[1]) { printf("workgroups %d / %d are not on one XCD (%u, %u): skipped\n", p[0], p[1], x[0], x[1]); continue; }
        if (t == 0) printf("L2 (workgroup-scope RMW) %2d <-> %2d on XCD %u: no progress\n", p[0], p[1], x[0]);
        else printf("L2 (workgroup-scope RMW) %2d <-> %2d on XCD %u: %.0f ns per round trip, %.0f ns one way\n", p[0], p[1], x[0],
                    t * 10.0 / n, t * 5.0 / n);
    }
    return 0;
}
