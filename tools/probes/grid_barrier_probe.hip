// tools/probes/grid_barrier_probe.hip -- what does a grid-wide barrier cost on one MI355X, and which part of it?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gbp tools/probes/grid_barrier_probe.hip && /tmp/gbp
// Variants: fences (agent-scope release + acquire = L2 write-back + invalidate) on / off; a two-level arrival tree or
// one flat counter; 0 or 32 KB of plain stores per workgroup in front of every barrier (what a scatter phase leaves
// dirty in the L2); the same stores written through (sc1) instead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) {
    return __hip_atomic_load(const_cast<uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// arrival tree + RELEASE FLAGS: the workgroup that completes the root stores the epoch into one flag per group (own
// line each) and every workgroup polls its group's flag -- 32 pollers per line instead of G on the root
__device__ __forceinline__ void barrier_flags(uint32_t *leaf, uint32_t *root, uint32_t *flags, uint32_t G, uint32_t &epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch++;
        const uint32_t g = blockIdx.x / 32, ngroups = (G + 31) / 32, gsz = min(32u, G - g * 32);
        const uint32_t old = __hip_atomic_fetch_add(&leaf[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == epoch * gsz) {
            const uint32_t r = __hip_atomic_fetch_add(root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (r + 1 == epoch * ngroups)
                for (uint32_t k = 0; k < ngroups; k++)
                    __hip_atomic_store(&flags[k * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (ld_agent(&flags[g * 32]) < epoch) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(1024) probe_flags(uint32_t *leaf, uint32_t *root, uint32_t *flags, int rounds) {
    uint32_t epoch = 0;
    for (int r = 0; r < rounds; r++) barrier_flags(leaf, root, flags, gridDim.x, epoch);
}

template <bool FENCE, bool TREE>
__device__ __forceinline__ void barrier(uint32_t *leaf, uint32_t *root, uint32_t G, uint32_t &epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch++;
        if (FENCE) __threadfence();
        if (TREE) {
            const uint32_t g = blockIdx.x / 32, ngroups = (G + 31) / 32, gsz = min(32u, G - g * 32);
            const uint32_t old = __hip_atomic_fetch_add(&leaf[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == epoch * gsz) __hip_atomic_fetch_add(root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_agent(root) < epoch * ngroups) __builtin_amdgcn_s_sleep(1);
        } else {
            __hip_atomic_fetch_add(root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_agent(root) < epoch * G) __builtin_amdgcn_s_sleep(1);
        }
        if (FENCE) __threadfence();
    }
    __syncthreads();
}

// STORES: 0 none, 1 plain stores of 8 words per thread (scattered by workgroup), 2 the same written through (agent scope)
template <bool FENCE, bool TREE, int STORES>
__global__ void __launch_bounds__(1024, 1) probe(uint32_t *leaf, uint32_t *root, uint32_t *buf, int rounds, int words_per_thread) {
    uint32_t epoch = 0;
    const uint32_t G = gridDim.x;
    for (int r = 0; r < rounds; r++) {
        if (STORES) {
            for (int k = 0; k < words_per_thread; k++) {
                // workgroup w writes the slice another workgroup read last round: position rotates with r
                const size_t i = (((size_t)((blockIdx.x + r) % G) * words_per_thread + k) * blockDim.x) + threadIdx.x;
                if (STORES == 1) buf[i] = r + k;
                else __hip_atomic_store(&buf[i], (uint32_t)(r + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        barrier<FENCE, TREE>(leaf, root, G, epoch);
        if (STORES) {  // read what the neighbour wrote (checks visibility when FENCE or write-through + agent loads)
            uint32_t acc = 0;
            for (int k = 0; k < words_per_thread; k++) {
                const size_t i = (((size_t)((blockIdx.x + r + 1) % G) * words_per_thread + k) * blockDim.x) + threadIdx.x;
                acc += STORES == 1 ? buf[i] : ld_agent(&buf[i]);
            }
            if (acc != (uint32_t)(words_per_thread * r + words_per_thread * (words_per_thread - 1) / 2)) atomicAdd(&leaf[4096], 1u);
            barrier<FENCE, TREE>(leaf, root, G, epoch);  // (the slice is rewritten next round)
        }
    }
}

template <bool FENCE, bool TREE, int STORES>
int run(const char *name, int G, int threads, int rounds, int wpt, uint32_t *ctrl, uint32_t *buf) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    uint32_t errs = 0;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipMemset(ctrl, 0, 4 * 8192));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<FENCE, TREE, STORES>), dim3(G), dim3(threads), 0, 0, ctrl, ctrl + 4100, buf, rounds, wpt);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CK(hipMemcpy(&errs, ctrl + 4096, 4, hipMemcpyDeviceToHost));
    }
    const int nb = rounds * (STORES ? 2 : 1);
    printf("%-58s G=%4d x %4d  %7.2f us per barrier (%d barriers, %.1f us kernel)  stale reads %u\n", name, G, threads,
           best * 1e3f / nb, nb, best * 1e3f, errs);
    return 0;
}

int run_flags(int G, int threads, int rounds, uint32_t *ctrl) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipMemset(ctrl, 0, 4 * 8192));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe_flags, dim3(G), dim3(threads), 0, 0, ctrl, ctrl + 4100, ctrl + 4200, rounds);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-58s G=%4d x %4d  %7.2f us per barrier (%d barriers)\n", "no fences, tree + per-group release flags", G, threads,
           best * 1e3f / rounds, rounds);
    return 0;
}

int main() {
    uint32_t *ctrl, *buf;
    CK(hipMalloc(&ctrl, 4 * 8192));
    CK(hipMalloc(&buf, (size_t)1024 * 1024 * 8 * 4 * 2));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("# %s, %d CUs\n", p.name, cus);
    for (int G : {cus, cus / 4}) {
        run<true, true, 0>("fences, tree, no stores", G, 1024, 50, 0, ctrl, buf);
        run<false, true, 0>("no fences, tree, no stores", G, 1024, 50, 0, ctrl, buf);
        run<true, false, 0>("fences, flat counter, no stores", G, 1024, 50, 0, ctrl, buf);
        run<false, false, 0>("no fences, flat counter, no stores", G, 1024, 50, 0, ctrl, buf);
        run<true, true, 1>("fences, tree, 32 KB plain stores + reads per workgroup", G, 1024, 25, 8, ctrl, buf);
        run<false, true, 2>("no fences, tree, 32 KB write-through stores + agent loads", G, 1024, 25, 8, ctrl, buf);
        run<false, true, 1>("no fences, tree, 32 KB plain stores + reads (expect stale)", G, 1024, 25, 8, ctrl, buf);
    }
    run_flags(cus, 1024, 50, ctrl);
    run_flags(cus / 4, 1024, 50, ctrl);
    run_flags(cus * 2, 512, 50, ctrl);
    run_flags(cus * 4, 512, 50, ctrl);
    run<false, true, 0>("no fences, tree, no stores", cus * 4, 512, 50, 0, ctrl, buf);
    run<false, true, 2>("no fences, tree, 16 KB write-through stores + agent loads", cus * 4, 512, 25, 8, ctrl, buf);
    run<true, true, 0>("fences, tree, no stores", cus * 2, 512, 50, 0, ctrl, buf);
    run<false, true, 0>("no fences, tree, no stores", cus * 2, 512, 50, 0, ctrl, buf);
    run<true, true, 1>("fences, tree, 16 KB plain stores + reads per workgroup", cus * 2, 512, 25, 8, ctrl, buf);
    run<false, true, 2>("no fences, tree, 16 KB write-through stores + agent loads", cus * 2, 512, 25, 8, ctrl, buf);
    return 0;
}
