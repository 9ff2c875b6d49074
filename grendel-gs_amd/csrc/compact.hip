// compact.hip -- N4 row primitives behind densification / redistribution of the Gaussian shards.
//
// The reference prunes, clones, splits and redistributes Gaussians with boolean indexing on each of its ~23
// per-Gaussian tensors (6 parameters, 12 Adam moments, 5 statistics: scene/gaussian_model.py:775-921 prune /
// cat, 922-1007 split / clone, 1073-1098 one masked copy PER DESTINATION RANK per tensor).  Every
// `tensor[mask]` is a nonzero (host sync) plus an index kernel.  Here the row selection is computed ONCE:
//   gsr_group_rows : stable grouping of row indices by destination (one one-sweep radix pass, radix.h) ->
//                    order[] (rows of destination 0, then 1, ... then the dropped rows) and per-group counts;
//   gsr_gather_rows: one launch copies the selected rows of up to 32 tensors (any row width / strides), e.g.
//                    into compacted tensors (prune), into the columns of ONE record matrix that a single
//                    all-to-all-v redistributes (the reference's disabled "implementation_2",
//                    scene/gaussian_model.py:1206-1238), or back out of it.
#include "common.h"

#include "radix.h"

namespace {

constexpr int GROUP_THREADS = 256;
constexpr int GATHER_MAX_TENSORS = 32;

__global__ void __launch_bounds__(GROUP_THREADS)
group_keys_kernel(int N, int G, const int32_t *__restrict__ dest, RadixPlan plan, uint32_t *__restrict__ keys,
                  uint32_t *__restrict__ vals, uint32_t *__restrict__ ghist) {
    __shared__ uint32_t mh[RADIX_MAX_PASSES][RADIX_DIGITS];
    for (int p = 0; p < RADIX_MAX_PASSES; p++) mh[p][threadIdx.x] = 0;
    __syncthreads();
    for (long long base = (long long)blockIdx.x * blockDim.x; base < N; base += (long long)gridDim.x * blockDim.x) {
        const long long i = base + threadIdx.x;
        const bool valid = i < N;
        uint32_t key = 0;
        if (valid) {
            const int32_t d = dest[i];
            key = (d >= 0 && d < G) ? (uint32_t)d : (uint32_t)G;  // anything else: dropped, sorts last
            keys[i] = key;
            vals[i] = (uint32_t)i;
        }
        multihist_add(mh, plan, key, valid);
    }
    __syncthreads();
    multihist_flush(mh, plan, ghist);
}

__global__ void group_counts_kernel(int G, const uint32_t *__restrict__ ghist, int64_t *__restrict__ counts) {
    const int d = threadIdx.x;
    if (d > G) return;
    uint32_t c = 0;
    for (int x = 0; x < RADIX_REPLICAS; x++) c += ghist[(size_t)x * RADIX_MAX_PASSES * RADIX_DIGITS + d];
    counts[d] = (int64_t)c;
}

struct GatherArgs {
    const uint32_t *src[GATHER_MAX_TENSORS];
    uint32_t *dst[GATHER_MAX_TENSORS];
    int32_t width[GATHER_MAX_TENSORS];  // 4-byte words per row
    int64_t src_stride[GATHER_MAX_TENSORS], dst_stride[GATHER_MAX_TENSORS];  // in words
};

// blockIdx.y = tensor; one thread per 4-byte word of the selected rows.  SCATTER: dst[order[r]] = src[r] instead of
// dst[r] = src[order[r]] (the inverse of a gather with the same order: N2 writes the reduced compact rows back)
template <bool SCATTER>
__global__ void __launch_bounds__(256)
gather_rows_kernel(long long n_out, const int32_t *__restrict__ order, GatherArgs a) {
    const int k = blockIdx.y;
    const int w = a.width[k];
    const long long total = n_out * w;
    const uint32_t *__restrict__ src = a.src[k];
    uint32_t *__restrict__ dst = a.dst[k];
    const int64_t ss = a.src_stride[k], ds = a.dst_stride[k];
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / w;
        const int c = (int)(e - r * w);
        const long long sr = order ? (long long)order[r] : r;
        if (SCATTER) dst[sr * ds + c] = src[r * ss + c];
        else dst[r * ds + c] = src[sr * ss + c];
    }
}

struct GroupLayout {
    size_t ctrl, kA, vA, kB, total;
    CtrlLayout C;
};
GroupLayout group_layout(long long N) {
    GroupLayout L;
    size_t o = 0;
    L.ctrl = o;
    L.C = ctrl_layout(N, 1, false);
    o += L.C.total;
    const size_t np = align_up((size_t)(N + 1) * 4);
    L.kA = o; o += np;
    L.vA = o; o += np;
    L.kB = o; o += np;
    L.total = o;
    return L;
}
}  // namespace

// The per-iteration densification statistics (densification.py:13-25 of the reference, after every backward of the
// densification phase -- half of a 30 000-iteration run): for the Gaussians visible in the view (radius > 0)
//   max_radii2D = max(max_radii2D, radius);  xyz_gradient_accum += |d loss / d means2D|;  denom += 1
// as ONE launch over the P rows instead of the reference's boolean indexing (two `nonzero` host syncs, six gather / scatter
// kernels).  `grad`: the means2D gradient, rows `grad_stride` floats apart (2 for a dense [P,2]; 9 for the view of K10's
// [P,9] record the operator hands out).  An invisible Gaussian's gradient row is exactly 0 (K10 never touches it), so the
// accumulators of invisible rows keep their bits whether or not the row is skipped; it IS skipped (radius 0).
__global__ void __launch_bounds__(256)
densify_stats_kernel(long long P, const int32_t *__restrict__ radii, const float *__restrict__ grad,
                     long long grad_stride, float *__restrict__ max_radii2D, float *__restrict__ accum,
                     float *__restrict__ denom) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int32_t r = radii[i];
    if (r <= 0) return;
    const float gx = grad[i * grad_stride], gy = grad[i * grad_stride + 1];
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
    accum[i] += sqrtf(__builtin_fmaf(gy, gy, __fmul_rn(gx, gx)));
    denom[i] += 1.0f;
}

extern "C" int gsr_densify_stats(int64_t P, const int32_t *radii, const float *grad, int64_t grad_stride,
                                 float *max_radii2D, float *accum, float *denom, gsr_stream_t stream) {
    if (P < 0 || grad_stride < 2) return GSR_EINVAL;
    if (P == 0) return 0;
    if (!radii || !grad || !max_radii2D || !accum || !denom) return GSR_EINVAL;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), (long long)P, radii, grad, (long long)grad_stride,
                       max_radii2D, accum, denom);
    GSR_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t gsr_group_rows_bytes(int64_t N) {
    if (N < 0) return 0;
    return group_layout(N).total;
}

extern "C" int gsr_group_rows(int64_t N, int G, const int32_t *dest, int32_t *order, int64_t *counts, void *workspace,
                              size_t workspace_bytes, gsr_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (N < 0 || G < 1 || G > 255 || !counts) return GSR_EINVAL;
    GSR_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)(G + 1), stream));
    if (N == 0) return 0;
    if (!dest || !order || !workspace) return GSR_EINVAL;
    if (N > RADIX_MAX_N) return GSR_EINVAL;
    const GroupLayout L = group_layout(N);
    if (workspace_bytes < L.total) return GSR_ENOSPACE;
    char *base = reinterpret_cast<char *>(workspace);
    char *ctrl = base + L.ctrl;
    uint32_t *kA = reinterpret_cast<uint32_t *>(base + L.kA), *vA = reinterpret_cast<uint32_t *>(base + L.vA);
    uint32_t *kB = reinterpret_cast<uint32_t *>(base + L.kB);
    GSR_HIP(hipMemsetAsync(ctrl, 0, L.C.total, stream));
    const RadixPlan plan = radix_plan(0, 8);
    const int grid = gsr_div_up(N, GROUP_THREADS) < 512 ? gsr_div_up(N, GROUP_THREADS) : 512;
    uint32_t *ghist = reinterpret_cast<uint32_t *>(ctrl + L.C.ghist);
    hipLaunchKernelGGL(group_keys_kernel, dim3(grid), dim3(GROUP_THREADS), 0, stream, (int)N, G, dest, plan, kA, vA,
                       ghist);
    hipLaunchKernelGGL(group_counts_kernel, dim3(1), dim3(256), 0, stream, G, ghist, counts);
    int in_first = 1;
    // single pass: the values (row indices) land directly in `order`; vB is never written
    int rc = radix_sort_pairs(kA, vA, kB, nullptr, N, plan, ctrl, L.C, &in_first, stream,
                              reinterpret_cast<uint32_t *>(order));
    if (rc) return rc;
    GSR_LAUNCH_CHECK();
    return 0;
}

namespace {
int launch_rows(bool scatter, int64_t n_out, const int32_t *order, int num_tensors, const void *const *srcs,
                void *const *dsts, const int32_t *widths, const int64_t *src_strides, const int64_t *dst_strides,
                hipStream_t stream) {
    if (n_out < 0 || num_tensors < 0 || num_tensors > GATHER_MAX_TENSORS) return GSR_EINVAL;
    if (n_out == 0 || num_tensors == 0) return 0;
    if (!srcs || !dsts || !widths || !src_strides || !dst_strides) return GSR_EINVAL;
    GatherArgs a{};
    long long widest = 0;
    for (int k = 0; k < num_tensors; k++) {
        if (!srcs[k] || !dsts[k] || widths[k] <= 0) return GSR_EINVAL;
        a.src[k] = reinterpret_cast<const uint32_t *>(srcs[k]);
        a.dst[k] = reinterpret_cast<uint32_t *>(dsts[k]);
        a.width[k] = widths[k];
        a.src_stride[k] = src_strides[k];
        a.dst_stride[k] = dst_strides[k];
        widest = widths[k] > widest ? widths[k] : widest;
    }
    long long blocks = (n_out * widest + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (scatter)
        hipLaunchKernelGGL(gather_rows_kernel<true>, dim3((unsigned)blocks, (unsigned)num_tensors), dim3(256), 0, stream,
                           (long long)n_out, order, a);
    else
        hipLaunchKernelGGL(gather_rows_kernel<false>, dim3((unsigned)blocks, (unsigned)num_tensors), dim3(256), 0,
                           stream, (long long)n_out, order, a);
    GSR_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int gsr_gather_rows(int64_t n_out, const int32_t *order, int num_tensors, const void *const *srcs,
                               void *const *dsts, const int32_t *widths, const int64_t *src_strides,
                               const int64_t *dst_strides, gsr_stream_t stream_) {
    return launch_rows(false, n_out, order, num_tensors, srcs, dsts, widths, src_strides, dst_strides,
                       reinterpret_cast<hipStream_t>(stream_));
}

extern "C" int gsr_scatter_rows(int64_t n_in, const int32_t *order, int num_tensors, const void *const *srcs,
                                void *const *dsts, const int32_t *widths, const int64_t *src_strides,
                                const int64_t *dst_strides, gsr_stream_t stream_) {
    if (n_in > 0 && !order) return GSR_EINVAL;
    return launch_rows(true, n_in, order, num_tensors, srcs, dsts, widths, src_strides, dst_strides,
                       reinterpret_cast<hipStream_t>(stream_));
}
