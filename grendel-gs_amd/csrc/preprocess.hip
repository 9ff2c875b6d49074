// preprocess.hip -- K1 (per-Gaussian projection + SH colour) and K11 (its backward) for gfx950.
//
// One lane per Gaussian, 256-thread workgroups (4 waves).  Both kernels are pure HBM streaming:
// 236 B in / 44+31 B out per Gaussian forward, 552 B per Gaussian backward.
// Camera matrices are wave-uniform and are read through the scalar cache into SGPRs.
//
// What the kernels compute restates SURVEY.md Appendix A.2 / A.6 (reference call site
// gaussian_renderer/__init__.py:949-960).  The forward geometry is compiled with FP contraction
// off so that integer outputs (radii, tile rects) are reproducible against a plain C evaluation.
#include "common.h"

#include <cstdlib>

// no FMA contraction in this file: see the header comment (memory-bound kernels, no cost)
#pragma clang fp contract(off)

namespace {

__constant__ const float SH_C0 = 0.28209479177387814f;
__constant__ const float SH_C1 = 0.4886025119029199f;
__constant__ const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                     -1.0925484305920792f, 0.5462742152960396f};
__constant__ const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                     0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                     -0.5900435899266435f};

struct Cam {
    float v[16];  // viewmatrix, row-major, row-vector convention
    float p[16];  // full projection
    float c[3];   // camera centre
};

__device__ __forceinline__ Cam load_cam(const float *__restrict__ view, const float *__restrict__ proj,
                                        const float *__restrict__ campos) {
    Cam cam;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        cam.v[i] = view[i];
        cam.p[i] = proj[i];
    }
    cam.c[0] = campos[0];
    cam.c[1] = campos[1];
    cam.c[2] = campos[2];
    return cam;
}

__device__ __forceinline__ void quat_to_R(const float4 q, float R[3][3]) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[0][1] = 2.f * (x * y - r * z);
    R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);
    R[2][1] = 2.f * (y * z + r * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// T = J * Wc (2x3) with the +-1.3 tanfov clamp applied for the Jacobian only
__device__ __forceinline__ void compute_T(const float t[3], const float *v, float fx, float fy, float tanfovx,
                                          float tanfovy, float T[2][3], float tc[3], bool &xin, bool &yin) {
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    xin = !(txtz < -limx || txtz > limx);
    yin = !(tytz < -limy || tytz > limy);
    tc[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    tc[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    tc[2] = t[2];
    const float J00 = fx / tc[2], J02 = -(fx * tc[0]) / (tc[2] * tc[2]);
    const float J11 = fy / tc[2], J12 = -(fy * tc[1]) / (tc[2] * tc[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        T[0][k] = J00 * v[k * 4 + 0] + J02 * v[k * 4 + 2];
        T[1][k] = J11 * v[k * 4 + 1] + J12 * v[k * 4 + 2];
    }
}

template <int DEG>
__device__ __forceinline__ void eval_sh(const float *__restrict__ sh, float x, float y, float z, float out[3]) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float r = SH_C0 * sh[0 * 3 + c];
        if (DEG > 0) {
            r = r - SH_C1 * y * sh[1 * 3 + c] + SH_C1 * z * sh[2 * 3 + c] - SH_C1 * x * sh[3 * 3 + c];
            if (DEG > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * sh[4 * 3 + c] + SH_C2[1] * yz * sh[5 * 3 + c] +
                    SH_C2[2] * (2.f * zz - xx - yy) * sh[6 * 3 + c] + SH_C2[3] * xz * sh[7 * 3 + c] +
                    SH_C2[4] * (xx - yy) * sh[8 * 3 + c];
                if (DEG > 2) {
                    r = r + SH_C3[0] * y * (3.f * xx - yy) * sh[9 * 3 + c] + SH_C3[1] * xy * z * sh[10 * 3 + c] +
                        SH_C3[2] * y * (4.f * zz - xx - yy) * sh[11 * 3 + c] +
                        SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * sh[12 * 3 + c] +
                        SH_C3[4] * x * (4.f * zz - xx - yy) * sh[13 * 3 + c] +
                        SH_C3[5] * z * (xx - yy) * sh[14 * 3 + c] + SH_C3[6] * x * (xx - 3.f * yy) * sh[15 * 3 + c];
                }
            }
        }
        out[c] = r;
    }
}

// ------------------------------------------------------------------------------------------- K1
// RAW = true: the inputs are the RAW parameters of GaussianModel (log-scales, un-normalised quaternions,
// opacity logits, SH split into _features_dc [N,1,3] = `shs` and _features_rest [N,M-1,3] = `shs_rest`) and the
// getters' activations (scene/gaussian_model.py:109-129: exp, normalize, sigmoid, cat) are applied in
// registers -- no activated copies of the 59 floats per Gaussian are written to / re-read from HBM.
template <int DEG, bool RAW>
__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
preprocess_forward_kernel(int P, int M, const float *__restrict__ means3D, const float *__restrict__ scales,
                          float scale_modifier, const float *__restrict__ rotations, const float *__restrict__ shs,
                          const float *__restrict__ shs_rest, const float *__restrict__ opacities, const float *__restrict__ view,
                          const float *__restrict__ proj, const float *__restrict__ campos, int W, int H,
                          float tanfovx, float tanfovy, float2 *__restrict__ means2D, float *__restrict__ depths,
                          int32_t *__restrict__ radii, float *__restrict__ cov3D, float4 *__restrict__ conic_opacity,
                          float *__restrict__ rgb, uint8_t *__restrict__ clamped) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const Cam cam = load_cam(view, proj, campos);
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    const float fx = W / (2.0f * tanfovx), fy = H / (2.0f * tanfovy);

    int radius = 0;
    float2 xy = make_float2(0.f, 0.f);
    float depth = 0.f;
    float cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
    float col[3] = {0.f, 0.f, 0.f};
    bool cl[3] = {false, false, false};

    const float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    float t[3];
    t[0] = cam.v[0] * p[0] + cam.v[4] * p[1] + cam.v[8] * p[2] + cam.v[12];
    t[1] = cam.v[1] * p[0] + cam.v[5] * p[1] + cam.v[9] * p[2] + cam.v[13];
    t[2] = cam.v[2] * p[0] + cam.v[6] * p[1] + cam.v[10] * p[2] + cam.v[14];
    do {
        if (t[2] <= 0.2f) break;  // near-plane cull
        const float phx = cam.p[0] * p[0] + cam.p[4] * p[1] + cam.p[8] * p[2] + cam.p[12];
        const float phy = cam.p[1] * p[0] + cam.p[5] * p[1] + cam.p[9] * p[2] + cam.p[13];
        const float phw = cam.p[3] * p[0] + cam.p[7] * p[1] + cam.p[11] * p[2] + cam.p[15];
        const float pw = 1.0f / (phw + 0.0000001f);
        const float pprojx = phx * pw, pprojy = phy * pw;

        // Sigma = R S S R^T
        float R[3][3];
        float4 q = *reinterpret_cast<const float4 *>(rotations + 4 * (size_t)i);
        float sc[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        if (RAW) {
            const float qn = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
            q = make_float4(q.x / qn, q.y / qn, q.z / qn, q.w / qn);
            sc[0] = expf(sc[0]);
            sc[1] = expf(sc[1]);
            sc[2] = expf(sc[2]);
        }
        quat_to_R(q, R);
        const float s[3] = {scale_modifier * sc[0], scale_modifier * sc[1], scale_modifier * sc[2]};
        float L[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) L[a][b] = R[a][b] * s[b];
        float S[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) S[a][b] = L[a][0] * L[b][0] + L[a][1] * L[b][1] + L[a][2] * L[b][2];
        float T[2][3], tc[3];
        bool xin, yin;
        compute_T(t, cam.v, fx, fy, tanfovx, tanfovy, T, tc, xin, yin);
        float ST0[3], ST1[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            ST0[r] = S[r][0] * T[0][0] + S[r][1] * T[0][1] + S[r][2] * T[0][2];
            ST1[r] = S[r][0] * T[1][0] + S[r][1] * T[1][1] + S[r][2] * T[1][2];
        }
        const float a = T[0][0] * ST0[0] + T[0][1] * ST0[1] + T[0][2] * ST0[2] + 0.3f;
        const float b = T[0][0] * ST1[0] + T[0][1] * ST1[1] + T[0][2] * ST1[2];
        const float c = T[1][0] * ST1[0] + T[1][1] * ST1[1] + T[1][2] * ST1[2] + 0.3f;
        const float det = a * c - b * b;
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        const float mid = 0.5f * (a + c);
        const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const int rad = (int)ceilf(3.f * sqrtf(lam));
        const float px = ((pprojx + 1.0f) * W - 1.0f) * 0.5f;
        const float py = ((pprojy + 1.0f) * H - 1.0f) * 0.5f;
        int minx, miny, maxx, maxy;
        gsr_get_rect(px, py, rad, gx, gy, minx, miny, maxx, maxy);
        if ((maxx - minx) * (maxy - miny) == 0) break;

        // colour from SH, view direction in world space
        float d[3] = {p[0] - cam.c[0], p[1] - cam.c[1], p[2] - cam.c[2]};
        const float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        d[0] *= inv;
        d[1] *= inv;
        d[2] *= inv;
        float shl[(DEG + 1) * (DEG + 1) * 3];
        const float *shp = shs + (size_t)i * M * 3;
        if (RAW) {
            shl[0] = shs[3 * (size_t)i];
            shl[1] = shs[3 * (size_t)i + 1];
            shl[2] = shs[3 * (size_t)i + 2];
            const float *rp = shs_rest + (size_t)i * (M - 1) * 3;
#pragma unroll
            for (int k = 3; k < (DEG + 1) * (DEG + 1) * 3; k++) shl[k] = rp[k - 3];
        } else if (DEG == 3 && M == 16) {
            const float4 *s4 = reinterpret_cast<const float4 *>(shp);
#pragma unroll
            for (int k = 0; k < 12; k++) {
                const float4 v = s4[k];
                shl[4 * k] = v.x;
                shl[4 * k + 1] = v.y;
                shl[4 * k + 2] = v.z;
                shl[4 * k + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < (DEG + 1) * (DEG + 1) * 3; k++) shl[k] = shp[k];
        }
        eval_sh<DEG>(shl, d[0], d[1], d[2], col);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            col[k] += 0.5f;
            cl[k] = col[k] < 0.f;
            col[k] = fmaxf(col[k], 0.f);
        }
        radius = rad;
        xy = make_float2(px, py);
        depth = t[2];
        cov[0] = S[0][0]; cov[1] = S[0][1]; cov[2] = S[0][2];
        cov[3] = S[1][1]; cov[4] = S[1][2]; cov[5] = S[2][2];
        co = make_float4(c * det_inv, -b * det_inv, a * det_inv,
                         RAW ? 1.0f / (1.0f + expf(-opacities[i])) : opacities[i]);
    } while (false);

    radii[i] = radius;
    means2D[i] = xy;
    depths[i] = depth;
    conic_opacity[i] = co;
#pragma unroll
    for (int k = 0; k < 6; k++) cov3D[6 * (size_t)i + k] = cov[k];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        rgb[3 * (size_t)i + k] = col[k];
        clamped[3 * (size_t)i + k] = cl[k] ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------ K11
// incoming gradients: dense [P,2] / [P,4] / [P,3] (gstride == 0, vector loads) or columns of rows that are
// `gstride` floats apart -- e.g. the [P,9] gradient record K10 accumulates into (no repacking pass in between)
__device__ __forceinline__ float2 grad_ld2(const float *__restrict__ p, size_t i, int gstride) {
    if (gstride == 0) return reinterpret_cast<const float2 *>(p)[i];
    const float *q = p + (size_t)gstride * i;
    return make_float2(q[0], q[1]);
}
__device__ __forceinline__ float4 grad_ld4(const float *__restrict__ p, size_t i, int gstride) {
    if (gstride == 0) return reinterpret_cast<const float4 *>(p)[i];
    const float *q = p + (size_t)gstride * i;
    return make_float4(q[0], q[1], q[2], q[3]);
}

constexpr int REST_W = 45;  // (16 - 1) * 3 floats of _features_rest per Gaussian
// workgroup size of K11 (one lane per Gaussian): the LDS stage below costs 180 B per lane whatever the size, so the
// resident waves per CU are the same; smaller workgroups interleave their load / compute / store phases more finely
#ifndef GSR_K11_BLOCK
#define GSR_K11_BLOCK 128
#endif
constexpr int K11_BLOCK = GSR_K11_BLOCK;
constexpr int K11_WAVES_PER_EU = 3;  // (hipcc's second launch bound is WAVES PER SIMD) 180 B of LDS per lane admit
                                     // 12 waves per CU whatever the block size: <= 168 registers
#ifndef GSR_K11_UN
#define GSR_K11_UN 4
#endif
// workgroup-cooperative, coalesced copy of the block's rows of a [P, 45] array into / out of LDS.  The way in is split
// in two: ISSUE loads the lane's twelve 16-byte pieces into registers (clamped indices, no branches: all twelve loads
// are in flight together -- a copy loop of load / wait / LDS-write iterations exposed the memory latency eleven times
// per workgroup, and three workgroups of four waves per CU cannot hide that), COMMIT writes them to LDS; the caller
// puts its own per-lane loads between the two.
constexpr int REST_SLOTS = (REST_W + 3) / 4;  // 16-byte pieces per lane: ceil(45 / 4) whatever the block size
struct RestStage {
    float4 v[REST_SLOTS];
};
__device__ __forceinline__ RestStage rest_stage_issue(const float *__restrict__ g, int P) {
    const size_t row0 = (size_t)blockIdx.x * K11_BLOCK;
    const int nw = (int)min((size_t)K11_BLOCK, (size_t)P - row0) * REST_W;
    const float4 *src4 = reinterpret_cast<const float4 *>(g + row0 * REST_W);  // 46080-byte blocks: 16-byte aligned
    const int n4 = nw / 4;  // >= 11
    RestStage st;
#pragma unroll
    for (int u = 0; u < REST_SLOTS; u++) st.v[u] = src4[min((int)threadIdx.x + u * K11_BLOCK, n4 - 1)];
    return st;
}
__device__ __forceinline__ void rest_stage_commit(float *__restrict__ s_rest, RestStage &st,
                                                  const float *__restrict__ g, int P) {
    const size_t row0 = (size_t)blockIdx.x * K11_BLOCK;
    const int nw = (int)min((size_t)K11_BLOCK, (size_t)P - row0) * REST_W;
    float4 *s4 = reinterpret_cast<float4 *>(s_rest);
    // the values are pinned in registers HERE (an empty asm that claims to modify them): without it the compiler sinks
    // every load into the conditional LDS write that uses it -- load, wait, write, twelve times in a row
#pragma unroll
    for (int u = 0; u < REST_SLOTS; u++)
        asm volatile("" : "+v"(st.v[u].x), "+v"(st.v[u].y), "+v"(st.v[u].z), "+v"(st.v[u].w));
#pragma unroll
    for (int u = 0; u < REST_SLOTS; u++) {
        const int k = (int)threadIdx.x + u * K11_BLOCK;
        if (k < nw / 4) s4[k] = st.v[u];
    }
    for (int k = (nw & ~3) + threadIdx.x; k < nw; k += K11_BLOCK) s_rest[k] = g[row0 * REST_W + k];
}
__device__ __forceinline__ void rest_stage_in(float *__restrict__ s_rest, const float *__restrict__ g, int P) {
    RestStage st = rest_stage_issue(g, P);
    rest_stage_commit(s_rest, st, g, P);
}
__device__ __forceinline__ void rest_stage_out(const float *__restrict__ s_rest, float *__restrict__ g, int P) {
    const size_t row0 = (size_t)blockIdx.x * K11_BLOCK;
    const int nw = (int)min((size_t)K11_BLOCK, (size_t)P - row0) * REST_W;
    float4 *dst4 = reinterpret_cast<float4 *>(g + row0 * REST_W);
    const float4 *s4 = reinterpret_cast<const float4 *>(s_rest);
    // streaming stores: this gradient (180 of the 236 gradient bytes per Gaussian) is read exactly once, by the
    // optimizer, and written with cacheable stores it evicts the PARAMETERS from the 256 MB memory-side cache -- which
    // the optimizer and the next iteration's K1 would otherwise hit (measured: Adam 0.253 -> 0.238 ms, K11 +0.004 ms)
    typedef float vf4 __attribute__((ext_vector_type(4)));
    for (int k = threadIdx.x; k < nw / 4; k += K11_BLOCK) {
        const float4 v = s4[k];
        __builtin_nontemporal_store(vf4{v.x, v.y, v.z, v.w}, reinterpret_cast<vf4 *>(dst4 + k));
    }
    for (int k = (nw & ~3) + threadIdx.x; k < nw; k += K11_BLOCK) g[row0 * REST_W + k] = s_rest[k];
}

// Fused K11 + Adam (ADAM = true): the six parameter gradients never reach HBM.  A Gaussian's gradient is complete
// when its lane leaves the camera loop (every band's contribution arrived through the exchange before the launch), so
// the dense Adam update of scene/gaussian_model.py:292's optimizer is applied right there: the 14 per-lane floats
// (xyz, scaling, rotation, features_dc, opacity) by the lane that owns them, the workgroup's 256 x 45 floats of
// _features_rest in the coalesced order of the LDS stage-out (the gradient is read from LDS instead of being
// streamed to HBM and back).  Saves the 236 B gradient write of K11, the 236 B gradient read and (through L2) most of
// the 236 B parameter read of the optimizer per Gaussian; the arithmetic is gsr_adam1's, bit for bit
// (tests/test_gpu_loss_and_step.py::test_fused_backward_step_equals_k11_then_adam).
// Tensor order: xyz, scaling, rotation, features_dc, features_rest, opacity.
struct K11Adam {
    float *m[6], *v[6];
    float lr_c[6], b1[6], b2[6], omb1[6], omb2[6], inv_sqrt_bc2[6], eps[6];
    float grad_scale;
    // hipGraph mode (gsr_preprocess_backward_adam_raw_batched_dyn): what changes from step to step is read from DEVICE
    // memory instead of the kernel arguments, so that one captured launch serves every replay --
    //   dyn[0..5] = lr / (1 - beta1^t), dyn[6..11] = 1 / sqrt(1 - beta2^t) per tensor (the host refreshes them with an
    //   asynchronous copy in front of the replay);  skip: a word that is non-zero when an earlier kernel of the SAME
    //   replay found a capacity exceeded (tile sort / exchange slab): the step then changes nothing and the host
    //   repeats the iteration eagerly.
    const float *dyn;
    const uint32_t *skip;
};
// the per-step constants of the captured launch (wave-uniform scalar loads, once per workgroup)
__device__ __forceinline__ K11Adam k11_adam_resolve(const K11Adam &in) {
    K11Adam ad = in;
    if (in.dyn) {
#pragma unroll
        for (int t = 0; t < 6; t++) {
            ad.lr_c[t] = in.dyn[t];
            ad.inv_sqrt_bc2[t] = in.dyn[6 + t];
        }
    }
    return ad;
}

// The 14 per-lane values (xyz 0-2, scaling 3-5, rotation 6-9, features_dc 10-12, opacity 13) are updated together at
// the END of the lane's work: their 28 moment loads are then in flight at once (one exposed latency instead of five --
// the first version updated each tensor where its gradient fell out and ran 0.50 ms against 0.355 ms for the two
// kernels it replaces).  Plain (cacheable) accesses: a lane touches 4-byte pieces of 12 / 16 byte rows, neighbouring
// lanes the rest of the line -- measured: streaming loads cost the kernel +0.027 ms (the line is fetched again for the
// next piece), streaming stores alone +0.010 ms.
__device__ __forceinline__ void k11_adam_small(const K11Adam &ad, size_t i, float *__restrict__ xyz,
                                               float *__restrict__ scaling, float *__restrict__ rotation,
                                               float *__restrict__ f_dc, float *__restrict__ opacity,
                                               const float (&pv)[14], const float (&g)[14]) {
    constexpr int T[14] = {0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 5};
    constexpr int K[14] = {3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 3, 3, 3, 1};
    constexpr int C[14] = {0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 0};
    float *const base[6] = {xyz, scaling, rotation, f_dc, nullptr, opacity};
    float m[14], v[14];
#pragma unroll
    for (int e = 0; e < 14; e++) {
        m[e] = ad.m[T[e]][i * K[e] + C[e]];
        v[e] = ad.v[T[e]][i * K[e] + C[e]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 14; e++) {
        const int t = T[e];
        float p = pv[e];
        gsr_adam1(p, __fmul_rn(g[e], ad.grad_scale), m[e], v[e], ad.lr_c[t], ad.b1[t], ad.b2[t], ad.omb1[t],
                  ad.omb2[t], ad.inv_sqrt_bc2[t], ad.eps[t]);
        base[t][i * K[e] + C[e]] = p;
        ad.m[t][i * K[e] + C[e]] = m[e];
        ad.v[t][i * K[e] + C[e]] = v[e];
    }
}

// The same update with the moments of the five small tensors moved through LDS (the one-camera fused kernel, after its
// stage-out, when the _features_rest stage is free): every (tensor, moment) chunk of the workgroup -- K11_BLOCK rows of
// 12 / 16 / 4 bytes, contiguous -- is read and written as ONE 16-byte streaming access per lane, and a lane picks its
// 14 + 14 values out of LDS.  The per-lane 4-byte accesses of k11_adam_small have to stay cacheable (a streaming load
// drops the line between a lane's pieces: +0.027 ms measured), and 224 MB of cacheable moment traffic per 10^6
// Gaussians pushes the PARAMETERS out of the 256 MB memory-side cache -- the next iteration's K1 then reads them from
// HBM (0.057 -> 0.075 ms).  Through LDS the moments stream past the cache like the optimizer kernel's.
__device__ __forceinline__ void k11_adam_small_lds(float *__restrict__ lds, const K11Adam &ad, int P, int i,
                                                   float *__restrict__ xyz, float *__restrict__ scaling,
                                                   float *__restrict__ rotation, float *__restrict__ f_dc,
                                                   float *__restrict__ opacity, const float (&pv)[14],
                                                   const float (&g)[14]) {
    constexpr int NT = 5;
    constexpr int TT[NT] = {0, 1, 2, 3, 5}, KK[NT] = {3, 3, 4, 3, 1}, OFF[NT] = {0, 3, 6, 10, 13};  // x K11_BLOCK words
    constexpr int MV = 14 * K11_BLOCK;  // words of one moment of the five tensors
    typedef float vf4 __attribute__((ext_vector_type(4)));
    const size_t row0 = (size_t)blockIdx.x * K11_BLOCK;
    const int rows = (int)min((size_t)K11_BLOCK, (size_t)P - row0);
    const int e4 = 4 * (int)threadIdx.x;
    // ---- in: one 16-byte piece per lane, tensor and moment -- ten independent loads, clamped instead of branched
    // around so that they stay together (a ragged last block takes the scalar loop)
    if (rows == K11_BLOCK) {
        vf4 in[NT][2];
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int mv = 0; mv < 2; mv++) {
                const float *src = (mv ? ad.v[TT[j]] : ad.m[TT[j]]) + row0 * KK[j];
                in[j][mv] = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(src + min(e4, K11_BLOCK * KK[j] - 4)));
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int mv = 0; mv < 2; mv++)
                if (e4 < K11_BLOCK * KK[j])
                    *reinterpret_cast<vf4 *>(lds + mv * MV + OFF[j] * K11_BLOCK + e4) = in[j][mv];
    } else {
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int mv = 0; mv < 2; mv++) {
                const float *src = (mv ? ad.v[TT[j]] : ad.m[TT[j]]) + row0 * KK[j];
                for (int e = threadIdx.x; e < rows * KK[j]; e += K11_BLOCK) lds[mv * MV + OFF[j] * K11_BLOCK + e] = src[e];
            }
    }
    __syncthreads();
    // ---- the lane's 14 values
    if (i < P) {
        constexpr int T[14] = {0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 5};
        constexpr int J[14] = {0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4};
        constexpr int C[14] = {0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 0};
        float *const base[6] = {xyz, scaling, rotation, f_dc, nullptr, opacity};
#pragma unroll
        for (int e = 0; e < 14; e++) {
            const int t = T[e], k = KK[J[e]];
            float *lm = lds + OFF[J[e]] * K11_BLOCK + (int)threadIdx.x * k + C[e];
            float p = pv[e], m = lm[0], v = lm[MV];
            gsr_adam1(p, __fmul_rn(g[e], ad.grad_scale), m, v, ad.lr_c[t], ad.b1[t], ad.b2[t], ad.omb1[t], ad.omb2[t],
                      ad.inv_sqrt_bc2[t], ad.eps[t]);
            base[t][(size_t)i * k + C[e]] = p;
            lm[0] = m;
            lm[MV] = v;
        }
    }
    __syncthreads();
    // ---- out
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int mv = 0; mv < 2; mv++) {
            float *dst = (mv ? ad.v[TT[j]] : ad.m[TT[j]]) + row0 * KK[j];
            if (rows == K11_BLOCK) {
                if (e4 < K11_BLOCK * KK[j])
                    __builtin_nontemporal_store(*reinterpret_cast<const vf4 *>(lds + mv * MV + OFF[j] * K11_BLOCK + e4),
                                                reinterpret_cast<vf4 *>(dst + e4));
            } else {
                for (int e = threadIdx.x; e < rows * KK[j]; e += K11_BLOCK) dst[e] = lds[mv * MV + OFF[j] * K11_BLOCK + e];
            }
        }
}

// the workgroup's rows of _features_rest: gradient from LDS, parameter (just staged in by this workgroup: L2) and
// both moments from HBM, 16 bytes per lane and access, FOUR accesses per lane in flight before the first is used
// (three workgroups of four waves per CU do not cover the memory latency with one).
__device__ __forceinline__ void rest_adam_out(const float *__restrict__ s_rest, float *__restrict__ param,
                                              const K11Adam &ad, int P) {
    constexpr int t = 4, UN = GSR_K11_UN;
    const size_t row0 = (size_t)blockIdx.x * K11_BLOCK;
    const int nw = (int)min((size_t)K11_BLOCK, (size_t)P - row0) * REST_W;
    typedef float vf4 __attribute__((ext_vector_type(4)));
    vf4 *p4 = reinterpret_cast<vf4 *>(param + row0 * REST_W);
    vf4 *m4 = reinterpret_cast<vf4 *>(ad.m[t] + row0 * REST_W);
    vf4 *v4 = reinterpret_cast<vf4 *>(ad.v[t] + row0 * REST_W);
    const float4 *s4 = reinterpret_cast<const float4 *>(s_rest);
    const float gs = ad.grad_scale, lr_c = ad.lr_c[t], b1 = ad.b1[t], b2 = ad.b2[t], omb1 = ad.omb1[t],
                omb2 = ad.omb2[t], isb = ad.inv_sqrt_bc2[t], eps = ad.eps[t];
    const int n4 = nw / 4;  // >= 11: a block holds at least one row of 45 words
    for (int k0 = threadIdx.x; k0 < n4; k0 += K11_BLOCK * UN) {
        vf4 pn[UN], mn[UN], vn[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {  // clamped, not branched around: the loads stay together
            const int k = min(k0 + u * K11_BLOCK, n4 - 1);
            pn[u] = p4[k];
            mn[u] = __builtin_nontemporal_load(m4 + k);
            vn[u] = __builtin_nontemporal_load(v4 + k);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int k = k0 + u * K11_BLOCK;
            if (k < n4) {
                const float4 g = s4[k];
                float pa[4] = {pn[u].x, pn[u].y, pn[u].z, pn[u].w}, ma[4] = {mn[u].x, mn[u].y, mn[u].z, mn[u].w},
                      va[4] = {vn[u].x, vn[u].y, vn[u].z, vn[u].w};
                const float ga[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int c = 0; c < 4; c++)
                    gsr_adam1(pa[c], __fmul_rn(ga[c], gs), ma[c], va[c], lr_c, b1, b2, omb1, omb2, isb, eps);
                p4[k] = vf4{pa[0], pa[1], pa[2], pa[3]};
                __builtin_nontemporal_store(vf4{ma[0], ma[1], ma[2], ma[3]}, m4 + k);
                __builtin_nontemporal_store(vf4{va[0], va[1], va[2], va[3]}, v4 + k);
            }
        }
    }
    for (int k = (nw & ~3) + threadIdx.x; k < nw; k += K11_BLOCK) {
        const size_t o = row0 * REST_W + k;
        float p = param[o], m = ad.m[t][o], v = ad.v[t][o];
        gsr_adam1(p, __fmul_rn(s_rest[k], gs), m, v, lr_c, b1, b2, omb1, omb2, isb, eps);
        param[o] = p;
        ad.m[t][o] = m;
        ad.v[t][o] = v;
    }
}

// Everything one lane reads from HBM in the one-camera K11 besides its _features_rest row, loaded TOGETHER at the top of
// the kernel (unconditionally: an invisible Gaussian costs its ~130 bytes of reads): the body used to fetch each piece
// where it was first needed -- radius, then position and conic gradient, then the covariance, ... -- six dependent
// waits per lane in a kernel whose occupancy (46 KB of LDS per workgroup) cannot hide them.
struct K11In {
    int32_t rad;
    float p[3], sc[3], dc[3], op;
    float4 q;
    float cv[6];
    float4 gco;
    float2 g2;
    float grgb[3];
    uint8_t cl[3];
};
template <bool RAW>
__device__ __forceinline__ K11In k11_load(size_t i, const float *__restrict__ means3D, const float *__restrict__ scales,
                                          const float *__restrict__ rotations, const float *__restrict__ shs,
                                          const float *__restrict__ opacities_raw, const int32_t *__restrict__ radii,
                                          const float *__restrict__ cov3D, const uint8_t *__restrict__ clamped,
                                          const float *__restrict__ dL_dmeans2D,
                                          const float *__restrict__ dL_dconic_opacity,
                                          const float *__restrict__ dL_drgb, int gstride) {
    K11In in;
    in.rad = radii[i];
#pragma unroll
    for (int e = 0; e < 3; e++) {
        in.p[e] = means3D[3 * i + e];
        in.sc[e] = scales[3 * i + e];
        in.dc[e] = RAW ? shs[3 * i + e] : 0.f;
        in.cl[e] = clamped[3 * i + e];
        in.grgb[e] = dL_drgb[(gstride ? gstride : 3) * i + e];
    }
    in.op = RAW ? opacities_raw[i] : 0.f;
    in.q = *reinterpret_cast<const float4 *>(rotations + 4 * i);
#pragma unroll
    for (int e = 0; e < 6; e++) in.cv[e] = cov3D[6 * i + e];
    in.gco = grad_ld4(dL_dconic_opacity, i, gstride);
    in.g2 = grad_ld2(dL_dmeans2D, i, gstride);
    return in;
}

template <int DEG, bool RAW, bool ADAM = false>
__device__ __forceinline__ void
preprocess_backward_body(const K11In &in, float (&sp)[14], float (&sg)[14], const int i, const float *__restrict__ rest_in, float *__restrict__ rest_out, int P, int M, const float *__restrict__ means3D, const float *__restrict__ scales,
                           float scale_modifier, const float *__restrict__ rotations, const float *__restrict__ shs,
                           const float *__restrict__ shs_rest, const float *__restrict__ opacities_raw,
                           const float *__restrict__ view, const float *__restrict__ proj,
                           const float *__restrict__ campos, int W, int H, float tanfovx, float tanfovy,
                           const int32_t *__restrict__ radii, const float *__restrict__ cov3D,
                           const uint8_t *__restrict__ clamped, const float *__restrict__ dL_dmeans2D,
                           const float *__restrict__ dL_dconic_opacity, const float *__restrict__ dL_drgb, int gstride,
                           float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dscales,
                           float4 *__restrict__ dL_drotations, float *__restrict__ dL_dshs,
                           float *__restrict__ dL_dshs_rest, float *__restrict__ dL_dopacities) {
    constexpr int NC = (DEG + 1) * (DEG + 1);
    float *dsh_out = dL_dshs + (size_t)i * M * 3;

    if (ADAM && in.rad <= 0) {  // invisible: zero gradient, the moments still decay and still move the parameter
#pragma unroll
        for (int e = 0; e < 3; e++) {
            sp[e] = in.p[e];
            sp[3 + e] = in.sc[e];
            sp[10 + e] = in.dc[e];
        }
        sp[6] = in.q.x; sp[7] = in.q.y; sp[8] = in.q.z; sp[9] = in.q.w;
        sp[13] = in.op;
#pragma unroll
        for (int e = 0; e < 14; e++) sg[e] = 0.f;
        float *rp = rest_out;
        for (int k = 0; k < (M - 1) * 3; k++) rp[k] = 0.f;
        return;
    }
    if (RAW && in.rad <= 0) {
        dL_dmeans3D[3 * (size_t)i] = dL_dmeans3D[3 * (size_t)i + 1] = dL_dmeans3D[3 * (size_t)i + 2] = 0.f;
        dL_dscales[3 * (size_t)i] = dL_dscales[3 * (size_t)i + 1] = dL_dscales[3 * (size_t)i + 2] = 0.f;
        dL_drotations[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        dL_dopacities[i] = 0.f;
        dL_dshs[3 * (size_t)i] = dL_dshs[3 * (size_t)i + 1] = dL_dshs[3 * (size_t)i + 2] = 0.f;
        float *rp = rest_out;
        for (int k = 0; k < (M - 1) * 3; k++) rp[k] = 0.f;
        return;
    }
    if (in.rad <= 0) {
        dL_dmeans3D[3 * (size_t)i] = dL_dmeans3D[3 * (size_t)i + 1] = dL_dmeans3D[3 * (size_t)i + 2] = 0.f;
        dL_dscales[3 * (size_t)i] = dL_dscales[3 * (size_t)i + 1] = dL_dscales[3 * (size_t)i + 2] = 0.f;
        dL_drotations[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        dL_dopacities[i] = 0.f;
        for (int k = 0; k < M * 3; k++) dsh_out[k] = 0.f;
        return;
    }
    const Cam cam = load_cam(view, proj, campos);
    const float fx = W / (2.0f * tanfovx), fy = H / (2.0f * tanfovy);
    const float p[3] = {in.p[0], in.p[1], in.p[2]};
    const float4 gco = in.gco;
    const float gA = gco.x, gB = gco.y, gC = gco.z;
    if (ADAM) {
        const float oraw = in.op;
        const float so = 1.0f / (1.0f + expf(-oraw));
        sp[13] = oraw;
        sg[13] = gco.w * so * (1.0f - so);
    } else if (RAW) {
        const float so = 1.0f / (1.0f + expf(-in.op));
        dL_dopacities[i] = gco.w * so * (1.0f - so);
    } else {
        dL_dopacities[i] = gco.w;
    }

    // ---- conic -> cov2D -> (cov3D, t)
    float t[3];
    t[0] = cam.v[0] * p[0] + cam.v[4] * p[1] + cam.v[8] * p[2] + cam.v[12];
    t[1] = cam.v[1] * p[0] + cam.v[5] * p[1] + cam.v[9] * p[2] + cam.v[13];
    t[2] = cam.v[2] * p[0] + cam.v[6] * p[1] + cam.v[10] * p[2] + cam.v[14];
    float T[2][3], tc[3];
    bool xin, yin;
    compute_T(t, cam.v, fx, fy, tanfovx, tanfovy, T, tc, xin, yin);
    const float *cv = in.cv;
    const float S[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
    float ST0[3], ST1[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        ST0[r] = S[r][0] * T[0][0] + S[r][1] * T[0][1] + S[r][2] * T[0][2];
        ST1[r] = S[r][0] * T[1][0] + S[r][1] * T[1][1] + S[r][2] * T[1][2];
    }
    const float a = T[0][0] * ST0[0] + T[0][1] * ST0[1] + T[0][2] * ST0[2] + 0.3f;
    const float b = T[0][0] * ST1[0] + T[0][1] * ST1[1] + T[0][2] * ST1[2];
    const float c = T[1][0] * ST1[0] + T[1][1] * ST1[1] + T[1][2] * ST1[2] + 0.3f;
    const float denom = a * c - b * b;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (denom2inv != 0.f) {
        dL_da = denom2inv * (-c * c * gA + b * c * gB + (denom - a * c) * gC);
        dL_dc = denom2inv * (-a * a * gC + a * b * gB + (denom - a * c) * gA);
        dL_db = denom2inv * (2.f * b * c * gA - (denom + 2.f * b * b) * gB + 2.f * a * b * gC);
        dcov[0] = T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc;
        dcov[3] = T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc;
        dcov[5] = T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc;
        dcov[1] = 2.f * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db +
                  2.f * T[1][0] * T[1][1] * dL_dc;
        dcov[2] = 2.f * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db +
                  2.f * T[1][0] * T[1][2] * dL_dc;
        dcov[4] = 2.f * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db +
                  2.f * T[1][1] * T[1][2] * dL_dc;
    }
    float dT0[3], dT1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        dT0[k] = 2.f * ST0[k] * dL_da + ST1[k] * dL_db;
        dT1[k] = 2.f * ST1[k] * dL_dc + ST0[k] * dL_db;
    }
    const float dJ00 = dT0[0] * cam.v[0] + dT0[1] * cam.v[4] + dT0[2] * cam.v[8];
    const float dJ02 = dT0[0] * cam.v[2] + dT0[1] * cam.v[6] + dT0[2] * cam.v[10];
    const float dJ11 = dT1[0] * cam.v[1] + dT1[1] * cam.v[5] + dT1[2] * cam.v[9];
    const float dJ12 = dT1[0] * cam.v[2] + dT1[1] * cam.v[6] + dT1[2] * cam.v[10];
    const float tz = 1.f / tc[2], tz2 = tz * tz, tz3 = tz2 * tz;
    float dt[3];
    dt[0] = (xin ? 1.f : 0.f) * (-fx * tz2 * dJ02);
    dt[1] = (yin ? 1.f : 0.f) * (-fy * tz2 * dJ12);
    dt[2] = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * tc[0]) * tz3 * dJ02 + (2.f * fy * tc[1]) * tz3 * dJ12;
    float dmean[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
        dmean[k] = cam.v[k * 4 + 0] * dt[0] + cam.v[k * 4 + 1] * dt[1] + cam.v[k * 4 + 2] * dt[2];

    // ---- means2D (NDC-scaled) -> mean through the perspective divide
    {
        const float phx = cam.p[0] * p[0] + cam.p[4] * p[1] + cam.p[8] * p[2] + cam.p[12];
        const float phy = cam.p[1] * p[0] + cam.p[5] * p[1] + cam.p[9] * p[2] + cam.p[13];
        const float phw = cam.p[3] * p[0] + cam.p[7] * p[1] + cam.p[11] * p[2] + cam.p[15];
        const float mw = 1.0f / (phw + 0.0000001f);
        const float mul1 = phx * mw * mw, mul2 = phy * mw * mw;
        const float2 g2 = in.g2;
#pragma unroll
        for (int k = 0; k < 3; k++)
            dmean[k] += (cam.p[k * 4 + 0] * mw - cam.p[k * 4 + 3] * mul1) * g2.x +
                        (cam.p[k * 4 + 1] * mw - cam.p[k * 4 + 3] * mul2) * g2.y;
    }

    // ---- colour -> SH coefficients and view direction
    {
        float sh[NC * 3];
        if (RAW) {
            sh[0] = in.dc[0];
            sh[1] = in.dc[1];
            sh[2] = in.dc[2];
            const float *rp = rest_in;
#pragma unroll
            for (int k = 3; k < NC * 3; k++) sh[k] = rp[k - 3];
        } else {
            const float *sp = shs + (size_t)i * M * 3;
#pragma unroll
            for (int k = 0; k < NC * 3; k++) sh[k] = sp[k];
        }
        const float dox = p[0] - cam.c[0], doy = p[1] - cam.c[1], doz = p[2] - cam.c[2];
        const float len = sqrtf(dox * dox + doy * doy + doz * doz);
        const float x = dox / len, y = doy / len, z = doz / len;
        float ddir[3] = {0.f, 0.f, 0.f};
        float dsh[NC * 3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float g = in.cl[ch] ? 0.f : in.grgb[ch];
            float dx = 0.f, dy = 0.f, dz = 0.f;
            dsh[0 * 3 + ch] = SH_C0 * g;
            if (DEG > 0) {
                dsh[1 * 3 + ch] = -SH_C1 * y * g;
                dsh[2 * 3 + ch] = SH_C1 * z * g;
                dsh[3 * 3 + ch] = -SH_C1 * x * g;
                dx = -SH_C1 * sh[3 * 3 + ch];
                dy = -SH_C1 * sh[1 * 3 + ch];
                dz = SH_C1 * sh[2 * 3 + ch];
                if (DEG > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    dsh[4 * 3 + ch] = SH_C2[0] * xy * g;
                    dsh[5 * 3 + ch] = SH_C2[1] * yz * g;
                    dsh[6 * 3 + ch] = SH_C2[2] * (2.f * zz - xx - yy) * g;
                    dsh[7 * 3 + ch] = SH_C2[3] * xz * g;
                    dsh[8 * 3 + ch] = SH_C2[4] * (xx - yy) * g;
                    dx += SH_C2[0] * y * sh[4 * 3 + ch] + SH_C2[2] * 2.f * -x * sh[6 * 3 + ch] +
                          SH_C2[3] * z * sh[7 * 3 + ch] + SH_C2[4] * 2.f * x * sh[8 * 3 + ch];
                    dy += SH_C2[0] * x * sh[4 * 3 + ch] + SH_C2[1] * z * sh[5 * 3 + ch] +
                          SH_C2[2] * 2.f * -y * sh[6 * 3 + ch] + SH_C2[4] * 2.f * -y * sh[8 * 3 + ch];
                    dz += SH_C2[1] * y * sh[5 * 3 + ch] + SH_C2[2] * 2.f * 2.f * z * sh[6 * 3 + ch] +
                          SH_C2[3] * x * sh[7 * 3 + ch];
                    if (DEG > 2) {
                        dsh[9 * 3 + ch] = SH_C3[0] * y * (3.f * xx - yy) * g;
                        dsh[10 * 3 + ch] = SH_C3[1] * xy * z * g;
                        dsh[11 * 3 + ch] = SH_C3[2] * y * (4.f * zz - xx - yy) * g;
                        dsh[12 * 3 + ch] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                        dsh[13 * 3 + ch] = SH_C3[4] * x * (4.f * zz - xx - yy) * g;
                        dsh[14 * 3 + ch] = SH_C3[5] * z * (xx - yy) * g;
                        dsh[15 * 3 + ch] = SH_C3[6] * x * (xx - 3.f * yy) * g;
                        dx += SH_C3[0] * sh[9 * 3 + ch] * 3.f * 2.f * xy + SH_C3[1] * sh[10 * 3 + ch] * yz +
                              SH_C3[2] * sh[11 * 3 + ch] * -2.f * xy + SH_C3[3] * sh[12 * 3 + ch] * -3.f * 2.f * xz +
                              SH_C3[4] * sh[13 * 3 + ch] * (-3.f * xx + 4.f * zz - yy) +
                              SH_C3[5] * sh[14 * 3 + ch] * 2.f * xz + SH_C3[6] * sh[15 * 3 + ch] * 3.f * (xx - yy);
                        dy += SH_C3[0] * sh[9 * 3 + ch] * 3.f * (xx - yy) + SH_C3[1] * sh[10 * 3 + ch] * xz +
                              SH_C3[2] * sh[11 * 3 + ch] * (-3.f * yy + 4.f * zz - xx) +
                              SH_C3[3] * sh[12 * 3 + ch] * -3.f * 2.f * yz + SH_C3[4] * sh[13 * 3 + ch] * -2.f * xy +
                              SH_C3[5] * sh[14 * 3 + ch] * -2.f * yz + SH_C3[6] * sh[15 * 3 + ch] * -3.f * 2.f * xy;
                        dz += SH_C3[1] * sh[10 * 3 + ch] * xy + SH_C3[2] * sh[11 * 3 + ch] * 4.f * 2.f * yz +
                              SH_C3[3] * sh[12 * 3 + ch] * 3.f * (2.f * zz - xx - yy) +
                              SH_C3[4] * sh[13 * 3 + ch] * 4.f * 2.f * xz + SH_C3[5] * sh[14 * 3 + ch] * (xx - yy);
                    }
                }
            }
            ddir[0] += dx * g;
            ddir[1] += dy * g;
            ddir[2] += dz * g;
        }
        const float dot = x * ddir[0] + y * ddir[1] + z * ddir[2];
        dmean[0] += (ddir[0] - x * dot) / len;
        dmean[1] += (ddir[1] - y * dot) / len;
        dmean[2] += (ddir[2] - z * dot) / len;
        if (RAW) {
            if (ADAM) {
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    sp[10 + e] = sh[e];
                    sg[10 + e] = dsh[e];
                }
            } else {
                dL_dshs[3 * (size_t)i] = dsh[0];
                dL_dshs[3 * (size_t)i + 1] = dsh[1];
                dL_dshs[3 * (size_t)i + 2] = dsh[2];
            }
            float *rp = rest_out;
#pragma unroll
            for (int k = 3; k < NC * 3; k++) rp[k - 3] = dsh[k];
            for (int k = NC * 3; k < M * 3; k++) rp[k - 3] = 0.f;
        } else if (DEG == 3 && M == 16) {
            float4 *o4 = reinterpret_cast<float4 *>(dsh_out);
#pragma unroll
            for (int k = 0; k < 12; k++) o4[k] = make_float4(dsh[4 * k], dsh[4 * k + 1], dsh[4 * k + 2], dsh[4 * k + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < NC * 3; k++) dsh_out[k] = dsh[k];
            for (int k = NC * 3; k < M * 3; k++) dsh_out[k] = 0.f;
        }
    }
    if (ADAM) {
#pragma unroll
        for (int e = 0; e < 3; e++) {
            sp[e] = p[e];
            sg[e] = dmean[e];
        }
    } else {
        dL_dmeans3D[3 * (size_t)i] = dmean[0];
        dL_dmeans3D[3 * (size_t)i + 1] = dmean[1];
        dL_dmeans3D[3 * (size_t)i + 2] = dmean[2];
    }

    // ---- cov3D -> scales, rotations.  Sigma = M^T M, M = S R^T
    {
        const float4 qraw = in.q;
        float4 q = qraw;
        float sc[3] = {in.sc[0], in.sc[1], in.sc[2]};
        const float scraw[3] = {sc[0], sc[1], sc[2]};
        float gsc[3];
        float qn = 1.f, qnr = 1.f;
        if (RAW) {
            qnr = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
            qn = fmaxf(qnr, 1e-12f);
            q = make_float4(q.x / qn, q.y / qn, q.z / qn, q.w / qn);
            sc[0] = expf(sc[0]);
            sc[1] = expf(sc[1]);
            sc[2] = expf(sc[2]);
        }
        float R[3][3];
        quat_to_R(q, R);
        const float s[3] = {scale_modifier * sc[0], scale_modifier * sc[1], scale_modifier * sc[2]};
        float Mm[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) Mm[r][cc] = s[r] * R[cc][r];
        const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
        float dM[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++)
                dM[r][cc] = 2.f * (Mm[r][0] * dS[0][cc] + Mm[r][1] * dS[1][cc] + Mm[r][2] * dS[2][cc]);
        float dR[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            gsc[r] = (RAW ? sc[r] : 1.f) *  // d exp(x) = exp(x) dx
                scale_modifier * (R[0][r] * dM[r][0] + R[1][r] * dM[r][1] + R[2][r] * dM[r][2]);
            if (!ADAM) dL_dscales[3 * (size_t)i + r] = gsc[r];
#pragma unroll
            for (int j = 0; j < 3; j++) dR[j][r] = s[r] * dM[r][j];
        }
        if (ADAM) {
#pragma unroll
            for (int e = 0; e < 3; e++) {
                sp[3 + e] = scraw[e];
                sg[3 + e] = gsc[e];
            }
        }
        const float r_ = q.x, x = q.y, y = q.z, z = q.w;
        float4 dq;
        dq.x = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
        dq.y = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r_ * dR[1][2] + z * dR[2][0] +
                      r_ * dR[2][1] - 2.f * x * dR[2][2]);
        dq.z = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r_ * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r_ * dR[2][0] +
                      z * dR[2][1] - 2.f * y * dR[2][2]);
        dq.w = 2.f * (-2.f * z * dR[0][0] - r_ * dR[0][1] + x * dR[0][2] + r_ * dR[1][0] - 2.f * z * dR[1][1] +
                      y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
        if (RAW) {  // through q = x / max(|x|, 1e-12)
            const float dot = qnr > 1e-12f ? (q.x * dq.x + q.y * dq.y + q.z * dq.z + q.w * dq.w) : 0.f;
            dq = make_float4((dq.x - q.x * dot) / qn, (dq.y - q.y * dot) / qn, (dq.z - q.z * dot) / qn,
                             (dq.w - q.w * dot) / qn);
        }
        if (ADAM) {
            sp[6] = qraw.x; sp[7] = qraw.y; sp[8] = qraw.z; sp[9] = qraw.w;
            sg[6] = dq.x; sg[7] = dq.y; sg[8] = dq.z; sg[9] = dq.w;
        } else {
            dL_drotations[i] = dq;
        }
    }
}

// K11 kernel: with the raw parameter layout and 16 SH coefficients the workgroup's 256 x 45 floats of
// _features_rest (and of its gradient) are contiguous in HBM; per-lane rows are 180 bytes apart, so reading them
// lane by lane touches a different cache line in every lane of every load.  They are staged through LDS with
// coalesced 16-byte accesses instead (row stride 45 words: conflict-free), both ways.
template <int DEG, bool RAW>
__global__ void __launch_bounds__(K11_BLOCK, K11_WAVES_PER_EU)
preprocess_backward_kernel(int P, int M, const float *__restrict__ means3D, const float *__restrict__ scales,
                           float scale_modifier, const float *__restrict__ rotations, const float *__restrict__ shs,
                           const float *__restrict__ shs_rest, const float *__restrict__ opacities_raw,
                           const float *__restrict__ view, const float *__restrict__ proj,
                           const float *__restrict__ campos, int W, int H, float tanfovx, float tanfovy,
                           const int32_t *__restrict__ radii, const float *__restrict__ cov3D,
                           const uint8_t *__restrict__ clamped, const float *__restrict__ dL_dmeans2D,
                           const float *__restrict__ dL_dconic_opacity, const float *__restrict__ dL_drgb, int gstride,
                           float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dscales,
                           float4 *__restrict__ dL_drotations, float *__restrict__ dL_dshs,
                           float *__restrict__ dL_dshs_rest, float *__restrict__ dL_dopacities) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float ad[14], ad2[14];  // (the fused kernel's per-lane parameter / gradient slots: unused here)
    const size_t ic = (size_t)min(i, P - 1);  // lanes past the end load the last Gaussian's inputs and compute nothing
    if constexpr (RAW) {
        __shared__ float s_rest[K11_BLOCK * REST_W];
        if (M == 16) {  // block-uniform
            RestStage st = rest_stage_issue(shs_rest, P);
            const K11In in = k11_load<RAW>(ic, means3D, scales, rotations, shs, opacities_raw, radii, cov3D, clamped,
                                           dL_dmeans2D, dL_dconic_opacity, dL_drgb, gstride);
            __builtin_amdgcn_sched_barrier(0);  // every load of the lane is issued before the first result is used
            rest_stage_commit(s_rest, st, shs_rest, P);
            __syncthreads();
            if (i < P)
                preprocess_backward_body<DEG, RAW>(in, ad, ad2, i, s_rest + threadIdx.x * REST_W, s_rest + threadIdx.x * REST_W, P, M,
                                                   means3D, scales, scale_modifier, rotations, shs, shs_rest,
                                                   opacities_raw, view, proj, campos, W, H, tanfovx, tanfovy, radii,
                                                   cov3D, clamped, dL_dmeans2D, dL_dconic_opacity, dL_drgb, gstride,
                                                   dL_dmeans3D, dL_dscales, dL_drotations, dL_dshs, dL_dshs_rest,
                                                   dL_dopacities);
            __syncthreads();
            rest_stage_out(s_rest, dL_dshs_rest, P);
            return;
        }
    }
    if (i >= P) return;
    const K11In in = k11_load<RAW>(ic, means3D, scales, rotations, shs, opacities_raw, radii, cov3D, clamped, dL_dmeans2D,
                                   dL_dconic_opacity, dL_drgb, gstride);
    preprocess_backward_body<DEG, RAW>(in, ad, ad2, i, RAW ? shs_rest + (size_t)i * (M - 1) * 3 : nullptr,
                                       RAW ? dL_dshs_rest + (size_t)i * (M - 1) * 3 : nullptr, P, M, means3D, scales,
                                       scale_modifier, rotations, shs, shs_rest, opacities_raw, view, proj, campos, W, H,
                                       tanfovx, tanfovy, radii, cov3D, clamped, dL_dmeans2D, dL_dconic_opacity, dL_drgb,
                                       gstride, dL_dmeans3D, dL_dscales, dL_drotations, dL_dshs, dL_dshs_rest,
                                       dL_dopacities);
}

// K11 + Adam, one camera, 16-coefficient model (the launcher checks): the same body with the stores replaced by the
// optimizer update (K11Adam above).  Own kernel name so that traces and counters tell the fused launch apart.
template <int DEG>
__global__ void __launch_bounds__(K11_BLOCK, K11_WAVES_PER_EU)
preprocess_backward_adam_kernel(int P, float *__restrict__ xyz, float *__restrict__ scaling, float scale_modifier,
                                float *__restrict__ rotation, float *__restrict__ f_dc, float *__restrict__ f_rest,
                                float *__restrict__ opacity, const float *__restrict__ view,
                                const float *__restrict__ proj, const float *__restrict__ campos, int W, int H,
                                float tanfovx, float tanfovy, const int32_t *__restrict__ radii,
                                const float *__restrict__ cov3D, const uint8_t *__restrict__ clamped,
                                const float *__restrict__ dL_dmeans2D, const float *__restrict__ dL_dconic_opacity,
                                const float *__restrict__ dL_drgb, int gstride, const K11Adam ad_in) {
    if (ad_in.skip && *ad_in.skip) return;  // wave-uniform: a capacity overflowed earlier in this replay
    const K11Adam ad = k11_adam_resolve(ad_in);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ float s_rest[K11_BLOCK * REST_W];
    const size_t ic = (size_t)min(i, P - 1);
    RestStage st = rest_stage_issue(f_rest, P);
    const K11In in = k11_load<true>(ic, xyz, scaling, rotation, f_dc, opacity, radii, cov3D, clamped, dL_dmeans2D,
                                    dL_dconic_opacity, dL_drgb, gstride);
    __builtin_amdgcn_sched_barrier(0);  // every load of the lane is issued before the first result is used
    rest_stage_commit(s_rest, st, f_rest, P);
    __syncthreads();
    float sp[14], sg[14];
    if (i < P)
        preprocess_backward_body<DEG, true, true>(in, sp, sg, i, s_rest + threadIdx.x * REST_W,
                                                  s_rest + threadIdx.x * REST_W, P, 16, xyz, scaling, scale_modifier,
                                                  rotation, f_dc, f_rest, opacity, view, proj, campos, W, H, tanfovx,
                                                  tanfovy, radii, cov3D, clamped, dL_dmeans2D, dL_dconic_opacity,
                                                  dL_drgb, gstride, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                  nullptr);
#ifdef GSR_K11_SMALL_DIRECT
    if (i < P) k11_adam_small(ad, i, xyz, scaling, rotation, f_dc, opacity, sp, sg);
    __syncthreads();
    rest_adam_out(s_rest, f_rest, ad, P);
#else
    __syncthreads();
    rest_adam_out(s_rest, f_rest, ad, P);
    __syncthreads();  // the stage is free: the moments of the small tensors go through it
    k11_adam_small_lds(s_rest, ad, P, i, xyz, scaling, rotation, f_dc, opacity, sp, sg);
#endif
}

// ------------------------------------------------------------------------- K1 / K11, batched over cameras
// A batch of B cameras (Grendel's --bsz B; every rank projects ITS Gaussians for ALL cameras of the batch,
// gaussian_renderer/__init__.py:919-963) in ONE launch each way: a lane reads its Gaussian's 59 raw floats
// once, loops over the cameras (40 floats each, wave-uniform -> scalar loads) and, in the backward, sums the
// cameras' gradients in registers before the single store.  HBM traffic 236 + B*44 forward and
// 236 + B*79 + 236 backward per Gaussian instead of B*(236+44) and B*552, and 2 launches instead of 2B.
// cams[b] = { view[16], proj[16], campos[3], tanfovx, tanfovy, pad[3] }.
constexpr int CAM_STRIDE = 40;

__device__ __forceinline__ Cam load_cam_packed(const float *__restrict__ c) {
    Cam cam;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        cam.v[i] = c[i];
        cam.p[i] = c[16 + i];
    }
    cam.c[0] = c[32];
    cam.c[1] = c[33];
    cam.c[2] = c[34];
    return cam;
}

template <int DEG>
__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
preprocess_forward_batched_kernel(int P, int B, int M, const float *__restrict__ xyz, const float *__restrict__ scaling,
                                  float scale_modifier, const float *__restrict__ rotation,
                                  const float *__restrict__ f_dc, const float *__restrict__ f_rest,
                                  const float *__restrict__ opacity, const float *__restrict__ cams, int W, int H,
                                  float2 *__restrict__ means2D, float *__restrict__ depths,
                                  int32_t *__restrict__ radii, float *__restrict__ cov3D,
                                  float4 *__restrict__ conic_opacity, float *__restrict__ rgb,
                                  uint8_t *__restrict__ clamped) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int NC = (DEG + 1) * (DEG + 1);
    if (i >= P) return;
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    // ---- camera-independent part, once
    const float p[3] = {xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
    float4 q = *reinterpret_cast<const float4 *>(rotation + 4 * (size_t)i);
    const float qn = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    q = make_float4(q.x / qn, q.y / qn, q.z / qn, q.w / qn);
    const float s[3] = {scale_modifier * expf(scaling[3 * (size_t)i]), scale_modifier * expf(scaling[3 * (size_t)i + 1]),
                        scale_modifier * expf(scaling[3 * (size_t)i + 2])};
    const float op = 1.0f / (1.0f + expf(-opacity[i]));
    float R[3][3];
    quat_to_R(q, R);
    float L[3][3], S[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) L[a][b] = R[a][b] * s[b];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) S[a][b] = L[a][0] * L[b][0] + L[a][1] * L[b][1] + L[a][2] * L[b][2];
    cov3D[6 * (size_t)i + 0] = S[0][0]; cov3D[6 * (size_t)i + 1] = S[0][1]; cov3D[6 * (size_t)i + 2] = S[0][2];
    cov3D[6 * (size_t)i + 3] = S[1][1]; cov3D[6 * (size_t)i + 4] = S[1][2]; cov3D[6 * (size_t)i + 5] = S[2][2];
    float shl[NC * 3];
    shl[0] = f_dc[3 * (size_t)i];
    shl[1] = f_dc[3 * (size_t)i + 1];
    shl[2] = f_dc[3 * (size_t)i + 2];
    {
        // (staging these 180-byte rows through LDS as K11 does does not pay here: with a copy loop 78 -> 96 us in round 2,
        // with K11's issue / commit split and 128-lane workgroups 60.5 -> 60.0 us at 10^6 and 342 -> 343 us at 6 10^6
        // Gaussians in round 3 -- the forward has the occupancy to hide the strided rows)
        const float *rp = f_rest + (size_t)i * (M - 1) * 3;
#pragma unroll
        for (int k = 3; k < NC * 3; k++) shl[k] = rp[k - 3];
    }
    // ---- per camera
    for (int bc = 0; bc < B; bc++) {
        const float *cp = cams + (size_t)bc * CAM_STRIDE;
        const Cam cam = load_cam_packed(cp);
        const float tanfovx = cp[35], tanfovy = cp[36];
        const float fx = W / (2.0f * tanfovx), fy = H / (2.0f * tanfovy);
        int radius = 0;
        float2 xy = make_float2(0.f, 0.f);
        float depth = 0.f;
        float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
        float col[3] = {0.f, 0.f, 0.f};
        bool cl[3] = {false, false, false};
        float t[3];
        t[0] = cam.v[0] * p[0] + cam.v[4] * p[1] + cam.v[8] * p[2] + cam.v[12];
        t[1] = cam.v[1] * p[0] + cam.v[5] * p[1] + cam.v[9] * p[2] + cam.v[13];
        t[2] = cam.v[2] * p[0] + cam.v[6] * p[1] + cam.v[10] * p[2] + cam.v[14];
        do {
            if (t[2] <= 0.2f) break;
            const float phx = cam.p[0] * p[0] + cam.p[4] * p[1] + cam.p[8] * p[2] + cam.p[12];
            const float phy = cam.p[1] * p[0] + cam.p[5] * p[1] + cam.p[9] * p[2] + cam.p[13];
            const float phw = cam.p[3] * p[0] + cam.p[7] * p[1] + cam.p[11] * p[2] + cam.p[15];
            const float pw = 1.0f / (phw + 0.0000001f);
            const float pprojx = phx * pw, pprojy = phy * pw;
            float T[2][3], tc[3];
            bool xin, yin;
            compute_T(t, cam.v, fx, fy, tanfovx, tanfovy, T, tc, xin, yin);
            float ST0[3], ST1[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                ST0[r] = S[r][0] * T[0][0] + S[r][1] * T[0][1] + S[r][2] * T[0][2];
                ST1[r] = S[r][0] * T[1][0] + S[r][1] * T[1][1] + S[r][2] * T[1][2];
            }
            const float a = T[0][0] * ST0[0] + T[0][1] * ST0[1] + T[0][2] * ST0[2] + 0.3f;
            const float b = T[0][0] * ST1[0] + T[0][1] * ST1[1] + T[0][2] * ST1[2];
            const float c = T[1][0] * ST1[0] + T[1][1] * ST1[1] + T[1][2] * ST1[2] + 0.3f;
            const float det = a * c - b * b;
            if (det == 0.0f) break;
            const float det_inv = 1.f / det;
            const float mid = 0.5f * (a + c);
            const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const int rad = (int)ceilf(3.f * sqrtf(lam));
            const float px = ((pprojx + 1.0f) * W - 1.0f) * 0.5f;
            const float py = ((pprojy + 1.0f) * H - 1.0f) * 0.5f;
            int minx, miny, maxx, maxy;
            gsr_get_rect(px, py, rad, gx, gy, minx, miny, maxx, maxy);
            if ((maxx - minx) * (maxy - miny) == 0) break;
            float d[3] = {p[0] - cam.c[0], p[1] - cam.c[1], p[2] - cam.c[2]};
            const float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] *= inv; d[1] *= inv; d[2] *= inv;
            eval_sh<DEG>(shl, d[0], d[1], d[2], col);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                col[k] += 0.5f;
                cl[k] = col[k] < 0.f;
                col[k] = fmaxf(col[k], 0.f);
            }
            radius = rad;
            xy = make_float2(px, py);
            depth = t[2];
            co = make_float4(c * det_inv, -b * det_inv, a * det_inv, op);
        } while (false);
        const size_t o = (size_t)bc * P + i;
        radii[o] = radius;
        means2D[o] = xy;
        depths[o] = depth;
        conic_opacity[o] = co;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            rgb[3 * o + k] = col[k];
            clamped[3 * o + k] = cl[k] ? 1 : 0;
        }
    }
}

// What one lane reads per CAMERA in the batched K11: radius, the nine floats of the camera's [P,9] gradient record row and the
// three clamp flags.  Round 6: the loop used to fetch them where they were first needed -- radius, branch, conic gradient,
// ..., position gradient, colour gradient: three dependent trips to memory per camera in a kernel that holds two waves per
// SIMD -- and ran at 0.47 of the HBM peak at four cameras against 0.78 for the one-camera kernel.  Now a queue of
// K11_CAMQ cameras is always in flight: the first ones are requested at the top of the kernel together with the
// Gaussian's own 59 floats and the _features_rest block, camera c + K11_CAMQ while camera c is being differentiated.
#ifndef GSR_K11_CAMQ
#define GSR_K11_CAMQ 2
#endif
constexpr int K11_CAMQ = GSR_K11_CAMQ;
#ifndef GSR_K11B_WAVES
#define GSR_K11B_WAVES 2
#endif
constexpr int K11B_WAVES_PER_EU = GSR_K11B_WAVES;  // waves per SIMD the camera-batched K11 kernels are compiled for
struct K11CamSH {  // the colour part's share
    int32_t rad;
    float grgb[3];
    uint8_t cl[3];
};
struct K11CamGeo {  // the geometry part's share
    int32_t rad;
    float4 gco;
    float2 g2;
};
__device__ __forceinline__ K11CamSH k11_cam_load_sh(size_t o, const int32_t *__restrict__ radii,
                                                    const uint8_t *__restrict__ clamped,
                                                    const float *__restrict__ dL_drgb, int gstride) {
    K11CamSH c;
    c.rad = radii[o];
#pragma unroll
    for (int e = 0; e < 3; e++) {
        c.grgb[e] = dL_drgb[(gstride ? gstride : 3) * o + e];
        c.cl[e] = clamped[3 * o + e];
    }
    return c;
}
__device__ __forceinline__ K11CamGeo k11_cam_load_geo(size_t o, const int32_t *__restrict__ radii,
                                                      const float *__restrict__ dL_dmeans2D,
                                                      const float *__restrict__ dL_dconic_opacity, int gstride) {
    K11CamGeo c;
    c.rad = radii[o];
    c.gco = grad_ld4(dL_dconic_opacity, o, gstride);
    c.g2 = grad_ld2(dL_dmeans2D, o, gstride);
    return c;
}

template <int DEG, bool ADAM>
__device__ __forceinline__ void
preprocess_backward_batched_body(int P, int B, int M, const float *__restrict__ xyz,
                                   const float *__restrict__ scaling, float scale_modifier,
                                   const float *__restrict__ rotation, const float *__restrict__ f_dc,
                                   const float *__restrict__ f_rest, const float *__restrict__ opacity,
                                   const float *__restrict__ cams, int W, int H, const int32_t *__restrict__ radii,
                                   const float *__restrict__ cov3D, const uint8_t *__restrict__ clamped,
                                   const float *__restrict__ dL_dmeans2D,
                                   const float *__restrict__ dL_dconic_opacity, const float *__restrict__ dL_drgb,
                                   int gstride,
                                   float *__restrict__ dL_dxyz, float *__restrict__ dL_dscaling,
                                   float4 *__restrict__ dL_drotation, float *__restrict__ dL_ddc,
                                   float *__restrict__ dL_drest, float *__restrict__ dL_dopacity, const K11Adam &ad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ float s_rest[K11_BLOCK * REST_W];
    const bool staged = M == 16;  // _features_rest in, its gradient out: coalesced through LDS (ADAM: always)
    constexpr int NC = (DEG + 1) * (DEG + 1);
    // ---- every load of the lane that does not depend on another is issued here, before the first result is used
    // (clamped index: lanes past the end load the last Gaussian's inputs and compute nothing)
    const size_t ic = (size_t)min(i, P - 1);
    RestStage st;
    if (staged) st = rest_stage_issue(f_rest, P);
    const float p[3] = {xyz[3 * ic], xyz[3 * ic + 1], xyz[3 * ic + 2]};
    float cvr[6];
#pragma unroll
    for (int e = 0; e < 6; e++) cvr[e] = cov3D[6 * ic + e];
    const float dc[3] = {f_dc[3 * ic], f_dc[3 * ic + 1], f_dc[3 * ic + 2]};
    const float oraw = opacity[ic];
    const float4 qraw = *reinterpret_cast<const float4 *>(rotation + 4 * ic);
    const float scraw[3] = {scaling[3 * ic], scaling[3 * ic + 1], scaling[3 * ic + 2]};
    K11CamSH shq[K11_CAMQ];
    K11CamGeo geoq[K11_CAMQ];
#pragma unroll
    for (int u = 0; u < K11_CAMQ; u++) {
        const size_t o = (size_t)min(u, B - 1) * P + ic;
        shq[u] = k11_cam_load_sh(o, radii, clamped, dL_drgb, gstride);
        geoq[u] = k11_cam_load_geo(o, radii, dL_dmeans2D, dL_dconic_opacity, gstride);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (staged) {
        rest_stage_commit(s_rest, st, f_rest, P);
        __syncthreads();
    }
    float sp[14], sg[14];  // ADAM: the per-lane parameters / gradients, updated after the stage-out (below)
    if (i < P) {
    const float S[3][3] = {{cvr[0], cvr[1], cvr[2]}, {cvr[1], cvr[3], cvr[4]}, {cvr[2], cvr[4], cvr[5]}};
    // the SH coefficients above the DC term: read from the LDS stage where they are needed (staged), not held in 45
    // registers across the camera loop next to the 48 accumulators (the kernel needed 270 registers and spilled)
    const float *const shl = s_rest + threadIdx.x * REST_W;  // (an LDS address: ds_read, not a flat load)
    float sh[NC * 3];
    sh[0] = dc[0];
    sh[1] = dc[1];
    sh[2] = dc[2];
    if (!staged) {
        const float *rp = f_rest + (size_t)i * (M - 1) * 3;
#pragma unroll
        for (int k = 3; k < NC * 3; k++) sh[k] = rp[k - 3];
    }
    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dsh[NC * 3];
#pragma unroll
    for (int k = 0; k < NC * 3; k++) dsh[k] = 0.f;
    float dop = 0.f;

#define SHV(k) (staged ? shl[(k) * 3 + ch - 3] : sh[(k) * 3 + ch])
    // Two loops over the cameras (round 6): the colour part first -- its 48 accumulators and the coefficients it reads
    // from the LDS stage are dead before the geometry part starts, whose covariance chain then has the registers to
    // itself (one loop needed 270 registers and spilled; the sums per camera are unchanged, dmean's order of summation
    // is colour part of all cameras, then geometry part of all cameras).
    for (int bc = 0; bc < B; bc++) {
        const K11CamSH cin = shq[0];
#pragma unroll
        for (int u = 0; u + 1 < K11_CAMQ; u++) shq[u] = shq[u + 1];
        if (bc + K11_CAMQ < B)  // (uniform) the camera K11_CAMQ ahead: in flight while this one is differentiated
            shq[K11_CAMQ - 1] = k11_cam_load_sh((size_t)(bc + K11_CAMQ) * P + i, radii, clamped, dL_drgb, gstride);
        if (cin.rad <= 0) continue;
        if (staged) asm volatile("" ::: "memory");  // (keeps the LDS reads of the coefficients inside the loop)
        const float *cp = cams + (size_t)bc * CAM_STRIDE;
        const float camc[3] = {cp[32], cp[33], cp[34]};
        {
            const float dox = p[0] - camc[0], doy = p[1] - camc[1], doz = p[2] - camc[2];
            const float len = sqrtf(dox * dox + doy * doy + doz * doz);
            const float x = dox / len, y = doy / len, z = doz / len;
            float ddir[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float g = cin.cl[ch] ? 0.f : cin.grgb[ch];
                float dx = 0.f, dy = 0.f, dz = 0.f;
                dsh[0 * 3 + ch] += SH_C0 * g;
                if (DEG > 0) {
                    dsh[1 * 3 + ch] += -SH_C1 * y * g;
                    dsh[2 * 3 + ch] += SH_C1 * z * g;
                    dsh[3 * 3 + ch] += -SH_C1 * x * g;
                    dx = -SH_C1 * SHV(3);
                    dy = -SH_C1 * SHV(1);
                    dz = SH_C1 * SHV(2);
                    if (DEG > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        dsh[4 * 3 + ch] += SH_C2[0] * xy * g;
                        dsh[5 * 3 + ch] += SH_C2[1] * yz * g;
                        dsh[6 * 3 + ch] += SH_C2[2] * (2.f * zz - xx - yy) * g;
                        dsh[7 * 3 + ch] += SH_C2[3] * xz * g;
                        dsh[8 * 3 + ch] += SH_C2[4] * (xx - yy) * g;
                        dx += SH_C2[0] * y * SHV(4) + SH_C2[2] * 2.f * -x * SHV(6) +
                              SH_C2[3] * z * SHV(7) + SH_C2[4] * 2.f * x * SHV(8);
                        dy += SH_C2[0] * x * SHV(4) + SH_C2[1] * z * SHV(5) +
                              SH_C2[2] * 2.f * -y * SHV(6) + SH_C2[4] * 2.f * -y * SHV(8);
                        dz += SH_C2[1] * y * SHV(5) + SH_C2[2] * 2.f * 2.f * z * SHV(6) +
                              SH_C2[3] * x * SHV(7);
                        if (DEG > 2) {
                            dsh[9 * 3 + ch] += SH_C3[0] * y * (3.f * xx - yy) * g;
                            dsh[10 * 3 + ch] += SH_C3[1] * xy * z * g;
                            dsh[11 * 3 + ch] += SH_C3[2] * y * (4.f * zz - xx - yy) * g;
                            dsh[12 * 3 + ch] += SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                            dsh[13 * 3 + ch] += SH_C3[4] * x * (4.f * zz - xx - yy) * g;
                            dsh[14 * 3 + ch] += SH_C3[5] * z * (xx - yy) * g;
                            dsh[15 * 3 + ch] += SH_C3[6] * x * (xx - 3.f * yy) * g;
                            dx += SH_C3[0] * SHV(9) * 3.f * 2.f * xy + SH_C3[1] * SHV(10) * yz +
                                  SH_C3[2] * SHV(11) * -2.f * xy + SH_C3[3] * SHV(12) * -3.f * 2.f * xz +
                                  SH_C3[4] * SHV(13) * (-3.f * xx + 4.f * zz - yy) +
                                  SH_C3[5] * SHV(14) * 2.f * xz + SH_C3[6] * SHV(15) * 3.f * (xx - yy);
                            dy += SH_C3[0] * SHV(9) * 3.f * (xx - yy) + SH_C3[1] * SHV(10) * xz +
                                  SH_C3[2] * SHV(11) * (-3.f * yy + 4.f * zz - xx) +
                                  SH_C3[3] * SHV(12) * -3.f * 2.f * yz + SH_C3[4] * SHV(13) * -2.f * xy +
                                  SH_C3[5] * SHV(14) * -2.f * yz + SH_C3[6] * SHV(15) * -3.f * 2.f * xy;
                            dz += SH_C3[1] * SHV(10) * xy + SH_C3[2] * SHV(11) * 4.f * 2.f * yz +
                                  SH_C3[3] * SHV(12) * 3.f * (2.f * zz - xx - yy) +
                                  SH_C3[4] * SHV(13) * 4.f * 2.f * xz + SH_C3[5] * SHV(14) * (xx - yy);
                        }
                    }
                }
                ddir[0] += dx * g;
                ddir[1] += dy * g;
                ddir[2] += dz * g;
            }
            const float dot = x * ddir[0] + y * ddir[1] + z * ddir[2];
            dmean[0] += (ddir[0] - x * dot) / len;
            dmean[1] += (ddir[1] - y * dot) / len;
            dmean[2] += (ddir[2] - z * dot) / len;
        }
    }
#undef SHV
    // the colour gradient is complete: DC to the per-lane slots / HBM, the rest over the coefficients in the stage
    if constexpr (ADAM) {
#pragma unroll
        for (int e = 0; e < 3; e++) {
            sp[10 + e] = sh[e];
            sg[10 + e] = dsh[e];
        }
    } else {
        dL_ddc[3 * (size_t)i] = dsh[0];
        dL_ddc[3 * (size_t)i + 1] = dsh[1];
        dL_ddc[3 * (size_t)i + 2] = dsh[2];
    }
    {
        float *rp = staged ? s_rest + threadIdx.x * REST_W : dL_drest + (size_t)i * (M - 1) * 3;
#pragma unroll
        for (int k = 3; k < NC * 3; k++) rp[k - 3] = dsh[k];
        for (int k = NC * 3; k < M * 3; k++) rp[k - 3] = 0.f;
    }
    for (int bc = 0; bc < B; bc++) {
        const K11CamGeo cin = geoq[0];
#pragma unroll
        for (int u = 0; u + 1 < K11_CAMQ; u++) geoq[u] = geoq[u + 1];
        if (bc + K11_CAMQ < B)
            geoq[K11_CAMQ - 1] = k11_cam_load_geo((size_t)(bc + K11_CAMQ) * P + i, radii, dL_dmeans2D, dL_dconic_opacity,
                                                  gstride);
        if (cin.rad <= 0) continue;
        const float *cp = cams + (size_t)bc * CAM_STRIDE;
        const Cam cam = load_cam_packed(cp);
        const float tanfovx = cp[35], tanfovy = cp[36];
        const float fx = W / (2.0f * tanfovx), fy = H / (2.0f * tanfovy);
        const float4 gco = cin.gco;
        const float gA = gco.x, gB = gco.y, gC = gco.z;
        dop += gco.w;
        float t[3];
        t[0] = cam.v[0] * p[0] + cam.v[4] * p[1] + cam.v[8] * p[2] + cam.v[12];
        t[1] = cam.v[1] * p[0] + cam.v[5] * p[1] + cam.v[9] * p[2] + cam.v[13];
        t[2] = cam.v[2] * p[0] + cam.v[6] * p[1] + cam.v[10] * p[2] + cam.v[14];
        float T[2][3], tc[3];
        bool xin, yin;
        compute_T(t, cam.v, fx, fy, tanfovx, tanfovy, T, tc, xin, yin);
        float ST0[3], ST1[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            ST0[r] = S[r][0] * T[0][0] + S[r][1] * T[0][1] + S[r][2] * T[0][2];
            ST1[r] = S[r][0] * T[1][0] + S[r][1] * T[1][1] + S[r][2] * T[1][2];
        }
        const float a = T[0][0] * ST0[0] + T[0][1] * ST0[1] + T[0][2] * ST0[2] + 0.3f;
        const float b = T[0][0] * ST1[0] + T[0][1] * ST1[1] + T[0][2] * ST1[2];
        const float c = T[1][0] * ST1[0] + T[1][1] * ST1[1] + T[1][2] * ST1[2] + 0.3f;
        const float denom = a * c - b * b;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-c * c * gA + b * c * gB + (denom - a * c) * gC);
            dL_dc = denom2inv * (-a * a * gC + a * b * gB + (denom - a * c) * gA);
            dL_db = denom2inv * (2.f * b * c * gA - (denom + 2.f * b * b) * gB + 2.f * a * b * gC);
            dcov[0] += T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc;
            dcov[3] += T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc;
            dcov[5] += T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc;
            dcov[1] += 2.f * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db +
                       2.f * T[1][0] * T[1][1] * dL_dc;
            dcov[2] += 2.f * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db +
                       2.f * T[1][0] * T[1][2] * dL_dc;
            dcov[4] += 2.f * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db +
                       2.f * T[1][1] * T[1][2] * dL_dc;
        }
        float dT0[3], dT1[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            dT0[k] = 2.f * ST0[k] * dL_da + ST1[k] * dL_db;
            dT1[k] = 2.f * ST1[k] * dL_dc + ST0[k] * dL_db;
        }
        const float dJ00 = dT0[0] * cam.v[0] + dT0[1] * cam.v[4] + dT0[2] * cam.v[8];
        const float dJ02 = dT0[0] * cam.v[2] + dT0[1] * cam.v[6] + dT0[2] * cam.v[10];
        const float dJ11 = dT1[0] * cam.v[1] + dT1[1] * cam.v[5] + dT1[2] * cam.v[9];
        const float dJ12 = dT1[0] * cam.v[2] + dT1[1] * cam.v[6] + dT1[2] * cam.v[10];
        const float tz = 1.f / tc[2], tz2 = tz * tz, tz3 = tz2 * tz;
        float dt[3];
        dt[0] = (xin ? 1.f : 0.f) * (-fx * tz2 * dJ02);
        dt[1] = (yin ? 1.f : 0.f) * (-fy * tz2 * dJ12);
        dt[2] = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * tc[0]) * tz3 * dJ02 + (2.f * fy * tc[1]) * tz3 * dJ12;
#pragma unroll
        for (int k = 0; k < 3; k++)
            dmean[k] += cam.v[k * 4 + 0] * dt[0] + cam.v[k * 4 + 1] * dt[1] + cam.v[k * 4 + 2] * dt[2];
        {
            const float phx = cam.p[0] * p[0] + cam.p[4] * p[1] + cam.p[8] * p[2] + cam.p[12];
            const float phy = cam.p[1] * p[0] + cam.p[5] * p[1] + cam.p[9] * p[2] + cam.p[13];
            const float phw = cam.p[3] * p[0] + cam.p[7] * p[1] + cam.p[11] * p[2] + cam.p[15];
            const float mw = 1.0f / (phw + 0.0000001f);
            const float mul1 = phx * mw * mw, mul2 = phy * mw * mw;
            const float2 g2 = cin.g2;
#pragma unroll
            for (int k = 0; k < 3; k++)
                dmean[k] += (cam.p[k * 4 + 0] * mw - cam.p[k * 4 + 3] * mul1) * g2.x +
                            (cam.p[k * 4 + 1] * mw - cam.p[k * 4 + 3] * mul2) * g2.y;
        }
    }
    // ---- stores + the camera-independent tail (cov3D -> scale / quaternion, activations), once
    if constexpr (ADAM) {
#pragma unroll
        for (int e = 0; e < 3; e++) {
            sp[e] = p[e];
            sg[e] = dmean[e];
        }
    } else {
        dL_dxyz[3 * (size_t)i] = dmean[0];
        dL_dxyz[3 * (size_t)i + 1] = dmean[1];
        dL_dxyz[3 * (size_t)i + 2] = dmean[2];
    }
    {
        const float so = 1.0f / (1.0f + expf(-oraw));
        if constexpr (ADAM) {
            sp[13] = oraw;
            sg[13] = dop * so * (1.0f - so);
        } else {
            dL_dopacity[i] = dop * so * (1.0f - so);
        }
    }
    {
        const float qnr = sqrtf(qraw.x * qraw.x + qraw.y * qraw.y + qraw.z * qraw.z + qraw.w * qraw.w);
        const float qn = fmaxf(qnr, 1e-12f);
        const float4 q = make_float4(qraw.x / qn, qraw.y / qn, qraw.z / qn, qraw.w / qn);
        const float sc[3] = {expf(scraw[0]), expf(scraw[1]), expf(scraw[2])};
        float gsc[3];
        float R[3][3];
        quat_to_R(q, R);
        const float s[3] = {scale_modifier * sc[0], scale_modifier * sc[1], scale_modifier * sc[2]};
        float Mm[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) Mm[r][cc] = s[r] * R[cc][r];
        const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
        float dM[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++)
                dM[r][cc] = 2.f * (Mm[r][0] * dS[0][cc] + Mm[r][1] * dS[1][cc] + Mm[r][2] * dS[2][cc]);
        float dR[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            gsc[r] = sc[r] * scale_modifier * (R[0][r] * dM[r][0] + R[1][r] * dM[r][1] + R[2][r] * dM[r][2]);
            if constexpr (!ADAM) dL_dscaling[3 * (size_t)i + r] = gsc[r];
#pragma unroll
            for (int j = 0; j < 3; j++) dR[j][r] = s[r] * dM[r][j];
        }
        if constexpr (ADAM) {
#pragma unroll
            for (int e = 0; e < 3; e++) {
                sp[3 + e] = scraw[e];
                sg[3 + e] = gsc[e];
            }
        }
        const float r_ = q.x, x = q.y, y = q.z, z = q.w;
        float4 dq;
        dq.x = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
        dq.y = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r_ * dR[1][2] + z * dR[2][0] +
                      r_ * dR[2][1] - 2.f * x * dR[2][2]);
        dq.z = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r_ * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r_ * dR[2][0] +
                      z * dR[2][1] - 2.f * y * dR[2][2]);
        dq.w = 2.f * (-2.f * z * dR[0][0] - r_ * dR[0][1] + x * dR[0][2] + r_ * dR[1][0] - 2.f * z * dR[1][1] +
                      y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
        const float dot = qnr > 1e-12f ? (q.x * dq.x + q.y * dq.y + q.z * dq.z + q.w * dq.w) : 0.f;
        const float4 gq = make_float4((dq.x - q.x * dot) / qn, (dq.y - q.y * dot) / qn, (dq.z - q.z * dot) / qn,
                                      (dq.w - q.w * dot) / qn);
        if constexpr (ADAM) {
            sp[6] = qraw.x; sp[7] = qraw.y; sp[8] = qraw.z; sp[9] = qraw.w;
            sg[6] = gq.x; sg[7] = gq.y; sg[8] = gq.z; sg[9] = gq.w;
        } else {
            dL_drotation[i] = gq;
        }
    }
    }  // i < P
    if (staged) {
        __syncthreads();
        if constexpr (ADAM) {
            rest_adam_out(s_rest, const_cast<float *>(f_rest), ad, P);
            __syncthreads();  // the stage is free: the moments of the small tensors go through it
            k11_adam_small_lds(s_rest, ad, P, i, const_cast<float *>(xyz), const_cast<float *>(scaling),
                               const_cast<float *>(rotation), const_cast<float *>(f_dc), const_cast<float *>(opacity),
                               sp, sg);
        } else {
            rest_stage_out(s_rest, dL_drest, P);
        }
    }
}

template <int DEG>
__global__ void __launch_bounds__(K11_BLOCK, K11B_WAVES_PER_EU)
preprocess_backward_batched_kernel(int P, int B, int M, const float *__restrict__ xyz,
                                   const float *__restrict__ scaling, float scale_modifier,
                                   const float *__restrict__ rotation, const float *__restrict__ f_dc,
                                   const float *__restrict__ f_rest, const float *__restrict__ opacity,
                                   const float *__restrict__ cams, int W, int H, const int32_t *__restrict__ radii,
                                   const float *__restrict__ cov3D, const uint8_t *__restrict__ clamped,
                                   const float *__restrict__ dL_dmeans2D,
                                   const float *__restrict__ dL_dconic_opacity, const float *__restrict__ dL_drgb,
                                   int gstride, float *__restrict__ dL_dxyz, float *__restrict__ dL_dscaling,
                                   float4 *__restrict__ dL_drotation, float *__restrict__ dL_ddc,
                                   float *__restrict__ dL_drest, float *__restrict__ dL_dopacity) {
    preprocess_backward_batched_body<DEG, false>(P, B, M, xyz, scaling, scale_modifier, rotation, f_dc, f_rest, opacity,
                                                 cams, W, H, radii, cov3D, clamped, dL_dmeans2D, dL_dconic_opacity,
                                                 dL_drgb, gstride, dL_dxyz, dL_dscaling, dL_drotation, dL_ddc, dL_drest,
                                                 dL_dopacity, K11Adam{});
}

template <int DEG>
__global__ void __launch_bounds__(K11_BLOCK, K11B_WAVES_PER_EU)
preprocess_backward_adam_batched_kernel(int P, int B, float *__restrict__ xyz, float *__restrict__ scaling,
                                        float scale_modifier, float *__restrict__ rotation,
                                        float *__restrict__ f_dc, float *__restrict__ f_rest,
                                        float *__restrict__ opacity, const float *__restrict__ cams, int W, int H,
                                        const int32_t *__restrict__ radii, const float *__restrict__ cov3D,
                                        const uint8_t *__restrict__ clamped, const float *__restrict__ dL_dmeans2D,
                                        const float *__restrict__ dL_dconic_opacity,
                                        const float *__restrict__ dL_drgb, int gstride, const K11Adam ad_in) {
    if (ad_in.skip && *ad_in.skip) return;
    const K11Adam ad = k11_adam_resolve(ad_in);
    preprocess_backward_batched_body<DEG, true>(P, B, 16, xyz, scaling, scale_modifier, rotation, f_dc, f_rest, opacity,
                                                cams, W, H, radii, cov3D, clamped, dL_dmeans2D, dL_dconic_opacity,
                                                dL_drgb, gstride, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                ad);
}

// -------------------------------------------------------------------------------------------- K2
__global__ void __launch_bounds__(GSR_ONE_DIM_BLOCK)
local2j_kernel(int P, int W, int H, int ws, const float2 *__restrict__ means2D, const int32_t *__restrict__ radii,
               const int32_t *__restrict__ div, uint8_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    const int rad = radii[i];
    int minx = 0, miny = 0, maxx = 0, maxy = 0;
    if (rad > 0) {
        const float2 xy = means2D[i];
        gsr_get_rect(xy.x, xy.y, rad, gx, gy, minx, miny, maxx, maxy);
    }
    const bool nonempty = rad > 0 && maxx > minx && maxy > miny;
    for (int j = 0; j < ws; j++) {
        const int lo = div[j], hi = div[j + 1];
        bool hit = false;
        if (nonempty) {
            // rows of the rect whose tile-id span [y*gx+minx, y*gx+maxx-1] meets [lo,hi)
            // y*gx+minx < hi  <=> y <= (hi-1-minx)/gx ;  y*gx+maxx-1 >= lo <=> y >= ceil((lo-maxx+1)/gx)
            const int num_hi = hi - 1 - minx;
            const int y_hi = num_hi < 0 ? -1 : num_hi / gx;
            const int num_lo = lo - maxx + 1;
            const int y_lo = num_lo <= 0 ? 0 : (num_lo + gx - 1) / gx;
            hit = max(y_lo, miny) <= min(y_hi, maxy - 1);
        }
        out[(size_t)i * ws + j] = hit ? 1 : 0;
    }
}

}  // namespace

#define GSR_DISPATCH_DEG(D, ...)                 \
    switch (D) {                                 \
        case 0: { constexpr int DEG = 0; __VA_ARGS__; } break; \
        case 1: { constexpr int DEG = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int DEG = 2; __VA_ARGS__; } break; \
        default: { constexpr int DEG = 3; __VA_ARGS__; } break; \
    }

int gsr_launch_preprocess_forward(int P, int D, int M, const float *means3D, const float *scales, float scale_modifier,
                                  const float *rotations, const float *shs, const float *shs_rest,
                                  const float *opacities,
                                  const float *viewmatrix, const float *projmatrix, const float *campos, int W, int H,
                                  float tanfovx, float tanfovy, float *means2D, float *depths, int32_t *radii,
                                  float *cov3D, float *conic_opacity, float *rgb, uint8_t *clamped,
                                  hipStream_t stream) {
    if (P == 0) return 0;
    const dim3 grid(gsr_div_up(P, GSR_ONE_DIM_BLOCK)), block(GSR_ONE_DIM_BLOCK);
#define GSR_FWD(RAWF)                                                                                              \
    GSR_DISPATCH_DEG(D, hipLaunchKernelGGL((preprocess_forward_kernel<DEG, RAWF>), grid, block, 0, stream, P, M,   \
                                           means3D, scales, scale_modifier, rotations, shs, shs_rest, opacities,   \
                                           viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy,                 \
                                           reinterpret_cast<float2 *>(means2D), depths, radii, cov3D,              \
                                           reinterpret_cast<float4 *>(conic_opacity), rgb, clamped))
    if (shs_rest) { GSR_FWD(true); } else { GSR_FWD(false); }
#undef GSR_FWD
    GSR_LAUNCH_CHECK();
    return 0;
}

int gsr_launch_preprocess_backward(int P, int D, int M, const float *means3D, const float *scales,
                                   float scale_modifier, const float *rotations, const float *shs,
                                   const float *shs_rest, const float *opacities_raw, const float *viewmatrix, const float *projmatrix, const float *campos, int W,
                                   int H, float tanfovx, float tanfovy, const int32_t *radii, const float *cov3D,
                                   const uint8_t *clamped, const float *dL_dmeans2D, const float *dL_dconic_opacity,
                                   const float *dL_drgb, int grad_row_stride, float *dL_dmeans3D, float *dL_dscales, float *dL_drotations,
                                   float *dL_dshs, float *dL_dshs_rest, float *dL_dopacities,
                                   hipStream_t stream) {
    if (P == 0) return 0;
    const dim3 grid(gsr_div_up(P, K11_BLOCK)), block(K11_BLOCK);
#define GSR_BWD(RAWF)                                                                                              \
    GSR_DISPATCH_DEG(D, hipLaunchKernelGGL((preprocess_backward_kernel<DEG, RAWF>), grid, block, 0, stream, P, M,  \
                                           means3D, scales, scale_modifier, rotations, shs, shs_rest,              \
                                           opacities_raw, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy,  \
                                           radii, cov3D, clamped, dL_dmeans2D, dL_dconic_opacity, dL_drgb,           \
                                           grad_row_stride,                                                          \
                                           dL_dmeans3D, dL_dscales, reinterpret_cast<float4 *>(dL_drotations),     \
                                           dL_dshs, dL_dshs_rest, dL_dopacities))
    if (shs_rest) { GSR_BWD(true); } else { GSR_BWD(false); }
#undef GSR_BWD
    GSR_LAUNCH_CHECK();
    return 0;
}

int gsr_launch_local2j(int P, int W, int H, int ws, const float *means2D, const int32_t *radii, const int32_t *div,
                       uint8_t *out, hipStream_t stream) {
    if (P == 0) return 0;
    hipLaunchKernelGGL(local2j_kernel, dim3(gsr_div_up(P, GSR_ONE_DIM_BLOCK)), dim3(GSR_ONE_DIM_BLOCK), 0, stream, P,
                       W, H, ws, reinterpret_cast<const float2 *>(means2D), radii, div, out);
    GSR_LAUNCH_CHECK();
    return 0;
}

extern "C" int gsr_preprocess_forward_raw_batched(int P, int B, int sh_degree, int sh_coeffs, const float *xyz,
                                                  const float *scaling, float scale_modifier, const float *rotation,
                                                  const float *features_dc, const float *features_rest,
                                                  const float *opacity, const float *cams, int width, int height,
                                                  float *means2D, float *depths, int32_t *radii, float *cov3D,
                                                  float *conic_opacity, float *rgb, uint8_t *clamped,
                                                  gsr_stream_t stream) {
    if (P < 0 || B < 1 || sh_degree < 0 || sh_degree > 3 || sh_coeffs < 2 ||
        sh_coeffs < (sh_degree + 1) * (sh_degree + 1) || width <= 0 || height <= 0)
        return GSR_EINVAL;
    if (P == 0) return 0;
    if (!xyz || !scaling || !rotation || !features_dc || !features_rest || !opacity || !cams || !means2D || !depths ||
        !radii || !cov3D || !conic_opacity || !rgb || !clamped)
        return GSR_EINVAL;
    const dim3 grid(gsr_div_up(P, GSR_ONE_DIM_BLOCK)), block(GSR_ONE_DIM_BLOCK);
    GSR_DISPATCH_DEG(sh_degree,
                     hipLaunchKernelGGL(preprocess_forward_batched_kernel<DEG>, grid, block, 0,
                                        reinterpret_cast<hipStream_t>(stream), P, B, sh_coeffs, xyz, scaling,
                                        scale_modifier, rotation, features_dc, features_rest, opacity, cams, width,
                                        height, reinterpret_cast<float2 *>(means2D), depths, radii, cov3D,
                                        reinterpret_cast<float4 *>(conic_opacity), rgb, clamped));
    GSR_LAUNCH_CHECK();
    return 0;
}

extern "C" int gsr_preprocess_backward_raw_batched(int P, int B, int sh_degree, int sh_coeffs, const float *xyz,
                                                   const float *scaling, float scale_modifier, const float *rotation,
                                                   const float *features_dc, const float *features_rest,
                                                   const float *opacity, const float *cams, int width, int height,
                                                   const int32_t *radii, const float *cov3D, const uint8_t *clamped,
                                                   const float *dL_dmeans2D, const float *dL_dconic_opacity,
                                                   const float *dL_drgb, int grad_row_stride, float *dL_dxyz, float *dL_dscaling,
                                                   float *dL_drotation, float *dL_dfeatures_dc,
                                                   float *dL_dfeatures_rest, float *dL_dopacity, gsr_stream_t stream) {
    if (P < 0 || B < 1 || sh_degree < 0 || sh_degree > 3 || sh_coeffs < 2 ||
        sh_coeffs < (sh_degree + 1) * (sh_degree + 1) || width <= 0 || height <= 0)
        return GSR_EINVAL;
    if (P == 0) return 0;
    if (!xyz || !scaling || !rotation || !features_dc || !features_rest || !opacity || !cams || !radii || !cov3D ||
        !clamped || !dL_dmeans2D || !dL_dconic_opacity || !dL_drgb || !dL_dxyz || !dL_dscaling || !dL_drotation ||
        !dL_dfeatures_dc || !dL_dfeatures_rest || !dL_dopacity)
        return GSR_EINVAL;
    const dim3 grid(gsr_div_up(P, K11_BLOCK)), block(K11_BLOCK);
    GSR_DISPATCH_DEG(sh_degree,
                     hipLaunchKernelGGL(preprocess_backward_batched_kernel<DEG>, grid, block, 0,
                                        reinterpret_cast<hipStream_t>(stream), P, B, sh_coeffs, xyz, scaling,
                                        scale_modifier, rotation, features_dc, features_rest, opacity, cams, width,
                                        height, radii, cov3D, clamped, dL_dmeans2D, dL_dconic_opacity, dL_drgb,
                                        grad_row_stride, dL_dxyz,
                                        dL_dscaling, reinterpret_cast<float4 *>(dL_drotation), dL_dfeatures_dc,
                                        dL_dfeatures_rest, dL_dopacity));
    GSR_LAUNCH_CHECK();
    return 0;
}

// K11 for a batch of cameras fused with the optimizer step of the six tensors it differentiates (see K11Adam):
// xyz .. opacity are read AND updated in place, exp_avgs / exp_avg_sqs are the moments in the tensor order
// xyz, scaling, rotation, features_dc, features_rest, opacity; lrs .. steps as in gsr_adam_step_multi.  tanfov0: HOST
// pointer to { tanfovx, tanfovy } of the camera when B == 1 (selects the one-camera kernel, as
// gsr_preprocess_backward_raw does), or NULL.
extern "C" int gsr_preprocess_backward_adam_raw_batched(
    int P, int B, int sh_degree, int sh_coeffs, float *xyz, float *scaling, float scale_modifier, float *rotation,
    float *features_dc, float *features_rest, float *opacity, const float *cams, int width, int height,
    const int32_t *radii, const float *cov3D, const uint8_t *clamped, const float *dL_dmeans2D,
    const float *dL_dconic_opacity, const float *dL_drgb, int grad_row_stride, float *const *exp_avgs,
    float *const *exp_avg_sqs, const double *lrs, const double *beta1s, const double *beta2s, const double *epss,
    const int64_t *steps, float grad_scale, const float *tanfov0, gsr_stream_t stream) {
    return gsr_preprocess_backward_adam_raw_batched_dyn(
        P, B, sh_degree, sh_coeffs, xyz, scaling, scale_modifier, rotation, features_dc, features_rest, opacity, cams,
        width, height, radii, cov3D, clamped, dL_dmeans2D, dL_dconic_opacity, dL_drgb, grad_row_stride, exp_avgs,
        exp_avg_sqs, lrs, beta1s, beta2s, epss, steps, grad_scale, tanfov0, nullptr, nullptr, stream);
}

// The same launch for a captured (hipGraph) iteration: `dyn_dev` (DEVICE, 12 floats: lr / (1 - beta1^t) and
// 1 / sqrt(1 - beta2^t) of the six tensors, in the tensor order above) replaces what `lrs` / `steps` would give -- both
// may then be NULL -- and `skip_flag_dev` (DEVICE word, may be NULL) turns the whole launch into a no-op when it is
// non-zero at execution time.  With both NULL this is gsr_preprocess_backward_adam_raw_batched.
extern "C" int gsr_preprocess_backward_adam_raw_batched_dyn(
    int P, int B, int sh_degree, int sh_coeffs, float *xyz, float *scaling, float scale_modifier, float *rotation,
    float *features_dc, float *features_rest, float *opacity, const float *cams, int width, int height,
    const int32_t *radii, const float *cov3D, const uint8_t *clamped, const float *dL_dmeans2D,
    const float *dL_dconic_opacity, const float *dL_drgb, int grad_row_stride, float *const *exp_avgs,
    float *const *exp_avg_sqs, const double *lrs, const double *beta1s, const double *beta2s, const double *epss,
    const int64_t *steps, float grad_scale, const float *tanfov0, const float *dyn_dev, const uint32_t *skip_flag_dev,
    gsr_stream_t stream) {
    if (P < 0 || B < 1 || sh_degree < 0 || sh_degree > 3 || sh_coeffs != 16 || width <= 0 || height <= 0)
        return GSR_EINVAL;  // the fused step needs the LDS stage of a 16-coefficient model
    if (P == 0) return 0;
    if (!xyz || !scaling || !rotation || !features_dc || !features_rest || !opacity || !cams || !radii || !cov3D ||
        !clamped || !dL_dmeans2D || !dL_dconic_opacity || !dL_drgb || !exp_avgs || !exp_avg_sqs || !beta1s ||
        !beta2s || !epss || (!dyn_dev && (!lrs || !steps)))
        return GSR_EINVAL;
    K11Adam ad{};
    ad.dyn = dyn_dev;
    ad.skip = skip_flag_dev;
    for (int t = 0; t < 6; t++) {
        if (!exp_avgs[t] || !exp_avg_sqs[t] || (!dyn_dev && steps[t] < 1)) return GSR_EINVAL;
        const double bc1 = dyn_dev ? 1.0 : 1.0 - pow(beta1s[t], (double)steps[t]);
        const double bc2 = dyn_dev ? 1.0 : 1.0 - pow(beta2s[t], (double)steps[t]);
        ad.m[t] = exp_avgs[t];
        ad.v[t] = exp_avg_sqs[t];
        ad.lr_c[t] = dyn_dev ? 0.f : (float)(lrs[t] / bc1);
        ad.b1[t] = (float)beta1s[t];
        ad.b2[t] = (float)beta2s[t];
        ad.omb1[t] = (float)(1.0 - beta1s[t]);
        ad.omb2[t] = (float)(1.0 - beta2s[t]);
        ad.inv_sqrt_bc2[t] = (float)(1.0 / sqrt(bc2));
        ad.eps[t] = (float)epss[t];
    }
    uintptr_t al = (uintptr_t)features_rest | (uintptr_t)rotation;
    for (int t = 0; t < 6; t++) al |= (uintptr_t)ad.m[t] | (uintptr_t)ad.v[t];
    if (al & 15) return GSR_EINVAL;  // 16-byte accesses on the moments and the _features_rest block
    ad.grad_scale = grad_scale;
    const dim3 grid(gsr_div_up(P, K11_BLOCK)), block(K11_BLOCK);
    static const bool one_cam_batched = [] { const char *e = getenv("GSR_K11_ONE_BATCHED"); return e && *e == '1'; }();
    if (B == 1 && tanfov0 && !one_cam_batched) {  // one camera: the leaner kernel without accumulators
        GSR_DISPATCH_DEG(sh_degree,
                         hipLaunchKernelGGL(preprocess_backward_adam_kernel<DEG>, grid, block, 0,
                                            reinterpret_cast<hipStream_t>(stream), P, xyz, scaling, scale_modifier,
                                            rotation, features_dc, features_rest, opacity, cams, cams + 16, cams + 32,
                                            width, height, tanfov0[0], tanfov0[1], radii, cov3D, clamped, dL_dmeans2D,
                                            dL_dconic_opacity, dL_drgb, grad_row_stride, ad));
        GSR_LAUNCH_CHECK();
        return 0;
    }
    GSR_DISPATCH_DEG(sh_degree,
                     hipLaunchKernelGGL(preprocess_backward_adam_batched_kernel<DEG>, grid, block, 0,
                                        reinterpret_cast<hipStream_t>(stream), P, B, xyz, scaling, scale_modifier,
                                        rotation, features_dc, features_rest, opacity, cams, width, height, radii,
                                        cov3D, clamped, dL_dmeans2D, dL_dconic_opacity, dL_drgb, grad_row_stride, ad));
    GSR_LAUNCH_CHECK();
    return 0;
}
