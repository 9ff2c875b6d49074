// api.hip -- the extern "C" entry points declared in include/gsraster.h (argument checking +
// dispatch to the kernels' launchers).  No torch types cross this boundary.
#include "common.h"

extern "C" {

const char *gsr_error_string(int code) {
    if (code == 0) return "success";
    if (code == GSR_EINVAL) return "gsraster: invalid argument";
    if (code == GSR_ENOSPACE) return "gsraster: workspace too small";
    if (code == GSR_ERETRY) return "gsraster: pair count valid, bounded sort must be repeated";
    if (code == GSR_EFAULT)
        return "gsraster: a persistent binning kernel of an earlier call timed out at a grid barrier (hung or preempted "
               "device); that view's lists are incomplete; the look-back pipeline is used from now on";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "gsraster: unknown error";
}

int gsr_abi_version(void) { return 13; }

int gsr_get_block_xy(int *block_x, int *block_y, int *one_dim_block) {
    if (!block_x || !block_y || !one_dim_block) return GSR_EINVAL;
    *block_x = GSR_BLOCK_X;
    *block_y = GSR_BLOCK_Y;
    *one_dim_block = GSR_ONE_DIM_BLOCK;
    return 0;
}

int gsr_preprocess_forward(int P, int sh_degree, int sh_coeffs, const float *means3D, const float *scales,
                           float scale_modifier, const float *rotations, const float *shs, const float *opacities,
                           const float *viewmatrix, const float *projmatrix, const float *campos, int width,
                           int height, float tanfovx, float tanfovy, float *means2D, float *depths, int32_t *radii,
                           float *cov3D, float *conic_opacity, float *rgb, uint8_t *clamped, gsr_stream_t stream) {
    if (P < 0 || sh_degree < 0 || sh_degree > 3 || sh_coeffs < (sh_degree + 1) * (sh_degree + 1) || width <= 0 ||
        height <= 0 || !(tanfovx > 0.f) || !(tanfovy > 0.f))
        return GSR_EINVAL;
    if (P == 0) return 0;
    if (!means3D || !scales || !rotations || !shs || !opacities || !viewmatrix || !projmatrix || !campos ||
        !means2D || !depths || !radii || !cov3D || !conic_opacity || !rgb || !clamped)
        return GSR_EINVAL;
    return gsr_launch_preprocess_forward(P, sh_degree, sh_coeffs, means3D, scales, scale_modifier, rotations, shs,
                                         nullptr, opacities, viewmatrix, projmatrix, campos, width, height, tanfovx, tanfovy,
                                         means2D, depths, radii, cov3D, conic_opacity, rgb, clamped,
                                         reinterpret_cast<hipStream_t>(stream));
}

int gsr_preprocess_backward(int P, int sh_degree, int sh_coeffs, const float *means3D, const float *scales,
                            float scale_modifier, const float *rotations, const float *shs, const float *viewmatrix,
                            const float *projmatrix, const float *campos, int width, int height, float tanfovx,
                            float tanfovy, const int32_t *radii, const float *cov3D, const uint8_t *clamped,
                            const float *dL_dmeans2D, const float *dL_dconic_opacity, const float *dL_drgb,
                            int grad_row_stride, float *dL_dmeans3D, float *dL_dscales, float *dL_drotations,
                            float *dL_dshs, float *dL_dopacities, gsr_stream_t stream) {
    if (P < 0 || sh_degree < 0 || sh_degree > 3 || sh_coeffs < (sh_degree + 1) * (sh_degree + 1) || width <= 0 ||
        height <= 0 || !(tanfovx > 0.f) || !(tanfovy > 0.f) || grad_row_stride < 0)
        return GSR_EINVAL;
    if (P == 0) return 0;
    if (!means3D || !scales || !rotations || !shs || !viewmatrix || !projmatrix || !campos || !radii || !cov3D ||
        !clamped || !dL_dmeans2D || !dL_dconic_opacity || !dL_drgb || !dL_dmeans3D || !dL_dscales || !dL_drotations ||
        !dL_dshs || !dL_dopacities)
        return GSR_EINVAL;
    return gsr_launch_preprocess_backward(P, sh_degree, sh_coeffs, means3D, scales, scale_modifier, rotations, shs,
                                          nullptr, nullptr, viewmatrix, projmatrix, campos, width, height, tanfovx, tanfovy, radii,
                                          cov3D, clamped, dL_dmeans2D, dL_dconic_opacity, dL_drgb, grad_row_stride,
                                          dL_dmeans3D, dL_dscales, dL_drotations, dL_dshs, nullptr, dL_dopacities,
                                          reinterpret_cast<hipStream_t>(stream));
}

int gsr_preprocess_forward_raw(int P, int sh_degree, int sh_coeffs, const float *xyz, const float *scaling,
                               float scale_modifier, const float *rotation, const float *features_dc,
                               const float *features_rest, const float *opacity, const float *viewmatrix,
                               const float *projmatrix, const float *campos, int width, int height, float tanfovx,
                               float tanfovy, float *means2D, float *depths, int32_t *radii, float *cov3D,
                               float *conic_opacity, float *rgb, uint8_t *clamped, gsr_stream_t stream) {
    if (P < 0 || sh_degree < 0 || sh_degree > 3 || sh_coeffs < 2 || sh_coeffs < (sh_degree + 1) * (sh_degree + 1) ||
        width <= 0 || height <= 0 || !(tanfovx > 0.f) || !(tanfovy > 0.f))
        return GSR_EINVAL;
    if (P == 0) return 0;
    if (!xyz || !scaling || !rotation || !features_dc || !features_rest || !opacity || !viewmatrix || !projmatrix ||
        !campos || !means2D || !depths || !radii || !cov3D || !conic_opacity || !rgb || !clamped)
        return GSR_EINVAL;
    return gsr_launch_preprocess_forward(P, sh_degree, sh_coeffs, xyz, scaling, scale_modifier, rotation, features_dc,
                                         features_rest, opacity, viewmatrix, projmatrix, campos, width, height,
                                         tanfovx, tanfovy, means2D, depths, radii, cov3D, conic_opacity, rgb, clamped,
                                         reinterpret_cast<hipStream_t>(stream));
}

int gsr_preprocess_backward_raw(int P, int sh_degree, int sh_coeffs, const float *xyz, const float *scaling,
                                float scale_modifier, const float *rotation, const float *features_dc,
                                const float *features_rest, const float *opacity, const float *viewmatrix,
                                const float *projmatrix, const float *campos, int width, int height, float tanfovx,
                                float tanfovy, const int32_t *radii, const float *cov3D, const uint8_t *clamped,
                                const float *dL_dmeans2D, const float *dL_dconic_opacity, const float *dL_drgb,
                                int grad_row_stride, float *dL_dxyz, float *dL_dscaling, float *dL_drotation,
                                float *dL_dfeatures_dc, float *dL_dfeatures_rest, float *dL_dopacity,
                                gsr_stream_t stream) {
    if (P < 0 || sh_degree < 0 || sh_degree > 3 || sh_coeffs < 2 || sh_coeffs < (sh_degree + 1) * (sh_degree + 1) ||
        width <= 0 || height <= 0 || !(tanfovx > 0.f) || !(tanfovy > 0.f) || grad_row_stride < 0)
        return GSR_EINVAL;
    if (P == 0) return 0;
    if (!xyz || !scaling || !rotation || !features_dc || !features_rest || !opacity || !viewmatrix || !projmatrix ||
        !campos || !radii || !cov3D || !clamped || !dL_dmeans2D || !dL_dconic_opacity || !dL_drgb || !dL_dxyz ||
        !dL_dscaling || !dL_drotation || !dL_dfeatures_dc || !dL_dfeatures_rest || !dL_dopacity)
        return GSR_EINVAL;
    return gsr_launch_preprocess_backward(P, sh_degree, sh_coeffs, xyz, scaling, scale_modifier, rotation, features_dc,
                                          features_rest, opacity, viewmatrix, projmatrix, campos, width, height,
                                          tanfovx, tanfovy, radii, cov3D, clamped, dL_dmeans2D, dL_dconic_opacity,
                                          dL_drgb, grad_row_stride, dL_dxyz, dL_dscaling, dL_drotation, dL_dfeatures_dc,
                                          dL_dfeatures_rest, dL_dopacity, reinterpret_cast<hipStream_t>(stream));
}

int gsr_get_local2j_ids_bool(int P, int width, int height, int world_size, const float *means2D,
                             const int32_t *radii, const int32_t *dist_global_strategy, uint8_t *out,
                             gsr_stream_t stream) {
    if (P < 0 || width <= 0 || height <= 0 || world_size <= 0) return GSR_EINVAL;
    if (P == 0) return 0;
    if (!means2D || !radii || !dist_global_strategy || !out) return GSR_EINVAL;
    return gsr_launch_local2j(P, width, height, world_size, means2D, radii, dist_global_strategy, out,
                              reinterpret_cast<hipStream_t>(stream));
}

int gsr_render_forward(int P, int width, int height, const int32_t *ranges, const uint32_t *point_list,
                       const float *means2D, const float *conic_opacity, const float *rgb,
                       const uint8_t *compute_locally, const float *bg, float *out_color, float *final_T,
                       int32_t *n_contrib, gsr_stream_t stream) {
    if (P < 0 || width <= 0 || height <= 0) return GSR_EINVAL;
    if (!ranges || !compute_locally || !bg || !out_color || !final_T || !n_contrib) return GSR_EINVAL;
    if (P > 0 && (!means2D || !conic_opacity || !rgb)) return GSR_EINVAL;
    return gsr_launch_composite_forward(P, width, height, ranges, point_list, means2D, conic_opacity, rgb,
                                        compute_locally, bg, out_color, final_T, n_contrib, nullptr, 0, 0, 0, nullptr, 0,
                                        reinterpret_cast<hipStream_t>(stream));
}

size_t gsr_render_seg_bytes(int width, int height) {
    if (width <= 0 || height <= 0) return 0;
    return gsr_composite_seg_bytes(width, height);
}

int gsr_render_forward_seg(int P, int width, int height, const int32_t *ranges, const uint32_t *point_list,
                           const float *means2D, const float *conic_opacity, const float *rgb,
                           const uint8_t *compute_locally, const float *bg, float *out_color, float *final_T,
                           int32_t *n_contrib, void *seg_ws, size_t seg_bytes, int row_lo, int row_hi,
                           gsr_stream_t stream) {
    if (P < 0 || width <= 0 || height <= 0) return GSR_EINVAL;
    if (!ranges || !compute_locally || !bg || !out_color || !final_T || !n_contrib) return GSR_EINVAL;
    if (P > 0 && (!means2D || !conic_opacity || !rgb)) return GSR_EINVAL;
    return gsr_launch_composite_forward(P, width, height, ranges, point_list, means2D, conic_opacity, rgb,
                                        compute_locally, bg, out_color, final_T, n_contrib, seg_ws, seg_bytes, row_lo,
                                        row_hi, nullptr, 0, reinterpret_cast<hipStream_t>(stream));
}

int gsr_render_forward_seg_z(int P, int width, int height, const int32_t *ranges, const uint32_t *point_list,
                             const float *means2D, const float *conic_opacity, const float *rgb,
                             const uint8_t *compute_locally, const float *bg, float *out_color, float *final_T,
                             int32_t *n_contrib, void *seg_ws, size_t seg_bytes, int row_lo, int row_hi, void *zero_ptr,
                             size_t zero_bytes, gsr_stream_t stream) {
    if (P < 0 || width <= 0 || height <= 0) return GSR_EINVAL;
    if (!ranges || !compute_locally || !bg || !out_color || !final_T || !n_contrib) return GSR_EINVAL;
    if (P > 0 && (!means2D || !conic_opacity || !rgb)) return GSR_EINVAL;
    return gsr_launch_composite_forward(P, width, height, ranges, point_list, means2D, conic_opacity, rgb,
                                        compute_locally, bg, out_color, final_T, n_contrib, seg_ws, seg_bytes, row_lo,
                                        row_hi, zero_ptr, zero_bytes, reinterpret_cast<hipStream_t>(stream));
}

int gsr_render_backward_seg(int P, int width, int height, const int32_t *ranges, const uint32_t *point_list,
                            const float *means2D, const float *conic_opacity, const float *rgb,
                            const uint8_t *compute_locally, const float *bg, const float *final_T,
                            const int32_t *n_contrib, const float *dL_dpixels, float *dL_record,
                            const float *out_color, void *seg_ws, size_t seg_bytes, int row_lo, int row_hi,
                            gsr_stream_t stream) {
    if (P < 0 || width <= 0 || height <= 0) return GSR_EINVAL;
    if (P == 0) return 0;
    if (!ranges || !compute_locally || !bg || !final_T || !n_contrib || !dL_dpixels || !means2D || !conic_opacity ||
        !rgb || !dL_record)
        return GSR_EINVAL;
    return gsr_launch_composite_backward(P, width, height, ranges, point_list, means2D, conic_opacity, rgb,
                                         compute_locally, bg, final_T, n_contrib, dL_dpixels, dL_record, out_color,
                                         seg_ws, seg_bytes, row_lo, row_hi, 0, reinterpret_cast<hipStream_t>(stream));
}

int gsr_render_backward_seg_z(int P, int width, int height, const int32_t *ranges, const uint32_t *point_list,
                              const float *means2D, const float *conic_opacity, const float *rgb,
                              const uint8_t *compute_locally, const float *bg, const float *final_T,
                              const int32_t *n_contrib, const float *dL_dpixels, float *dL_record,
                              const float *out_color, void *seg_ws, size_t seg_bytes, int row_lo, int row_hi,
                              int record_is_zero, gsr_stream_t stream) {
    if (P < 0 || width <= 0 || height <= 0) return GSR_EINVAL;
    if (P == 0) return 0;
    if (!ranges || !compute_locally || !bg || !final_T || !n_contrib || !dL_dpixels || !means2D || !conic_opacity ||
        !rgb || !dL_record)
        return GSR_EINVAL;
    return gsr_launch_composite_backward(P, width, height, ranges, point_list, means2D, conic_opacity, rgb,
                                         compute_locally, bg, final_T, n_contrib, dL_dpixels, dL_record, out_color,
                                         seg_ws, seg_bytes, row_lo, row_hi, record_is_zero ? 1 : 0,
                                         reinterpret_cast<hipStream_t>(stream));
}

int gsr_render_backward(int P, int width, int height, const int32_t *ranges, const uint32_t *point_list,
                        const float *means2D, const float *conic_opacity, const float *rgb,
                        const uint8_t *compute_locally, const float *bg, const float *final_T,
                        const int32_t *n_contrib, const float *dL_dpixels, float *dL_record, gsr_stream_t stream) {
    if (P < 0 || width <= 0 || height <= 0) return GSR_EINVAL;
    if (P == 0) return 0;
    if (!ranges || !compute_locally || !bg || !final_T || !n_contrib || !dL_dpixels || !means2D || !conic_opacity ||
        !rgb || !dL_record)
        return GSR_EINVAL;
    return gsr_launch_composite_backward(P, width, height, ranges, point_list, means2D, conic_opacity, rgb,
                                         compute_locally, bg, final_T, n_contrib, dL_dpixels, dL_record, nullptr,
                                         nullptr, 0, 0, 0, 0, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
