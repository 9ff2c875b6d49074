/*
 * gsraster.h -- C-ABI of the MI355X (gfx950) Gaussian-splatting rasterizer: the drop-in boundary.
 *
 * Every entry point below is what the reference's Python wrapper module
 * `diff_gaussian_rasterization` (un-vendored submodule, .gitmodules:4-6 of the reference) binds
 * through its `_C` extension for the hot path; the reference-side call sites are cited per function
 * (paths relative to the reference root).  The host-side mirror that re-creates the reference's
 * operator surface on top of this ABI is grendel-gs_amd/diff_gaussian_rasterization/__init__.py;
 * the binding a maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - all data pointers are DEVICE pointers (HBM) unless the name ends in `_host`;
 *  - fp32 / int32 / uint8, dense row-major ("contiguous") layouts, shapes in the comments;
 *  - `stream` is a hipStream_t passed as void* (0 = the null stream); every call only ENQUEUES
 *    work on that stream unless stated otherwise;
 *  - return value: 0 on success, otherwise a hipError_t value (>0) or a GSR_E* code (<0);
 *    gsr_error_string() explains either.  No call aborts the process.
 *  - matrices are in the reference's row-vector convention (scene/cameras.py:84-100):
 *    p_view = [p,1] @ viewmatrix, p_clip = [p,1] @ projmatrix, both 4x4 row-major.
 *  - tiles are 16x16 pixels (utils/general_utils.py:78-93); tile id = ty * tiles_x + tx.
 */
#ifndef GSRASTER_H
#define GSRASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_BLOCK_X 16
#define GSR_BLOCK_Y 16
#define GSR_ONE_DIM_BLOCK 256

#define GSR_EINVAL (-1)   /* bad argument (negative size, null pointer, sh_degree > 3, ...) */
#define GSR_ENOSPACE (-2) /* caller-provided workspace too small */
#define GSR_ERETRY (-3)   /* gsr_bin_count_wait: the count is valid, but a gsr_bin_sort_bounded launched for it wrote nothing */
#define GSR_EFAULT (-4)   /* a persistent binning kernel of an EARLIER call gave up at a barrier behind its first one (a hung or
                             heavily preempted device): that call's lists are incomplete (in bounds); reported once, the
                             device keeps to the look-back pipeline afterwards (gsr_bin_persist_status has the code) */

typedef void *gsr_stream_t;

const char *gsr_error_string(int code);

/* ABI version of this library (bumped on any signature change). */
int gsr_abi_version(void);

/* `_C.get_block_XY()` -- arguments/__init__.py:254-257.  Always (16, 16, 256). */
int gsr_get_block_xy(int *block_x, int *block_y, int *one_dim_block);

/* ---------------------------------------------------------------------------------------------
 * K1  preprocess forward -- GaussianRasterizer.preprocess_gaussians,
 *     gaussian_renderer/__init__.py:949-956 (outputs consumed at :960).
 * in : means3D [P,3], scales [P,3] (activated), rotations [P,4] (r,x,y,z, normalised),
 *      shs [P,sh_coeffs,3], opacities [P] (activated), viewmatrix/projmatrix [4,4], campos [3]
 * out: means2D [P,2] (pixels), depths [P], radii int32 [P] (0 = culled), cov3D [P,6],
 *      conic_opacity [P,4] = (A,B,C,opacity), rgb [P,3], clamped uint8 [P,3]
 * Culled Gaussians get radii 0 and zeros everywhere else. */
int gsr_preprocess_forward(int P, int sh_degree, int sh_coeffs, const float *means3D, const float *scales,
                           float scale_modifier, const float *rotations, const float *shs, const float *opacities,
                           const float *viewmatrix, const float *projmatrix, const float *campos, int width,
                           int height, float tanfovx, float tanfovy, float *means2D, float *depths, int32_t *radii,
                           float *cov3D, float *conic_opacity, float *rgb, uint8_t *clamped, gsr_stream_t stream);

/* K11 preprocess backward -- autograd backward of the call above (train_internal.py:194-196).
 * in : the forward inputs, the saved radii / cov3D / clamped, and the incoming gradients
 *      dL_dmeans2D [P,2] (NDC-scaled units, scene/gaussian_model.py:1046-1064),
 *      dL_dconic_opacity [P,4] (true partials wrt A,B,C,opacity), dL_drgb [P,3].
 *      grad_row_stride == 0: the three are dense; > 0: each pointer addresses a column block of rows that are
 *      grad_row_stride floats apart (e.g. columns 0, 5 and 2 of gsr_render_backward's [P,9] record, stride 9),
 *      so K10's output feeds K11 without a repacking pass
 * out: dL_dmeans3D [P,3], dL_dscales [P,3], dL_drotations [P,4], dL_dshs [P,sh_coeffs,3],
 *      dL_dopacities [P]   (fully overwritten, zeros for culled Gaussians) */
int gsr_preprocess_backward(int P, int sh_degree, int sh_coeffs, const float *means3D, const float *scales,
                            float scale_modifier, const float *rotations, const float *shs, const float *viewmatrix,
                            const float *projmatrix, const float *campos, int width, int height, float tanfovx,
                            float tanfovy, const int32_t *radii, const float *cov3D, const uint8_t *clamped,
                            const float *dL_dmeans2D, const float *dL_dconic_opacity, const float *dL_drgb,
                            int grad_row_stride, float *dL_dmeans3D, float *dL_dscales, float *dL_drotations,
                            float *dL_dshs, float *dL_dopacities, gsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2  `_C.get_local2j_ids_bool` -- gaussian_renderer/workload_division.py:721-744.
 * dist_global_strategy int32 [world_size+1]: band j owns flattened tile ids [d[j], d[j+1]).
 * out uint8 (bool) [P, world_size]: Gaussian i must be sent to band j. */
int gsr_get_local2j_ids_bool(int P, int width, int height, int world_size, const float *means2D,
                             const int32_t *radii, const int32_t *dist_global_strategy, uint8_t *out,
                             gsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3-K7  binning of one camera's (received) Gaussians into the locally computed tiles, in two
 * calls because the number of (tile, Gaussian) pairs D ("num_rendered") sizes the second one.
 * Part of GaussianRasterizer.render_gaussians, gaussian_renderer/__init__.py:1271-1282.
 *
 * gsr_bin_prepare : per Gaussian, count the locally computed tiles it can contribute to (its 3-sigma
 *   rect intersected with the box around its alpha >= 1/255 ellipse: tiles outside that box get
 *   nothing under the alpha < 1/255 rule, so dropping them changes no pixel), order the
 *   Gaussians by depth (stable) and prefix-sum the counts in that order.  Writes D to
 *   *num_rendered_host once the device has produced it (the one host wait of the render op: the host polls a
 *   pinned word the last workgroup writes, and falls back to synchronising `stream` after 2 ms).
 *   `prep` is an opaque device workspace of gsr_bin_prepare_bytes(P, width, height) bytes that must stay
 *   untouched until gsr_bin_sort returns.
 * gsr_bin_sort    : emit the D pairs in depth order and stable-sort them by tile id, giving
 *   point_list uint32 [D] (Gaussian index per pair, grouped by tile, front-to-back inside a
 *   tile, ties by Gaussian index) and ranges int32 [tiles + 1, 2]: rows 0 .. tiles-1 = [start,end) per tile, row `tiles`
 *   = [lo, hi) the tile ROWS that contain locally computed tiles (the band this rank renders; (0,0) when there is
 *   none) -- gsr_render_forward / _backward read it to spread the band, not the whole grid, over the eight XCDs.
 *   The caller allocates tiles + 1 rows (ABI 7).
 *   `scratch` is a device workspace of gsr_bin_sort_bytes(P, D, width, height) bytes. */
size_t gsr_bin_prepare_bytes(int P, int width, int height);
int gsr_bin_prepare(int P, int width, int height, const float *means2D, const float *depths, const int32_t *radii,
                    const float *conic_opacity, const uint8_t *compute_locally, void *prep, size_t prep_bytes,
                    int64_t *num_rendered_host, gsr_stream_t stream);
size_t gsr_bin_sort_bytes(int P, int64_t num_rendered, int width, int height);
int gsr_bin_sort(int P, int width, int height, const uint8_t *compute_locally, const void *prep,
                 int64_t num_rendered, void *scratch, size_t scratch_bytes, uint32_t *point_list, int32_t *ranges,
                 gsr_stream_t stream);
/* The same two steps WITHOUT the idle GPU between them.  The reference's rasterizer (and gsr_bin_prepare above) reads
 * num_rendered back to size its sort buffers, and the GPU has nothing to do until the host has seen the count and
 * launched the sort (~25-35 us per view).  Callers that keep a grow-only scratch can instead
 *   gsr_bin_prepare_async  : launch K3-K4; `*ticket` identifies the count (0 = P was 0, the count is 0);
 *   gsr_bin_sort_bounded   : launch the sort for up to `capacity` pairs -- the kernels read the pair count D from the
 *                            prep workspace ON THE DEVICE and do nothing past it; if D > capacity they write nothing
 *                            useful, and the caller, who learns D from gsr_bin_count_wait, must run gsr_bin_sort with
 *                            buffers of the right size.  point_list must hold `capacity` words.  Only frames of
 *                            <= 256 x 256 tiles (GSR_EINVAL otherwise: use gsr_bin_sort);
 *   gsr_bin_count_wait     : the pair count of `ticket` (polls a pinned word; up to 64 counts may be outstanding per
 *                            device);
 *   gsr_bin_sort_capacity  : the largest num_rendered whose gsr_bin_sort_bytes fits `scratch_bytes` (0 when bounded
 *                            launches do not apply to this frame size). */
/* Process-wide switch for the order of Gaussians with EXACTLY equal depth inside a tile list.  0 (default): by their
 * index in the arrays the op is given -- the reference's order; at world size > 1 that index is (source rank, index on
 * the source), gaussian_renderer/__init__.py:624-640, so the tie order depends on how the Gaussians are sharded.
 * 1: by screen position (means2D.x, then .y, then index): independent of the number of ranks and of the order of the
 * Gaussians; one extra P-sized kernel in gsr_bin_prepare*.  Takes effect for subsequent gsr_bin_prepare* calls. */
int gsr_set_depth_tie_order(int mode);
int gsr_bin_prepare_async(int P, int width, int height, const float *means2D, const float *depths, const int32_t *radii,
                          const float *conic_opacity, const uint8_t *compute_locally, void *prep, size_t prep_bytes,
                          uint32_t *ticket, gsr_stream_t stream);
int gsr_bin_count_wait(uint32_t ticket, int64_t *num_rendered_host, gsr_stream_t stream);
int64_t gsr_bin_sort_capacity(int P, size_t scratch_bytes, int width, int height);
/* The three calls above as ONE (ABI 11; one host-side call per view instead of three): prepare, then -- `capacity` > 0 --
 * the bounded sort into (scratch, point_list [capacity]), then the count.  *status = 0: lists and ranges are complete;
 * 1: run gsr_bin_sort with buffers for *num_rendered_host pairs (no capacity given, the count outgrew it, or the
 * persistent prepare kernel had to repeat itself: see gsr_set_bin_persistent). */
int gsr_bin_speculative(int P, int width, int height, const float *means2D, const float *depths, const int32_t *radii,
                        const float *conic_opacity, const uint8_t *compute_locally, void *prep, size_t prep_bytes,
                        int64_t capacity, void *scratch, size_t scratch_bytes, uint32_t *point_list, int32_t *ranges,
                        int64_t *num_rendered_host, int *status, gsr_stream_t stream);
/* gsr_bin_speculative without its wait (ABI 12): prepare, then -- `capacity` > 0 -- the bounded sort; *ticket names the pair
 * count for gsr_bin_count_wait, *sorted = 1 when the bounded sort was launched.  The kernels that consume the lists (K8,
 * K10) read the range table, never the count, so the caller launches them FIRST and looks at the count afterwards: had it
 * outgrown the capacity (or gsr_bin_count_wait returned GSR_ERETRY) the bounded sort wrote nothing and left every range
 * empty -- the consumer drew the background -- and the caller repeats gsr_bin_sort + the consumer with exact sizes.
 * Stands where the reference's rasterizer reads `num_rendered` back between its sort-key emission and its sort
 * (analyze_statistic.py:1972-1991 stage list: "24 updateDistributedStatLocally.updateTileTouched" -> "50 SortPairs"). */
int gsr_bin_speculative_async(int P, int width, int height, const float *means2D, const float *depths,
                              const int32_t *radii, const float *conic_opacity, const uint8_t *compute_locally, void *prep,
                              size_t prep_bytes, int64_t capacity, void *scratch, size_t scratch_bytes,
                              uint32_t *point_list, int32_t *ranges, uint32_t *ticket, int *sorted, gsr_stream_t stream);
/* K3-K7 run as TWO persistent launches whose workgroups meet at grid-wide barriers (ABI 11; csrc/binning_persist.h)
 * instead of the nine launches of the look-back pipeline, when the device can hold the grid at once, the frame has
 * <= 256 x 256 tiles and P <= 8 x 4096 x CUs.  Lists, ranges and offsets are bit-identical.  A barrier kernel needs
 * its whole grid resident: the library never has two of them in flight on different streams of one process (the second
 * call takes the look-back pipeline); a device shared with ANOTHER PROCESS's barrier kernel is covered by a time-out at
 * the first barrier -- the prepare step then repeats itself on the look-back pipeline inside gsr_bin_count_wait, which
 * returns GSR_ERETRY (the count is valid; a gsr_bin_sort_bounded already launched for it wrote nothing: call
 * gsr_bin_sort); in the sort step the time-out (250 ms) is decided for the whole grid, every workgroup but one leaves and
 * that one sorts the view alone inside the same launch -- correct lists, no host round trip, works in a captured graph --
 * after which the next 64 sorts of the device take the look-back pipeline (ABI 12; ABI 11 trapped here).  No kernel of
 * this library traps: a time-out at a LATER barrier (whole grid resident: a hung or preempted device) leaves a code in the
 * status words and the next binning call on the device returns GSR_EFAULT.  mode: -1 the environment's GSR_BIN_PERSIST (0 | 1 | p | s,
 * default 1), 0 off, 1 prepare only, 2 sort only, 3 both. */
int gsr_set_bin_persistent(int mode);
/* K5-K7 with ONE pass over the D (tile, Gaussian) pairs (ABI 12; csrc/binning_rows.h): the h ROW SEGMENTS of every rect
 * (R = sum of heights ~ D / 3..6) are sorted stably by tile row, the pairs are emitted from that order and scattered
 * stably by column inside their row, straight into point_list; the ranges follow from per-tile counts.  Replaces the two
 * D-sized sort passes + K7 of the split-key pipelines (their ranking is bound by VALU issue at every size: 0.28-0.33 of
 * the HBM peak) where it applies: frames of <= 256 x 256 tiles, uncut rects (gsr_set_tile_cull off).  Lists, ranges and
 * counts are bit-identical.  mode: -1 the environment's GSR_BIN_ROWS (0 | 1, default 1), 0 off, 1 on.
 * Reference stages replaced: 40 duplicateWithKeys, 50 SortPairs, 60 identifyTileRanges (analyze_statistic.py:1972-1991). */
int gsr_set_bin_rowmajor(int mode);
/* Exact tile culling in K3 (ABI 11; csrc/binning_persist.h: gsr_tile_mask): per tile row of a Gaussian's rect the exact span
 * of tiles its alpha >= 1/255 ellipse reaches (the quadratic form and tolerance of the composite kernels' own skip test),
 * kept as a 64-bit mask (rects of <= 64 tiles on frames of <= 256 x 256 tiles).  The lists stay order-preserving
 * subsequences of the uncut ones -- a dropped (tile, Gaussian) has alpha < 1/255 on every pixel of the tile -- so the image
 * and the gradients are unchanged up to the summation order of the blend; D shrinks by 9-19 %.  OFF by default: measured
 * neutral (K3 pays per Gaussian what the D-sized passes and the composite kernels save per pair).  mode: -1 the
 * environment's GSR_TILE_CULL (0 | 1 | auto, default 0), 0 off, 1 on, 2 auto = on for frames of more than
 * GSR_TILE_CULL_TILES (default 16384) tiles -- a static rule: the lists do not depend on earlier views. */
int gsr_set_tile_cull(int mode);
/* Diagnostics of the persistent launches on the current device: out4 = { sequence number of the last launch that passed
 * its last barrier, code of the last barrier fault (0x100 + n: barrier n of the prepare kernel, 0x200 + n: of the sort
 * kernel; 0 = none), number of faults (each is reported once as GSR_EFAULT by the next binning call), number of views
 * the sort kernel finished with one workgroup after its first barrier timed out }.  ABI 12: four words (ABI 11: three). */
int gsr_bin_persist_status(uint32_t *out4);
/* Diagnostics (GSR_BIN_TIMELINE=1 in the environment): the per-workgroup phase stamps (100 MHz clock, 32 per workgroup)
 * of the last persistent prepare (which = 0) / sort (which = 1) launch; *grid = its workgroups.  Synchronises. */
int gsr_bin_timeline(int which, unsigned long long *out, int max_words, int *grid);
/* Byte offset, inside the prep workspace of gsr_bin_prepare*, of the uint32 pair count D that K4 leaves on the DEVICE
 * (what gsr_bin_sort_bounded's kernels read).  For callers that capture the iteration in a hipGraph and therefore
 * cannot take D through gsr_bin_count_wait inside a replay: they copy the word out / test it with
 * gsr_flag_if_greater.  Stands where the reference reads `num_rendered` back (its rasterizer's forward). */
size_t gsr_bin_total_offset(int P, int width, int height);
/* The same for the uint32 count of ROW SEGMENTS R = sum of the rects' heights (ABI 12; the R-sized steps of
 * csrc/binning_rows.h work on it; 0 on frames above 256 x 256 tiles).  A measurement aid: bench.py reads it in its
 * instrumented replay to state the bytes the row-major pipeline moves (68 P + 28 R + 4 D). */
size_t gsr_bin_segments_offset(int P, int width, int height);

/* Capacity checks of a captured (hipGraph) training iteration -- the reference sizes everything from counts it reads
 * back (num_rendered; the exchange's i2j sizes, gaussian_renderer/__init__.py:572-585), a replayed graph cannot.
 * Buffers are sized from earlier iterations instead and these launches raise bits of ONE device word when a capacity
 * does not hold; gsr_preprocess_backward_adam_raw_batched_dyn skips its update when the word is non-zero and the host,
 * which reads the word after the replay, repeats the iteration eagerly.
 *   gsr_flag_if_greater : *flag_dev |= bit  when  *value_dev > limit  (e.g. the pair count against the sort capacity)
 *   gsr_exchange_check  : all_counts_dev / caps_dev int32 [W][W][B] (rows rank i sends rank j of camera k, and the
 *                         slab reserved for them): bit_over when a count exceeds its slab; bit_few when a band this
 *                         rank (`me`) renders -- bit k of rendered_mask -- receives fewer than `few` rows in total
 *                         (the reference's < 10-Gaussian stand-in rule, gaussian_renderer/__init__.py:1260-1269).
 *   Both also leave what they looked at in PINNED, device-accessible host memory when host_copy_pinned is not NULL
 *   (the value / the W*W*B counts, system-scope stores): the host reads them after the replay, no copy node needed.
 *   gsr_publish_flag    : last launch of a captured iteration: { *flag_dev, *seq_dev } are stored (system scope, the
 *                         stamp last) into words 2 s, 2 s + 1 of a pinned, device-accessible ring of `slots` pairs,
 *                         s = *seq_dev % slots; seq_dev is a device word the host refreshes in front of every replay.
 *                         The host polls the stamp instead of synchronising the stream. */
/* A device timestamp inside a captured iteration (ABI 13): word `index` (< per_slot) of slot *seq_dev % slots of a pinned,
 * device-accessible ring of uint64 receives the 100 MHz s_memrealtime counter when the launch runs (seq_dev NULL: slot 0).
 * Two stamps around a group of launches are what the HIP event pair of the eager loop measures (events recorded inside a
 * capture cannot be read): the load balancer's render / loss times (workload_division.py:953-966 of the reference) under
 * graph replay.  Valid for the host once gsr_publish_flag's stamp of the same replay has landed. */
int gsr_stamp(const uint32_t *seq_dev, uint64_t *host_ring_pinned, uint32_t slots, uint32_t per_slot, uint32_t index,
              gsr_stream_t stream);
/* compute_locally of B row bands from DEVICE data (ABI 13): mask uint8 [B][grid_y][grid_x], mask[k][ty][.] = lo_k <= ty <
 * hi_k with { lo_k, hi_k } = the first two words of record k of band_rows_dev (records of stride_words int32).  What
 * DivisionStrategyFinal.get_compute_locally (workload_division.py:773-787 of the reference) builds on the host per
 * partition, as a launch a hipGraph can replay for any partition. */
int gsr_band_mask(int grid_x, int grid_y, int B, const int32_t *band_rows_dev, int stride_words, uint8_t *mask,
                  gsr_stream_t stream);
int gsr_publish_flag(const uint32_t *flag_dev, const uint32_t *seq_dev, uint32_t *host_ring_pinned, uint32_t slots,
                     gsr_stream_t stream);
int gsr_flag_if_greater(const uint32_t *value_dev, uint32_t limit, uint32_t *flag_dev, uint32_t bit,
                        uint32_t *host_copy_pinned, gsr_stream_t stream);
int gsr_exchange_check(const int32_t *all_counts_dev, const int32_t *caps_dev, int W, int B, int me,
                       uint64_t rendered_mask, int few, uint32_t *flag_dev, uint32_t bit_over, uint32_t bit_few,
                       int32_t *host_copy_pinned, gsr_stream_t stream);
int gsr_bin_sort_bounded(int P, int width, int height, const uint8_t *compute_locally, const void *prep,
                         int64_t capacity, void *scratch, size_t scratch_bytes, uint32_t *point_list, int32_t *ranges,
                         gsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K8  composite forward -- the rest of render_gaussians (gaussian_renderer/__init__.py:1271-1282).
 * out_color [3,H,W]: sum c*alpha*T + T_final*bg on locally computed tiles, exactly 0 elsewhere
 * (images are assembled by SUM all-reduce, train_internal.py:466-469); final_T [H,W];
 * n_contrib int32 [H,W] (1-based position of the last blended entry of the tile's list). */
int gsr_render_forward(int P, int width, int height, const int32_t *ranges, const uint32_t *point_list,
                       const float *means2D, const float *conic_opacity, const float *rgb,
                       const uint8_t *compute_locally, const float *bg, float *out_color, float *final_T,
                       int32_t *n_contrib, gsr_stream_t stream);

/* K10 composite backward -- autograd backward of render_gaussians.
 * out (fully overwritten): dL_record [P,9], one row per Gaussian =
 *   [0:2] dL_dmeans2D (pixel gradient x (W/2, H/2)), [2:5] dL_drgb, [5:9] dL_dconic_opacity
 * -- the column order of the exchange's differentiable record (means2D, rgb, conic_opacity:
 * gaussian_renderer/__init__.py:647-650), so that at world size > 1 the record IS the message of the mirror
 * all-to-all.  One record instead of three arrays lets the kernel flush the 9 sums of a (tile, Gaussian) pair
 * from 9 adjacent lanes into one 36-byte row: ~8x fewer memory-side atomic requests. */
int gsr_render_backward(int P, int width, int height, const int32_t *ranges, const uint32_t *point_list,
                        const float *means2D, const float *conic_opacity, const float *rgb,
                        const uint8_t *compute_locally, const float *bg, const float *final_T,
                        const int32_t *n_contrib, const float *dL_dpixels, float *dL_record, gsr_stream_t stream);
/* K8 / K10 with list SEGMENTS (round 4).  One workgroup per tile walks the tile's list serially, so a launch lasts at
 * least as long as its longest list takes one wave -- which is the whole kernel on a thin row band (one round of
 * resident workgroups at world size 8).  The backward can be cut exactly: per pixel it needs the transmittance and the
 * colour accumulated IN FRONT of a list position, which the forward knows when it passes that position.  With a
 * workspace (gsr_render_seg_bytes(width, height) bytes, caller-allocated, alive until the backward has run)
 *   gsr_render_forward_seg  leaves a checkpoint (T, C.rgb per pixel) every 256 list entries it really walks and queues
 *                           the segment that starts there;
 *   gsr_render_backward_seg runs segment 0 of every tile in its usual workgroups and hands the queued segments to
 *                           persistent worker workgroups.  out_color = the forward's image (the colour BEHIND a
 *                           boundary is the final colour minus the checkpointed one).
 * Early termination is untouched (segments exist only for entries the forward walked); gradients equal the one-segment
 * kernel's up to fp32 rounding (the checkpointed T replaces a chain of divisions).  seg_ws == NULL: exactly
 * gsr_render_forward / gsr_render_backward.  Same call sites as those (gaussian_renderer/__init__.py:1271-1282).
 * [row_lo, row_hi): the TILE ROWS of the caller's band when it knows them on the host (Grendel's strategies do:
 * compute_locally must then be false outside these rows), else 0, 0.  The launches then cover the band's tiles only; a
 * grid over all tiles costs a constant ~30 us (forward) / ~60 us (backward) of workgroup dispatch for tiles that are not
 * ours, whatever the band.  The forward still leaves every pixel outside the band exactly 0.
 * row_lo == -1 (ABI 13): the band is DEVICE data -- row_hi (1 .. grid rows) is only the CAPACITY of the launch in tile
 * rows, the band itself is the row hull of compute_locally that the tile sort left behind the range table (row `tiles`
 * of `ranges`, written by gsr_bin_sort / _bounded / _speculative* on the device; a band of more rows than row_hi is the
 * caller's error: its last rows are not drawn).  One launch captured in a hipGraph then serves every band of at most
 * row_hi rows (graphed_step.py: one graph for all cameras of a live partition, workload_division.py:806-849 of the
 * reference moves the cut points per camera); workgroups above the band's tile count only help clearing the pixels
 * outside. */
size_t gsr_render_seg_bytes(int width, int height);
int gsr_render_forward_seg(int P, int width, int height, const int32_t *ranges, const uint32_t *point_list,
                           const float *means2D, const float *conic_opacity, const float *rgb,
                           const uint8_t *compute_locally, const float *bg, float *out_color, float *final_T,
                           int32_t *n_contrib, void *seg_ws, size_t seg_bytes, int row_lo, int row_hi,
                           gsr_stream_t stream);
int gsr_render_backward_seg(int P, int width, int height, const int32_t *ranges, const uint32_t *point_list,
                            const float *means2D, const float *conic_opacity, const float *rgb,
                            const uint8_t *compute_locally, const float *bg, const float *final_T,
                            const int32_t *n_contrib, const float *dL_dpixels, float *dL_record,
                            const float *out_color, void *seg_ws, size_t seg_bytes, int row_lo, int row_hi,
                            gsr_stream_t stream);
/* The same pair with the backward's fill moved into the forward (ABI 12).  K10 adds into the [P,9] record with atomics,
 * so the record starts at zero: gsr_render_backward[_seg] clears it with a fill launch at the head of every backward
 * (36 MB per 10^6 Gaussians).  gsr_render_forward_seg_z additionally clears [zero_ptr, zero_ptr + zero_bytes) -- 16-byte
 * aligned, a multiple of 4 bytes; the caller passes the record it will hand to the backward -- from the composite
 * kernel's own workgroups (the kernel is bound by VALU issue, its memory pipes are idle), and
 * gsr_render_backward_seg_z(record_is_zero = 1) skips the fill.  Nothing else may write the record in between. */
int gsr_render_forward_seg_z(int P, int width, int height, const int32_t *ranges, const uint32_t *point_list,
                             const float *means2D, const float *conic_opacity, const float *rgb,
                             const uint8_t *compute_locally, const float *bg, float *out_color, float *final_T,
                             int32_t *n_contrib, void *seg_ws, size_t seg_bytes, int row_lo, int row_hi, void *zero_ptr,
                             size_t zero_bytes, gsr_stream_t stream);
int gsr_render_backward_seg_z(int P, int width, int height, const int32_t *ranges, const uint32_t *point_list,
                              const float *means2D, const float *conic_opacity, const float *rgb,
                              const uint8_t *compute_locally, const float *bg, const float *final_T,
                              const int32_t *n_contrib, const float *dL_dpixels, float *dL_record, const float *out_color,
                              void *seg_ws, size_t seg_bytes, int row_lo, int row_hi, int record_is_zero,
                              gsr_stream_t stream);
/* Measurement aid (bench.py's roofline leg; the reference has nothing to bind here): list entries the composite kernels
 * WALKED since the last reset, summed over launches -- out2[0] K8, out2[1] K10; per tile the entries its longest-walking
 * quadrant goes through (K8: up to the chunk in which the last pixel saturates; K10: the largest n_contrib of the
 * tile).  Early termination leaves most of every list untouched, so D-based byte formulas over-credit these kernels.
 * Synchronous (a device-to-host copy of two words); reset != 0 zeroes the counters afterwards. */
int gsr_composite_walked(unsigned long long *out2, int reset);

/* ---------------------------------------------------------------------------------------------
 * N1  fused band-local L1 + SSIM loss -- the arithmetic of final_system_loss_computation,
 * gaussian_renderer/loss_distribution.py:2536-2585 (pixelwise_l1_with_mask / pixelwise_ssim_with_mask,
 * utils/loss_utils.py:88-132): 11x11 Gaussian window (sigma 1.5), zero padding at the band edges,
 * C1 = 0.01^2, C2 = 0.03^2, ground truth = uint8 / 255.
 * `image` points at the band's first row of channel 0 inside a [C, H, W] image (row stride = width,
 * channel stride = image_channel_stride elements); gt is the dense uint8 band [C, rows, width].
 * forward : partials [gsr_l1_ssim_num_partials()][2] = per-workgroup (sum |x-y|, sum ssim_map); the
 *           caller adds them up.  dm_* [C, rows, width] receive the derivative maps the backward
 *           needs (all three NULL for a forward-only evaluation).
 * backward: grad_image (same addressing as `image`, channel stride grad_channel_stride) =
 *           scale_l1 * *grad_l1_sum * sign(x - y) + scale_ssim * *grad_ssim_sum * d(sum ssim)/dx ; the two device
 *           scalars are read on the device (no host sync) -- a caller whose loss is c_l1 * S_l1 + c_ssim * S_ssim passes
 *           the incoming dL/dloss for both pointers and (c_l1, c_ssim) as the scales: no glue kernel. */
int gsr_l1_ssim_num_partials(int channels, int rows, int width);
int gsr_l1_ssim_forward(int channels, int rows, int width, const float *image, int64_t image_channel_stride,
                        const uint8_t *gt, float *partials, float *dm_dmu1, float *dm_dE11, float *dm_dE12,
                        gsr_stream_t stream);
int gsr_l1_ssim_backward(int channels, int rows, int width, const float *image, int64_t image_channel_stride,
                         const uint8_t *gt, const float *dm_dmu1, const float *dm_dE11, const float *dm_dE12,
                         const float *grad_l1_sum, const float *grad_ssim_sum, float scale_l1, float scale_ssim,
                         float *grad_image, int64_t grad_channel_stride, gsr_stream_t stream);
/* The same pair for a band whose rows are DEVICE data (ABI 13; one captured launch for every band of a camera):
 * band_rows = { y0, y1 } int32 pixel rows of `image` (NOT offset: the full image's base pointer) on the device;
 * rows_capacity >= y1 - y0 sizes the launch, the partials (gsr_l1_ssim_num_partials(C, rows_capacity, width): the slots
 * above the band's own count receive zeros) and the CHANNEL STRIDE of gt and dm_* ([C, rows_capacity, width], the band
 * in the first y1 - y0 rows of every channel).  Tile ids and the order of the partial sums are those of a launch sized
 * for the band: the finalized loss is bit-equal to gsr_l1_ssim_forward's on the same rows. */
int gsr_l1_ssim_forward_band(int channels, int rows_capacity, int width, const float *image, int64_t image_channel_stride,
                             const uint8_t *gt, float *partials, float *dm_dmu1, float *dm_dE11, float *dm_dE12,
                             const int32_t *band_rows, gsr_stream_t stream);
int gsr_l1_ssim_backward_band(int channels, int rows_capacity, int width, const float *image,
                              int64_t image_channel_stride, const uint8_t *gt, const float *dm_dmu1, const float *dm_dE11,
                              const float *dm_dE12, const float *grad_l1_sum, const float *grad_ssim_sum, float scale_l1,
                              float scale_ssim, float *grad_image, int64_t grad_channel_stride, const int32_t *band_rows,
                              gsr_stream_t stream);
/* finalize: adds the partials up (fixed order, fp64 accumulation) and forms the band's loss terms in one
 * launch: out3[0] = c_l1 * S_l1 + c_ssim * S_ssim + bias  (batched_loss_computation's
 * (1 - lambda) * Ll1 + lambda * (1 - ssim) with c_l1 = (1-lambda)/n, c_ssim = -lambda/n, bias = lambda;
 * gaussian_renderer/loss_distribution.py:2627-2629), out3[1] = S_l1 * inv_n, out3[2] = S_ssim * inv_n. */
int gsr_l1_ssim_finalize(int num_partials, const float *partials, float c_l1, float c_ssim, float bias, float inv_n,
                         float *out3, gsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * N3  fused Adam step for one parameter tensor of n fp32 elements (16-byte aligned, dense):
 * the update stock torch.optim.Adam performs as configured at scene/gaussian_model.py:292
 * (no weight decay, no amsgrad), with the reference's `grad /= bsz` (train_internal.py:319-324)
 * folded in as grad_scale.  `step` is the 1-based step count AFTER this update (bias correction).
 * Hyper-parameters are doubles so that 1 - beta and the bias corrections round like the stock optimizer's. */
int gsr_adam_step(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, double lr,
                  double beta1, double beta2, double eps, int64_t step, float grad_scale, gsr_stream_t stream);

/* The same update for up to 16 parameter tensors in ONE launch (the reference's optimizer walks its six param
 * groups, scene/gaussian_model.py:244-292: xyz, f_dc, f_rest, opacity, scaling, rotation): arrays of length
 * num_tensors with each tensor's element count, pointers and its group's hyper-parameters / step count.
 * Tensors with 0 elements are skipped. */
int gsr_adam_step_multi(int num_tensors, const int64_t *numels, float *const *params, const float *const *grads,
                        float *const *exp_avgs, float *const *exp_avg_sqs, const double *lrs, const double *beta1s,
                        const double *beta2s, const double *epss, const int64_t *steps, float grad_scale,
                        gsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2 for a whole camera batch, in the layout of the exchange (rows a5/a6).  The reference calls
 * get_local2j_ids_bool once per camera and then nonzero() per (camera, band) (workload_division.py:721-744,
 * gaussian_renderer/__init__.py:586-622).  With the camera-batched K1 the state is [B,P,.]:
 * means2D fp32 [B,P,2], radii int32 [B,P]; bands int32 [B][W][2] = tile rows [lo,hi) of camera k rendered by
 * GLOBAL rank g ((0,0): none); need uint8 [W][B][P] = 1 iff the Gaussian's 3-sigma tile rect (the K2 rule)
 * meets that band -- flattened, this is the (destination, camera, Gaussian) send order of the all-to-all-v;
 * counts int32 [W][B] = row sums of need (zeroed here).  W <= 256. */
int gsr_exchange_need(int P, int B, int W, int width, int height, const float *means2D, const int32_t *radii,
                      const int32_t *bands, uint8_t *need, int32_t *counts, gsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The exchange without the mask (rows a5/a6; what the mirror's _batched_exchange_final calls).  The reference's
 * all_to_all_communication_final (gaussian_renderer/__init__.py:542-698) runs K2, then nonzero() per (camera, band),
 * index_select + cat per destination; its autograd backward is an index_put with accumulate.  Here, for the cameras
 * [k0, k0 + B) of a batch of B_total whose state is camera-major ([B_total,P,.] arrays):
 *   gsr_exchange_count : counts int32 [W][B] (every entry written here) = number of Gaussians rank g needs of camera
 *       k0 + kk, and chunkcnt int32 [W * B][gsr_exchange_chunks(P)] = the same per 1024-Gaussian chunk (a workspace for
 *       pack; counts is its row sums, formed by a second launch -- no atomics);
 *   gsr_exchange_pack  : after the caller has turned the counts of all ranks into the layout of its send buffer --
 *       segment_offsets[g * B + kk] (HOST array, W * B <= 512 entries) = first row of the (destination g, camera
 *       k0 + kk) segment -- writes the 11-float records (means2D 2, rgb 3, conic_opacity 4, radius bits, depth) into
 *       msg fp32 [n_send][11] in (destination, camera, local index) order and send_idx int32 [n_send] = kk * P + i,
 *       the row each record came from; chunkcnt is what gsr_exchange_count wrote for the camera range
 *       [count_first, count_first + count_cameras) (which must contain [k0, k0 + B): one count launch for the batch,
 *       one pack launch per camera when the exchanges are pipelined);
 *   gsr_scatter_add_rows : dst[idx[r]][0:9] += src[r][0:9] -- the backward's mirror step on the gradient rows the
 *       peers send back (dst fp32 [rows][9], zeroed by the caller; a Gaussian needed by two bands is added twice);
 *       rows with idx[r] < 0 are skipped. */
size_t gsr_exchange_chunks(int P);
int gsr_exchange_count(int P, int B_total, int k0, int B, int W, int width, int height, const float *means2D,
                       const int32_t *radii, const int32_t *bands, int32_t *chunkcnt, int32_t *counts,
                       gsr_stream_t stream);
int gsr_exchange_pack(int P, int B_total, int k0, int B, int W, int width, int height, int count_cameras,
                      int count_first, const float *means2D,
                      const float *rgb, const float *conic_opacity, const int32_t *radii, const float *depths,
                      const int32_t *bands, const int32_t *chunkcnt, const int32_t *segment_offsets, int64_t n_send,
                      float *msg, int32_t *send_idx, gsr_stream_t stream);
int gsr_scatter_add_rows(int64_t n, const int32_t *idx, const float *src, float *dst, gsr_stream_t stream);
/* The exchange WITHOUT the host read-back (round 3).  gsr_exchange_pack needs the segment layout, i.e. the counts of
 * this very step on the host: the reference's `.cpu()` of gaussian_renderer/__init__.py:572-585, one device round trip
 * per iteration in the middle of the forward.  gsr_exchange_pack_slab instead packs into CAPACITY slabs the caller
 * chose beforehand (from earlier iterations): capacities[g * B + kk] (HOST, W * B <= 512 entries, sum == n_rows) rows
 * are reserved for (destination g, camera k0 + kk), back to back; the records go to the front of their slab in the
 * same (local index) order, records past the capacity are dropped, and the unused tail of every slab is written as
 * all-zero records (radius 0: the receiving rank's binning culls them) with send_idx -1 (gsr_scatter_add_rows skips
 * negative rows).  `counts` is the DEVICE array gsr_exchange_count wrote (same launch as chunkcnt).  The caller sends
 * whole slabs (all-to-all-v with the capacities as split sizes), learns the true counts later (an asynchronous copy
 * that has long completed when it polls the pair count of the first render) and repeats the exchange with
 * gsr_exchange_pack when any count exceeded its capacity -- the same speculate / verify scheme as
 * gsr_bin_sort_bounded.  Received rows keep the reference's order (source rank, then the source's index); padding
 * rows sit between the sources' blocks. */
/* gsr_exchange_unpack: the received message recv fp32 [n][11] (any mix of records and padding rows) -> the five dense
 * tensors of the render op: means2D [n,2], rgb [n,3], conic_opacity [n,4], radii int32 [n] (the record's radius BITS),
 * depths [n].  gsr_zero_async: hipMemsetAsync(ptr, 0, bytes) on the caller's stream (the gradient record the mirror
 * exchange's scatter-add accumulates into). */
int gsr_exchange_unpack(int64_t n, const float *recv, float *means2D, float *rgb, float *conic_opacity, int32_t *radii,
                        float *depths, gsr_stream_t stream);
int gsr_zero_async(void *ptr, size_t bytes, gsr_stream_t stream);
int gsr_exchange_pack_slab(int P, int B_total, int k0, int B, int W, int width, int height, int count_cameras,
                           int count_first, const float *means2D, const float *rgb, const float *conic_opacity,
                           const int32_t *radii, const float *depths, const int32_t *bands, const int32_t *chunkcnt,
                           const int32_t *counts, const int32_t *capacities, int64_t n_rows, float *msg,
                           int32_t *send_idx, gsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * N4  `simple_knn._C.distCUDA2(points)` (scene/gaussian_model.py:20,163-166; submodule
 * https://gitlab.inria.fr/bkerbl/simple-knn, .gitmodules:1-3, absent from the reference tree):
 * mean_dist2[i] = mean of the squared distances from point i to its 3 nearest OTHER points (only i itself is
 * excluded, by index; coincident points count with distance 0).  Exact (Morton order + bounded boxes only
 * prune the search).  With fewer than 4 points the mean runs over the neighbours that exist (0 for one point).
 * points: fp32 [P,3]; workspace: gsr_knn_workspace_bytes(P) bytes of device memory, any content. */
size_t gsr_knn_workspace_bytes(int P);
int gsr_knn_mean_dist2(int P, const float *points, float *mean_dist2, void *workspace, size_t workspace_bytes,
                       gsr_stream_t stream);

/* N4: the per-iteration densification statistics in ONE launch (ABI 12) -- densification.py:13-25 of the reference, run after
 * every backward of the densification phase: for the rows with radii > 0
 *   max_radii2D = max(max_radii2D, radii);  xyz_gradient_accum += |grad[:, :2]|;  denom += 1
 * (scene/gaussian_model.py:1046-1052 `add_densification_stats` + the max_radii2D update in front of it), without the
 * reference's boolean indexing (two `nonzero` host syncs + six gather / scatter kernels per camera and iteration).
 * grad: the means2D gradient (NDC-scaled, as the op returns it), rows `grad_stride` floats apart -- 9 for the view of K10's
 * [P,9] record that the operator hands out as means2D.grad, 2 for a dense [P,2]. */
int gsr_densify_stats(int64_t P, const int32_t *radii, const float *grad, int64_t grad_stride, float *max_radii2D,
                      float *accum, float *denom, gsr_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * N4  row primitives for densification / redistribution of the Gaussian shard.  The reference selects rows
 * with boolean indexing once PER TENSOR (and per destination rank): prune_points / _prune_optimizer
 * scene/gaussian_model.py:775-835, densify_and_clone / densify_and_split :922-1007, all2all_gaussian_state
 * :1073-1098.  Here the selection is computed once and applied to all tensors in one launch.
 *
 * gsr_group_rows: dest[i] in [0,G) = group of row i, anything else = dropped.  order[] receives the row
 *   indices grouped by destination, original order kept inside a group (group 0, 1, ..., G-1, then the dropped
 *   rows); counts[0..G-1] = group sizes, counts[G] = number dropped (device int64).  1 <= G <= 255.
 * gsr_gather_rows: for every tensor k, dst_k[r, 0:width_k] = src_k[order ? order[r] : r, 0:width_k] for
 *   r < n_out.  Rows are 4-byte words (fp32 / int32); strides are in words, so a dst may be a column block of a
 *   wider record matrix (one all-to-all-v for all per-Gaussian state) and a src may be such a block.
 *   srcs/dsts/widths/strides are HOST arrays of num_tensors <= 32 entries. */
size_t gsr_group_rows_bytes(int64_t N);
int gsr_group_rows(int64_t N, int G, const int32_t *dest, int32_t *order, int64_t *counts, void *workspace,
                   size_t workspace_bytes, gsr_stream_t stream);
int gsr_gather_rows(int64_t n_out, const int32_t *order, int num_tensors, const void *const *srcs, void *const *dsts,
                    const int32_t *widths, const int64_t *src_strides, const int64_t *dst_strides,
                    gsr_stream_t stream);
/* gsr_scatter_rows: the inverse, dst_k[order[r], 0:width_k] = src_k[r, 0:width_k] for r < n_in (rows not named in
 * order[0:n_in] keep their contents).  N2 (scene/gaussian_model.py:1350-1391, the sparse gradient sync of the
 * replicated-storage mode) packs the touched rows of the six gradient tensors into ONE compact [nnz,59] buffer with
 * gsr_gather_rows, all-reduces it, and writes the sums back with this call. */
int gsr_scatter_rows(int64_t n_in, const int32_t *order, int num_tensors, const void *const *srcs, void *const *dsts,
                     const int32_t *widths, const int64_t *src_strides, const int64_t *dst_strides,
                     gsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K1 / K11 on the RAW parameters of GaussianModel (scene/gaussian_model.py:219-242): `scaling` log-scales
 * [P,3], `rotation` un-normalised quaternions [P,4], `opacity` logits [P,1], `features_dc` [P,1,3],
 * `features_rest` [P,sh_coeffs-1,3].  The getters' activations (scene/gaussian_model.py:109-129: exp,
 * normalize with eps 1e-12, sigmoid, cat) are applied in registers, so the activated copies are never
 * written to or re-read from HBM; outputs / saved tensors are those of gsr_preprocess_forward, and the
 * backward returns gradients with respect to the raw parameters.  Used by this package's
 * gaussian_renderer mirror; the reference-shaped entry points above remain the drop-in surface. */
int gsr_preprocess_forward_raw(int P, int sh_degree, int sh_coeffs, const float *xyz, const float *scaling,
                               float scale_modifier, const float *rotation, const float *features_dc,
                               const float *features_rest, const float *opacity, const float *viewmatrix,
                               const float *projmatrix, const float *campos, int width, int height, float tanfovx,
                               float tanfovy, float *means2D, float *depths, int32_t *radii, float *cov3D,
                               float *conic_opacity, float *rgb, uint8_t *clamped, gsr_stream_t stream);
int gsr_preprocess_backward_raw(int P, int sh_degree, int sh_coeffs, const float *xyz, const float *scaling,
                                float scale_modifier, const float *rotation, const float *features_dc,
                                const float *features_rest, const float *opacity, const float *viewmatrix,
                                const float *projmatrix, const float *campos, int width, int height, float tanfovx,
                                float tanfovy, const int32_t *radii, const float *cov3D, const uint8_t *clamped,
                                const float *dL_dmeans2D, const float *dL_dconic_opacity, const float *dL_drgb,
                                int grad_row_stride, float *dL_dxyz, float *dL_dscaling, float *dL_drotation, float *dL_dfeatures_dc,
                                float *dL_dfeatures_rest, float *dL_dopacity, gsr_stream_t stream);

/* The same for a BATCH of B cameras (--bsz B: every rank projects its Gaussians for all cameras of the batch,
 * gaussian_renderer/__init__.py:919-963) in one launch each way: a Gaussian's raw parameters are read once,
 * the backward sums the cameras' gradients in registers.  cams [B][40] floats on the device, per camera
 * { viewmatrix[16], projmatrix[16], campos[3], tanfovx, tanfovy, pad[3] }; outputs / incoming gradients are
 * camera-major [B,P,...]; cov3D [P,6] is camera independent. */
int gsr_preprocess_forward_raw_batched(int P, int B, int sh_degree, int sh_coeffs, const float *xyz,
                                       const float *scaling, float scale_modifier, const float *rotation,
                                       const float *features_dc, const float *features_rest, const float *opacity,
                                       const float *cams, int width, int height, float *means2D, float *depths,
                                       int32_t *radii, float *cov3D, float *conic_opacity, float *rgb,
                                       uint8_t *clamped, gsr_stream_t stream);
int gsr_preprocess_backward_raw_batched(int P, int B, int sh_degree, int sh_coeffs, const float *xyz,
                                        const float *scaling, float scale_modifier, const float *rotation,
                                        const float *features_dc, const float *features_rest, const float *opacity,
                                        const float *cams, int width, int height, const int32_t *radii,
                                        const float *cov3D, const uint8_t *clamped, const float *dL_dmeans2D,
                                        const float *dL_dconic_opacity, const float *dL_drgb, int grad_row_stride, float *dL_dxyz,
                                        float *dL_dscaling, float *dL_drotation, float *dL_dfeatures_dc,
                                        float *dL_dfeatures_rest, float *dL_dopacity, gsr_stream_t stream);

/* K11 of a batch FUSED with the optimizer step of the six tensors it differentiates (N3 inside a9's neighbour: the
 * reference runs `loss.backward()` and then `gaussians.optimizer.step()` over the same 59 floats per Gaussian,
 * train_internal.py:195 and :316-328, scene/gaussian_model.py:292).  A Gaussian's gradient is complete when its lane
 * leaves the camera loop, so the dense Adam update (gsr_adam_step_multi's arithmetic, bit for bit) is applied there and
 * the gradients never reach HBM: xyz .. opacity are read AND updated in place; exp_avgs / exp_avg_sqs are host arrays of
 * the six device pointers of the moments in the order xyz, scaling, rotation, features_dc, features_rest, opacity; lrs ..
 * steps host arrays of the six groups' hyper-parameters (steps = the 1-based count AFTER this update); grad_scale as
 * in gsr_adam_step.  Needs sh_coeffs == 16 (GSR_EINVAL otherwise: run the two unfused calls).  tanfov0: HOST pointer
 * to { tanfovx, tanfovy } when B == 1 (selects the one-camera kernel like gsr_preprocess_backward_raw), or NULL. */
int gsr_preprocess_backward_adam_raw_batched(int P, int B, int sh_degree, int sh_coeffs, float *xyz, float *scaling,
                                             float scale_modifier, float *rotation, float *features_dc,
                                             float *features_rest, float *opacity, const float *cams, int width,
                                             int height, const int32_t *radii, const float *cov3D,
                                             const uint8_t *clamped, const float *dL_dmeans2D,
                                             const float *dL_dconic_opacity, const float *dL_drgb,
                                             int grad_row_stride, float *const *exp_avgs, float *const *exp_avg_sqs,
                                             const double *lrs, const double *beta1s, const double *beta2s,
                                             const double *epss, const int64_t *steps, float grad_scale,
                                             const float *tanfov0, gsr_stream_t stream);
/* The same launch for an iteration captured in a hipGraph (train_internal.py:134-208,316-329 replayed as one graph):
 * what changes from step to step is read from DEVICE memory at execution time -- dyn_dev: 12 floats, lr / (1 - beta1^t)
 * of the six tensors then 1 / sqrt(1 - beta2^t) of the six tensors (lrs / steps may then be NULL) -- and skip_flag_dev
 * (DEVICE uint32, may be NULL) makes the launch a no-op when non-zero (a capacity check of the same replay failed, see
 * gsr_flag_if_greater).  With both NULL this is gsr_preprocess_backward_adam_raw_batched. */
int gsr_preprocess_backward_adam_raw_batched_dyn(int P, int B, int sh_degree, int sh_coeffs, float *xyz, float *scaling,
                                                 float scale_modifier, float *rotation, float *features_dc,
                                                 float *features_rest, float *opacity, const float *cams, int width,
                                                 int height, const int32_t *radii, const float *cov3D,
                                                 const uint8_t *clamped, const float *dL_dmeans2D,
                                                 const float *dL_dconic_opacity, const float *dL_drgb,
                                                 int grad_row_stride, float *const *exp_avgs,
                                                 float *const *exp_avg_sqs, const double *lrs, const double *beta1s,
                                                 const double *beta2s, const double *epss, const int64_t *steps,
                                                 float grad_scale, const float *tanfov0, const float *dyn_dev,
                                                 const uint32_t *skip_flag_dev, gsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a19  fused parameter activations -- GaussianModel.get_scaling / get_rotation / get_opacity /
 * get_features (scene/gaussian_model.py:109-129): scales = exp(_scaling) [N,3], rotations =
 * normalize(_rotation) [N,4] (eps 1e-12), opacities = sigmoid(_opacity) [N,1], shs = cat(_features_dc
 * [N,1,3], _features_rest [N,sh_rest,3]) [N,1+sh_rest,3]; and the matching backward. */
int gsr_activate_forward(int N, int sh_rest, const float *scaling, const float *rotation, const float *opacity,
                         const float *features_dc, const float *features_rest, float *scales, float *rotations,
                         float *opacities, float *shs, gsr_stream_t stream);
int gsr_activate_backward(int N, int sh_rest, const float *rotation, const float *scales, const float *opacities,
                          const float *g_scales, const float *g_rotations, const float *g_opacities,
                          const float *g_shs, float *d_scaling, float *d_rotation, float *d_opacity,
                          float *d_features_dc, float *d_features_rest, gsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSRASTER_H */
