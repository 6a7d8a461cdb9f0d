/*
 * mspa.h -- C ABI of libmspa.so, the MI355X (gfx950) geometry engine behind the MultiSPA
 * data pipeline of facebookresearch/Multi-SpatialMLLM.
 *
 * The reference is pure Python and has no FFI of its own: its hot path sits behind plain Python
 * callables.  Each entry point below names the reference callable(s) it
 * replaces (paths relative to the reference tree; IH = spatial_engine/utils/scannet_utils/
 * handler/info_handler.py, OPS = .../handler/ops.py, CFR = spatial_engine/camera_movement/
 * calculate_frames_relations.py, MVI = spatial_engine/utils/scannet_utils/make_visibility_info.py,
 * CME = spatial_engine/camera_movement/camera_movement_engine_train_val.py, OM_C =
 * spatial_engine/object_movement/single_object_movement_engine_coord.py).  The ctypes binding a
 * maintainer would add on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer (HBM) unless the name ends in _host.  The library never
 *    allocates, frees or copies user buffers: the caller (PyTorch-ROCm tensors in the shipped host
 *    layer) owns all memory and passes data_ptr() values plus element counts.
 *  - `stream` is a hipStream_t (NULL = the default stream).  Calls enqueue work and return; they
 *    never synchronise.  Entry points are re-entrant; there is no global mutable state besides the
 *    thread-local error string.
 *  - Matrices are 4x4, row-major, float64, affine (last row exactly 0 0 0 1; the host layer
 *    verifies this and the finiteness of every entry, as the reference's pose-validity filter
 *    IH:409-418 does, before calling).  Inverses are computed by the caller with the same LAPACK
 *    routine the reference uses (numpy.linalg.inv) and passed in ready-made.
 *  - All arithmetic is IEEE float64 in the reference's operation order (each matrix row is the
 *    chain m0*x, fma(m1,y,.), fma(m2,z,.), +m3 -- what NumPy/OpenBLAS computes for K = 4), so
 *    integer outputs and masks are bit-exact and float64 outputs are bit-identical to the C oracle.
 *  - Return value: MSPA_OK or a negative MSPA_E* code; mspa_last_error_string() describes the
 *    last failure on the calling thread.  Nothing throws or aborts.
 */
#ifndef MSPA_H
#define MSPA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSPA_VERSION 160            /* 0.6.0: depth-frame decode on the device (mspa_inflate_blocks_device, mspa_png_unfilter_device); 0.5.0: host-side depth-PNG ingest (mspa_read_depth_png_host); 0.4.0: frame records carry
                                       guard-bound coefficients (slot MSPA_MAT_BOUNDS, MSPA_FRAME_MATS 7 -> 8) */

#define MSPA_OK 0
#define MSPA_EINVAL (-1)            /* bad argument (null pointer, size out of range, ...) */
#define MSPA_EHIP (-2)              /* a HIP runtime call failed; see mspa_last_error_string() */
#define MSPA_EUNSUPPORTED (-3)      /* valid request this build has no kernel for */

typedef void *mspa_stream_t;        /* hipStream_t */

/* Matrix slots of one frame record in `frame_mats` ([n_frames][MSPA_FRAME_MATS][16] float64). */
#define MSPA_MAT_KINV 0             /* inv(K)            OPS:313 */
#define MSPA_MAT_E 1                /* E, camera->world  OPS:316 */
#define MSPA_MAT_A 2                /* A, world->aligned OPS:319-320 (identity when absent) */
#define MSPA_MAT_EINV_ALIGNED 3     /* inv(A @ E)        IH:113-124, IH:57 */
#define MSPA_MAT_K 4                /* K                 IH:66 */
#define MSPA_MAT_UNPROJ 5           /* A @ E @ inv(K)    composed on the host (fast path only) */
#define MSPA_MAT_REPROJ 6           /* K @ inv(A @ E)    composed on the host (fast path only) */
#define MSPA_MAT_BOUNDS 7           /* not a matrix: the frame's guard-bound coefficients (fast path only), filled by
                                       mspa_frame_bounds_host from slots 0-4.  With |.| entrywise, Uabs = |A| |E| |inv(K)|,
                                       Nabs = |K| |inv(A @ E)| and c = MSPA_GUARD_C * 2^-53:
                                         [0..2]  max over rows r < 3 of Uabs[r][j], j = 0, 1, 2      [3]  1000 * max_r Uabs[r][3]
                                         [4..6]  c * (Nabs[k][0] + Nabs[k][1] + Nabs[k][2]), k = 0, 1, 2
                                         [8..10] c * 1000 * Nabs[k][3]                              (other entries 0)
                                       For a pixel block with columns <= X, rows <= Y and depth samples <= D millimetres,
                                       w = D * ([0] X + [1] Y + [2]) + [3] bounds the aligned-world coordinates (mm) and
                                       B_k = [4 + k] * w + [8 + k] bounds the difference between ANY two float64 evaluation
                                       orders of the k-th homogeneous image coordinate (pixel * millimetre; k = 2: the
                                       camera-2 depth in mm) -- the reference's five products vs the composed 3x4 of the
                                       fast kernels.  The guard band of MSPA_PAIR_FAST is derived from B_k per tile. */
#define MSPA_FRAME_MATS 8
#define MSPA_GUARD_C 256.0          /* >= the roundings of either evaluation order (89 counted) */

/* Slots of one image record in `cam_mats` ([n_images][MSPA_CAM_MATS][16] float64) of K1 / K6b. */
#define MSPA_CAM_EINV 0             /* inv(A @ E)        IH:57 */
#define MSPA_CAM_K 1                /* K                 IH:66 */
#define MSPA_CAM_BOUNDS 2           /* not a matrix: guard-bound coefficients of the composed K1 kernels, filled by
                                       mspa_camera_bounds_host from slots 0-1.  Nabs = |K| |inv(A @ E)|, c = MSPA_GUARD_C * 2^-53 * 1000:
                                         [0] c (nr_0 + nr_1)   [1] c nr_2   [2] c (Nabs[0][3] + Nabs[1][3])   [3] c Nabs[2][3]
                                       with nr_k = Nabs[k][0] + Nabs[k][1] + Nabs[k][2] (other entries 0).  For a vertex with
                                       s = |x| + |y| + |z| (metres), [0] s + [2] bounds the difference between any two float64
                                       evaluation orders of the homogeneous image x and y (summed; pixel * millimetre) and
                                       [1] s + [3] that of the camera depth (millimetres). */
#define MSPA_CAM_MATS 3

/* flags of mspa_pair_reproject */
#define MSPA_PAIR_FAST 1u           /* composed-matrix evaluation with exact re-evaluation of every
                                       lane near a decision boundary: masks, pixel indices and counters
                                       stay bit-exact, float32 points agree to ~1e-12 relative.  Needs
                                       slots 5-7 filled and K's third row == 0 0 1 0 (pinhole); ignored
                                       (the exact kernel runs) when any float64 output is requested.  A bitset
                                       output on a width that is not a multiple of 64 takes a linear pixel mapping
                                       (64 consecutive pixel indices per wave) so that ballots stay whole words. */
#define MSPA_PAIR_STREAM 2u         /* caller's hint: the frames of this launch are (mostly) not revisited by other pairs of
                                       the same launch -- frame 1 is read with the non-temporal hint so that it does not
                                       displace frame-2 lines that ARE revisited (gathers).  Results are identical with or
                                       without it; measured +3 % with distinct frames, -10 % when 48 frames serve 1000 pairs */

#define MSPA_PAIR_WORD_STRIPES 0x200u /* diagnostic / A-B: at ScanNet's shape (1296x968 over 640x480), correspondence / minimal sets, take
                                       round 2-3's word-aligned wobbling-stripe kernel (MSPA_KERNEL_PAIR_FAST_SCALED) instead of
                                       the rectangular-tile kernel that is the default since round 4.  Results are identical. */

int mspa_version(void);
const char *mspa_last_error_string(void);

/*
 * Host-side: fills slot MSPA_MAT_BOUNDS of n_frames frame records from their slots 0-4 (HOST pointer to the
 * [n_frames, MSPA_FRAME_MATS, 16] table, before it is uploaded).  The shipped host layer computes the same numbers in NumPy
 * (mspa.engine.frame_bounds); a C caller that builds its own table calls this once per table.
 */
int mspa_frame_bounds_host(double *frame_mats_host, int32_t n_frames);
/* The same for K1's image records: fills slot MSPA_CAM_BOUNDS of [n_images, MSPA_CAM_MATS, 16] from slots 0-1 (HOST pointer). */
int mspa_camera_bounds_host(double *cam_mats_host, int32_t n_images);

/* Device facts the host layer reports next to measurements.  Any out pointer may be NULL. */
int mspa_device_info(int device, int *n_cu, int *wave_size, int64_t *hbm_bytes, int *clock_khz,
                     char *name_host, int name_len);

/*
 * K3 -- the frame-pair pipe (BASELINE.json north_star).  For every pair (f1, f2) and every colour
 * pixel (mx, my) of frame f1:
 *   project_mask_to_3d          OPS:235-329   depth -> camera -> world -> aligned point
 *   project_3d_point_to_image   IH:313-335, project_points IH:46-72   into frame f2
 *   check_point_visibility      IH:337-386    bounds + strict depth-buffer test in frame f2
 *
 *   depth       [n_frames, dh, dw] uint16 millimetres
 *   rgb         [n_frames, H, W, 3] uint8 or NULL (needed only for out_rgba)
 *   frame_mats  [n_frames, MSPA_FRAME_MATS, 16] float64
 *   pairs       [n_pairs, 2] int32 frame indices (f1, f2)
 * Outputs, each optional (NULL = not produced), P = H*W pixels in row-major order:
 *   out_vis_bits  [n_pairs, ceil(P/64)] uint64   bit (i & 63) of word (i >> 6) = pixel i visible
 *   out_vis_u8    [n_pairs, P] uint8             1 = visible in f2, 0 otherwise
 *   out_valid_u8  [n_pairs, P] uint8             1 = depth sample of f1 > 0 (OPS:297)
 *   out_pix_i16   [n_pairs, P, 2] int16          depth-pixel index (xi, yi) in f2 (IH:362-366) where the
 *                                                point lands inside f2 in front of its camera (the
 *                                                candidate correspondence); (-1,-1) otherwise
 *   out_xyz_f32   [n_pairs, P, 3] float32        aligned world point, NaN where invalid
 *   out_rgba      [n_pairs, P] uint32            r | g<<8 | b<<16 | (valid ? 255 : 0)<<24
 *   out_xyz_f64   [n_pairs, P, 3], out_uv_f64 [n_pairs, P, 2], out_depth_f64 [n_pairs, P]
 *                                                full-precision point, un-rounded (u, v) in f2
 *                                                and camera-2 depth; NaN where invalid
 *   out_counts    [n_pairs, 2] int32             (#valid, #visible); zeroed by the call
 * Requires 2 <= dw,dh,W,H <= 32767 and H*W*W < 2^32.
 */
int mspa_pair_reproject(const uint16_t *depth, const uint8_t *rgb, const double *frame_mats,
                        int32_t n_frames, const int32_t *pairs, int64_t n_pairs,
                        int32_t dh, int32_t dw, int32_t H, int32_t W,
                        uint64_t *out_vis_bits, uint8_t *out_vis_u8, uint8_t *out_valid_u8,
                        int16_t *out_pix_i16, float *out_xyz_f32, uint32_t *out_rgba,
                        double *out_xyz_f64, double *out_uv_f64, double *out_depth_f64,
                        int32_t *out_counts, uint32_t flags, mspa_stream_t stream);

/* Which kernel the most recent mspa_pair_reproject call OF THE CALLING THREAD enqueued (a diagnostic the parity tests
 * use to prove that the instantiation they mean to check is the one that ran; no effect on results). */
#define MSPA_KERNEL_NONE 0
#define MSPA_KERNEL_PAIR_EXACT 1          /* reference operation order */
#define MSPA_KERNEL_PAIR_FAST 2           /* composed + guarded, any shape, stripe mapping */
#define MSPA_KERNEL_PAIR_FAST_LINEAR 3    /* the same with the linear pixel mapping (bitset, W % 64 != 0) */
#define MSPA_KERNEL_PAIR_FAST_TIGHT 4     /* whole-tile images (W % 64 == 0, H % 48 == 0, colour == depth grid), output set
                                             corr / dense / dense without colour / minimal / compact */
#define MSPA_KERNEL_PAIR_FAST_SCALED 5    /* ScanNet's 1296x968 over 640x480 on word-aligned wobbling stripes (MSPA_PAIR_WORD_STRIPES only) */
#define MSPA_KERNEL_PAIR_FAST_RECT 6      /* the tight kernel on rectangular tiles of a colour / depth grid pair with dw <= W, dh <= H,
                                             W % 16 == 0, H % 4 == 0, dw % 4 == 0, dh % 2 == 0, H * W * 4 < 2^31 and
                                             dh * dw * 2 < 2^31 that is not a whole-tile shape (ScanNet's shape included):
                                             correspondence / minimal / compacted sets */
int mspa_pair_reproject_last_kernel(void);

/*
 * K3, compacted correspondence output -- "overlap + correspondence extraction" of the frame-pair pipe without the dense
 * [P, 2] table, three quarters of which is (-1, -1) fill on overlap-sampled pairs: per pair the visibility bitset and, per
 * 64 x 48-pixel tile of frame f1, the (xi, yi) of its VISIBLE pixels only.
 *
 *   Tiles: n_stripes = ceil(W / 64) across, ceil(H / 48) down, tile t = band * n_stripes + stripe (ragged at the right /
 *   bottom edge); mspa_corr_tiles(H, W) returns their number.
 *   out_vis_bits     [n_pairs, ceil(P/64)] uint64   as in mspa_pair_reproject
 *   out_cpix_i16     [n_pairs, n_tiles, MSPA_CORR_TILE_CAP, 2] int16   tile t's segment holds, for its visible pixels in
 *                    (row, column) order, the depth-pixel index (xi, yi) in f2 (IH:362-366) -- the k-th set bit of the
 *                    tile's part of the bitset <-> entry k; entries beyond the tile's count are unspecified (the fused
 *                    kernel writes 4 bytes per VISIBLE pixel and nothing else; only a tile in which the exact re-evaluation
 *                    of a guarded pixel took a pixel OUT of the visible set may keep stale entries behind its count)
 *   out_tile_counts  [n_pairs, n_tiles] int32       visible pixels per tile (= entries of its segment)
 *   out_counts       [n_pairs, 2] int32 or NULL     (#valid, #visible); zeroed by the call
 * With MSPA_PAIR_FAST on a whole-tile shape (W % 64 == 0, H % 48 == 0, colour grid == depth grid: BASELINE's 640x480) and on
 * every shape with dw <= W, dh <= H, W % 16 == 0, H % 4 == 0, dw % 4 == 0, dh % 2 == 0, H * W * 4 < 2^31 and dh * dw * 2 < 2^31
 * (the predicate of MSPA_KERNEL_PAIR_FAST_RECT: ScanNet's own 1296x968 colour over 640x480 depth; ragged tiles hold fewer
 * pixels, same indexing) one fused kernel produces all of it -- mspa_pair_correspondences_workspace_bytes() is the authority
 * on which shapes those are (it returns 0 for them); every other shape / mode runs mspa_pair_reproject into a dense table in
 * `workspace` (caller-owned, 16-byte aligned, at least mspa_pair_correspondences_workspace_bytes(...) bytes; NULL / 0 when
 * that returns 0) and compacts it with mspa_compact_correspondences.  Identical results either way (bit-exact integers).
 * out_cpix_i16 must be 16-byte aligned.  MSPA_PAIR_STREAM as in mspa_pair_reproject.  The fused kernel additionally needs
 * `depth` 4-byte aligned (any allocator's base address is): a table at an odd 2-byte offset takes the dense route as well,
 * i.e. needs the workspace although mspa_pair_correspondences_workspace_bytes, which cannot see the pointer, returned 0
 * (the error string says so).  Depth samples are millimetres here and in mspa_pair_reproject (project_mask_to_3d's literal
 * 0.001, OPS:292-294): a handler's depth_value_scale applies to K1 / K6b / the predicates only.
 */
#define MSPA_CORR_TILE_W 64
#define MSPA_CORR_TILE_H 48
#define MSPA_CORR_TILE_CAP (MSPA_CORR_TILE_W * MSPA_CORR_TILE_H)
int64_t mspa_corr_tiles(int32_t H, int32_t W);
int64_t mspa_pair_correspondences_workspace_bytes(int64_t n_pairs, int32_t dh, int32_t dw, int32_t H, int32_t W,
                                                  uint32_t flags);
int mspa_pair_correspondences(const uint16_t *depth, const double *frame_mats, int32_t n_frames,
                              const int32_t *pairs, int64_t n_pairs, int32_t dh, int32_t dw, int32_t H, int32_t W,
                              uint64_t *out_vis_bits, int16_t *out_cpix_i16, int32_t *out_tile_counts,
                              int32_t *out_counts, void *workspace, int64_t workspace_bytes, uint32_t flags,
                              mspa_stream_t stream);
/* The compaction step on its own: (vis_bits, dense pix_i16 [n_pairs, P, 2]) of mspa_pair_reproject -> segments + counts. */
int mspa_compact_correspondences(const uint64_t *vis_bits, const int16_t *pix_i16, int64_t n_pairs, int32_t H, int32_t W,
                                 int16_t *out_cpix_i16, int32_t *out_tile_counts, mspa_stream_t stream);

/*
 * K1 -- vertex visibility, HOT LOOP 1 of CFR.process_scene (CFR:152-164) and
 * MVI.process_scene (MVI:93-100): project_3d_point_to_image (IH:313-335) + check_point_visibility
 * (IH:375-386) of all scene vertices into a batch of images.
 *
 *   xyz         vertex coordinates, element (i, c) at xyz[i*point_stride + c*comp_stride]
 *               ([N,3] rows: 3,1;  aligned_points.npy [N,6] rows: 6,1;  SoA [3,N]: 1,N)
 *   cam_mats    [n_images, MSPA_CAM_MATS, 16] float64: inv(A @ E), K, guard-bound coefficients (mspa_camera_bounds_host), per image
 *   depth       [n_images, dh, dw] uint16
 * Outputs, each optional:
 *   out_bits    [n_images, ceil(n_points/64)] uint64  visibility bitset (tail bits zero)
 *   out_mask    [n_images, n_points] uint8
 *   out_uv      [n_images, n_points, 2] float64        un-rounded projection (IH:72)
 *   out_depth   [n_images, n_points] float64           signed camera depth (IH:63)
 *   out_count   [n_images] int32                       visible vertices per image; zeroed by the call
 */
int mspa_vertex_visibility(const double *xyz, int64_t n_points, int64_t point_stride,
                           int64_t comp_stride, const double *cam_mats, int32_t n_images,
                           const uint16_t *depth, int32_t dh, int32_t dw, int32_t H, int32_t W,
                           uint64_t *out_bits, uint8_t *out_mask, double *out_uv, double *out_depth,
                           int32_t *out_count, mspa_stream_t stream);

/*
 * The same with the two parameters the reference's interface leaves open:
 *   homogeneous != 0   the points are general homogeneous rows (x, y, z, w), the fourth coordinate at
 *                      xyz[i*point_stride + 3*comp_stride] -- project_points (IH:46-72) takes any [N, 4] array; both 4x4
 *                      products are then evaluated in full (w == 1 gives the bits of the affine form)
 *   depth_value_scale  metres per depth-image unit (SceneInfoHandler(depth_value_scale=...), IH:76, applied at IH:368);
 *                      0.001 in mspa_vertex_visibility.  Positive and finite.
 * Either one selects the reference-order kernel (the composed kernels fold w == 1 and the millimetre in).
 */
int mspa_vertex_visibility_ex(const double *xyz, int64_t n_points, int64_t point_stride, int64_t comp_stride,
                              int32_t homogeneous, const double *cam_mats, int32_t n_images, const uint16_t *depth,
                              int32_t dh, int32_t dw, int32_t H, int32_t W, double depth_value_scale,
                              uint64_t *out_bits, uint8_t *out_mask, double *out_uv, double *out_depth,
                              int32_t *out_count, mspa_stream_t stream);

/*
 * The three predicates of the reference on already-projected points, for callers that hold (uv, depth)
 * from an earlier projection: check_point_in_image_boundary (IH:337-344),
 * check_point_visibility_by_depth (IH:346-373), check_point_visibility (IH:375-386).
 *   uv [n, 2] f64, point_depth [n] f64, depth_image [dh, dw] u16 (may be NULL for the bounds test alone)
 *   out_in_bounds / out_by_depth / out_visible [n] u8, each optional
 */
int mspa_check_visibility(const double *uv, const double *point_depth, int64_t n,
                          const uint16_t *depth_image, int32_t dh, int32_t dw, int32_t H, int32_t W,
                          uint8_t *out_in_bounds, uint8_t *out_by_depth, uint8_t *out_visible,
                          mspa_stream_t stream);
/* ... with the handler's depth_value_scale (IH:368) instead of the default 0.001 */
int mspa_check_visibility_ex(const double *uv, const double *point_depth, int64_t n,
                             const uint16_t *depth_image, int32_t dh, int32_t dw, int32_t H, int32_t W,
                             double depth_value_scale, uint8_t *out_in_bounds, uint8_t *out_by_depth,
                             uint8_t *out_visible, mspa_stream_t stream);

/*
 * K2 -- pair overlap, HOT LOOP 2 of CFR.process_scene: calculate_camera_overlap (CFR:102-137)
 * on K1's bitsets.  overlap = |a & b| / |a | b| * 100 in float64 (NaN for an empty union).
 *
 *   bits        [n_images, n_words] uint64
 *   pairs       [n_pairs, 2] int32 image indices
 *   out_overlap [n_pairs] float64;  out_inter / out_union [n_pairs] int32, optional
 */
int mspa_pair_overlap(const uint64_t *bits, int32_t n_images, int64_t n_words, const int32_t *pairs,
                      int64_t n_pairs, double *out_overlap, int32_t *out_inter, int32_t *out_union,
                      mspa_stream_t stream);

/*
 * K2, tiled forms.  The reference visits every pair of a scene (CFR:176-189: F(F-1)/2 calls of
 * calculate_camera_overlap) and, for object visibility, every (object, image) combination
 * (compute_object_visibility.py:72-152).  These two entry points do a whole scene / rectangle in one pass: 32 x 32
 * blocks of rows, bitset chunks staged in LDS, only |a & b| counted (|a | b| = |a| + |b| - |a & b|).  Results are
 * identical to mspa_pair_overlap's.  `workspace` is caller-owned scratch of at least
 * mspa_overlap_workspace_bytes(n_a, n_b, n_words) bytes (4-byte aligned); its contents are undefined afterwards.
 *
 * mspa_scene_overlap: all pairs i < j of one scene in the reference's nested-loop order (CFR:176-178),
 *   out_overlap [F(F-1)/2] float64 (NaN for an empty union, CFR:136), out_inter / out_union [F(F-1)/2] int32 optional.
 * mspa_overlap_matrix: out_inter [n_a, n_b] int32 = |a_i & b_j| (bits_a == bits_b with n_a == n_b is recognised as
 *   symmetric: half the blocks are computed, the matrix is returned full; its diagonal holds the row popcounts).
 */
int64_t mspa_overlap_workspace_bytes(int32_t n_a, int32_t n_b, int64_t n_words);
int mspa_scene_overlap(const uint64_t *bits, int32_t n_images, int64_t n_words, void *workspace,
                       int64_t workspace_bytes, double *out_overlap, int32_t *out_inter, int32_t *out_union,
                       mspa_stream_t stream);
int mspa_overlap_matrix(const uint64_t *bits_a, int32_t n_a, const uint64_t *bits_b, int32_t n_b, int64_t n_words,
                        void *workspace, int64_t workspace_bytes, int32_t *out_inter, mspa_stream_t stream);

/*
 * K9 -- visibility bitsets -> index lists on the device: what MVI.process_scene (make_visibility_info.py:103-118) builds as
 * Python lists -- np.where(mask)[0].tolist() per image, and per vertex the sorted ids of the images that see it -- as CSR
 * tables.  The second table is the compaction of the TRANSPOSED bit matrix.
 *   mspa_bits_popcount   out_counts[i] = popcount(bits[i]) over a flat table of n_words_total words
 *   mspa_bits_expand     bits [n_rows, n_words]; word_offsets [n_rows * n_words] int64 = exclusive prefix sum of the
 *                        popcounts (row r's list starts at word_offsets[r * n_words]); out_indices[k] int32 = position of
 *                        the k-th set bit WITHIN ITS ROW, rows back to back, ascending within a row
 *   mspa_bits_transpose  bits [n_rows, n_words] -> out [n_words * 64, ceil(n_rows / 64)]: bit (r & 63) of
 *                        out[c, r >> 6] = bit (c & 63) of bits[r, c >> 6]; padding bits are zero
 */
int mspa_bits_popcount(const uint64_t *bits, int64_t n_words_total, int32_t *out_counts, mspa_stream_t stream);
int mspa_bits_expand(const uint64_t *bits, int64_t n_rows, int64_t n_words, const int64_t *word_offsets,
                     int32_t *out_indices, mspa_stream_t stream);
int mspa_bits_transpose(const uint64_t *bits, int32_t n_rows, int64_t n_words, uint64_t *out, mspa_stream_t stream);

/*
 * Host-side text formatting of index columns (HOST pointers).  The visibility parquet stores each list as its JSON text
 * (json.dumps form: "[1, 2, 3]", "[\"00000\", \"00005\"]", "[]"; make_visibility_info.py:38-73) under keys
 * "scene:point_to_images:idx"; these write that text straight into arrow's string layout (int32 offsets + UTF-8 data) from
 * the CSR tables K9 produces.  Each returns the number of bytes written (>= 0) or a negative MSPA_E* code;
 * out_text_offsets_host has n + 1 entries.  A buffer of 2 + 13 * len bytes per integer list, 2 + (longest token + 2) * len
 * per token list and strlen(prefix) + 21 per key is always large enough.
 *   mspa_format_int_lists_host    list k = values[offsets[k] .. offsets[k+1])
 *   mspa_format_token_lists_host  list k = tokens named by token_ids[offsets[k] .. offsets[k+1]); token t is the text
 *                                 tokens[token_offsets[t] .. token_offsets[t+1]) (e.g. an already quoted image id)
 *   mspa_format_int_keys_host     key k = prefix followed by the decimal text of first + k
 */
int64_t mspa_format_int_lists_host(const int64_t *offsets_host, const int32_t *values_host, int64_t n_lists,
                                   char *out_text_host, int64_t capacity, int32_t *out_text_offsets_host);
int64_t mspa_format_token_lists_host(const int64_t *offsets_host, const int32_t *token_ids_host, int64_t n_lists,
                                     const char *tokens_host, const int32_t *token_offsets_host, int32_t n_tokens,
                                     char *out_text_host, int64_t capacity, int32_t *out_text_offsets_host);
int64_t mspa_format_int_keys_host(const char *prefix_host, int64_t first, int64_t n, char *out_text_host,
                                  int64_t capacity, int32_t *out_text_offsets_host);

/*
 * The same text written ON the device (round 6, K10: csrc/format_lists.hip) -- every list item formatted by its own lane at its
 * final byte position -- for the visibility-index sweep, whose encoder threads spent half of their time per scene in the host
 * loops above (make_visibility_info.py:38-73: json.dumps per key).  DEVICE pointers; two launches around two prefix sums
 * that are the caller's plumbing (as for mspa_bits_popcount -> prefix sum -> mspa_bits_expand):
 *   mspa_format_list_costs_device  out_cost[e] = bytes of item e's text + 2.  token_offsets_dev NULL: items are integers
 *                                  (decimal, '-' for negatives); else item e is token values[e] of n_tokens tokens
 *                                  (token t = tokens[token_offsets[t] .. token_offsets[t+1])); an id outside sets *bad_flag_dev
 *   mspa_format_lists_device       cost_prefix_dev: exclusive prefix sum of out_cost as int64, nnz + 1 entries;
 *                                  nonempty_prefix_dev[r]: lists q < r with offsets[q+1] > offsets[q], n_lists + 1 int64 entries;
 *                                  text_bytes = 2 n_lists + cost_prefix[nnz] - 2 nonempty_prefix[n_lists] (< 2 GiB);
 *                                  writes the text and arrow's int32 string offsets (n_lists + 1 entries).
 * Bit-exact with the host formatters (and json.dumps).
 */
int mspa_format_list_costs_device(const int32_t *values_dev, int64_t nnz, const int32_t *token_offsets_dev, int32_t n_tokens,
                                  int32_t *out_cost_dev, int32_t *bad_flag_dev, mspa_stream_t stream);
int mspa_format_lists_device(const int64_t *offsets_dev, const int32_t *values_dev, int64_t n_lists, int64_t nnz,
                             const int64_t *cost_prefix_dev, const int64_t *nonempty_prefix_dev, const char *tokens_dev,
                             const int32_t *token_offsets_dev, char *out_text_dev, int64_t text_bytes,
                             int32_t *out_text_offsets_dev, mspa_stream_t stream);

/*
 * Host-side staging of a scene's depth frames (the loop that fills the frame stack in the reference: CFR / MVI read one
 * image per frame into its own array): n_blocks equally sized host blocks are gathered into one contiguous destination
 * -- the pinned buffer the H2D copy reads -- by up to n_threads copy threads.  Host pointers only; no device work.
 */
int mspa_gather_blocks_host(const void *const *src_blocks_host, int64_t n_blocks, int64_t block_bytes, void *dst_host,
                            int32_t n_threads);

/*
 * Host-side ingest of zlib-compressed frames: the depth payloads of a ScanNet .sens stream (`zlib_ushort`,
 * extract_posed_images.py:49-57 inflates them one by one).  Block k (src_bytes_host[k] bytes at src_blocks_host[k], e.g.
 * inside a memory-mapped file) is inflated into dst_host + k * block_bytes by up to n_threads threads.  Every block
 * must inflate to exactly block_bytes; otherwise MSPA_EINVAL and the error string names the block.
 */
int mspa_inflate_blocks_host(const void *const *src_blocks_host, const int64_t *src_bytes_host, int64_t n_blocks,
                             int64_t block_bytes, void *dst_host, int32_t n_threads);

/*
 * Host-side ingest of a scene's depth frames from disk: the per-frame `cv2.imread(depth_png, -1)` of
 * SceneInfoHandler.get_depth_image (info_handler.py:149-155), which CFR.process_scene / MVI.process_scene call once per
 * image (CFR:152-157, MVI:93-100) and which the reference hides behind a process pool over scenes (CFR:222-229,
 * MVI:151-156).  The n_files files named by paths_host (NUL-terminated) are read and decoded into
 * dst_host[k * h * w ...] (host byte order) by up to n_threads native threads; no interpreter lock is involved.
 * Decoded natively: PNG, 16-bit greyscale, non-interlaced, of exactly h x w pixels (what extract_posed_images.py:118-123
 * writes).  status_host[k]: 0 decoded; 1 file unreadable; 2 a PNG of another pixel format or size (decode that frame with
 * a general reader); 3 corrupt.  Returns MSPA_OK when the call ran (the per-file status tells), MSPA_EINVAL on a bad argument.
 *   mspa_png_header_host   (h, w, bit depth, colour type, interlace flag) of one file's IHDR; each output optional
 */
int mspa_read_depth_png_host(const char *const *paths_host, int64_t n_files, int32_t h, int32_t w, uint16_t *dst_host,
                             int32_t n_threads, int32_t *status_host);
int mspa_png_header_host(const char *path_host, int32_t *h, int32_t *w, int32_t *bit_depth, int32_t *color_type,
                         int32_t *interlace);
/*
 * The table-driven zlib decoder the two ingest entry points above try first (csrc/inflate_fast.h; the from-disk sweep is bound
 * by the inflate of its depth frames -- zlib's own delivers ~155 MB/s per thread on them).  One whole zlib stream in host memory
 * -> exactly dst_bytes of output.  Returns 0 when the stream was decoded (output size AND Adler-32 match), 1 when the decoder
 * declines it (damaged or truncated stream, other output size, preset dictionary): the callers above then use zlib itself.
 */
int mspa_inflate_zlib_fast_host(const void *src_host, int64_t src_bytes, void *dst_host, int64_t dst_bytes);

/*
 * Depth-frame decode ON THE DEVICE (csrc/device_ingest.hip): the same two reference stages -- `cv2.imread(depth_png, -1)` per
 * frame (info_handler.py:149-155) and `zlib.decompress` per .sens frame (extract_posed_images.py:49-57) -- with the COMPRESSED
 * bytes crossing PCIe and the frames landing in HBM as the [F, h, w] uint16 block K1 / K3 read.
 *
 * mspa_inflate_blocks_device   n_blocks zlib streams (RFC 1950 / 1951: stored, fixed and dynamic blocks, 32 K window), stream k =
 *   src_bytes_dev[k] bytes at src_dev + src_offsets_dev[k] (offsets multiples of 8; the buffer src_dev .. + src_capacity must
 *   hold every stream and may be read up to the 8-byte unit a stream ends in), each inflated by ONE wave into dst_dev + k *
 *   dst_pitch.  status_dev[k] (int32, written for every k): 0 the stream inflated to EXACTLY block_bytes, ended inside its
 *   input and the Adler-32 of the output equals its trailer; 1 not a valid / supported stream or another size; 2 checksum
 *   mismatch.  A block with a non-zero status holds garbage: decode that frame on the host (mspa_inflate_blocks_host /
 *   mspa_read_depth_png_host) -- the same "accept only what verifies, hand the rest on" contract as csrc/inflate_fast.h.
 *   work_dev: n_blocks uint32 of scratch.  src_dev 16-byte, dst_dev and dst_pitch 256-byte aligned; block_bytes <= 64 MiB.
 * mspa_png_unfilter_device     n_images inflated scanline blocks (h rows of 1 filter byte + 2 w sample bytes, image k at raw_dev +
 *   k * raw_pitch) -> out_dev[k, h, w] uint16 in host byte order: the five PNG row filters (None / Sub / Up / Average / Paeth)
 *   undone, big-endian samples swapped.  Images whose status_dev[k] is non-zero on entry are skipped; an image with a filter
 *   byte > 4 gets status 3.
 * mspa_png_pack_idat_host      the host half: the n_files 16-bit greyscale non-interlaced h x w PNG files are read by up to
 *   n_threads threads and the payloads of their IDAT chunks -- the scanlines' zlib stream -- packed into dst_host (pinned memory
 *   the H2D copy reads) at offsets_host[k] (multiples of 16), bytes_host[k] long.  Call with dst_host == NULL first: only the
 *   files' sizes are looked up and *capacity_needed is set (offsets are assigned from the file sizes, so both calls agree).
 *   status_host[k]: 0 packed; 1 unreadable; 2 another pixel format or size; 3 corrupt chunk structure.
 */
/*
 * A stream for long-running background kernels (the depth decode above) that leaves `reserve_cus` compute units alone, spread
 * over the device (every (n_cu / reserve_cus)-th unit; hipExtStreamCreateWithCUMask): kernels launched on other streams always
 * find free LDS and wave slots there instead of waiting for a 100 ms decode wave to retire.  reserve_cus = 0: a plain
 * non-blocking stream.  *stream_out is a hipStream_t; release it with mspa_stream_destroy.
 */
int mspa_stream_create_reserving(int32_t reserve_cus, void **stream_out);
int mspa_stream_destroy(void *stream);

int mspa_inflate_blocks_device(const void *src_dev, const int64_t *src_offsets_dev, const int64_t *src_bytes_dev,
                               int64_t src_capacity, int64_t n_blocks, int64_t block_bytes, void *dst_dev, int64_t dst_pitch,
                               int32_t *status_dev, uint32_t *work_dev, void *stream);
int mspa_png_unfilter_device(const void *raw_dev, int64_t raw_pitch, int64_t n_images, int32_t h, int32_t w, uint16_t *out_dev,
                             int32_t *status_dev, void *stream);
int mspa_png_pack_idat_host(const char *const *paths_host, int64_t n_files, int32_t h, int32_t w, void *dst_host,
                            int64_t dst_capacity, int64_t *offsets_host, int64_t *bytes_host, int32_t *status_host,
                            int64_t *capacity_needed, int32_t n_threads);

/*
 * K4 -- per-pair camera relations: the distance / yaw / pitch columns of CFR.process_scene's pair
 * table (CFR:176-189) and the relative-pose translation of CME.build_training_sample (CME:185-190).
 *
 *   E_aligned, Einv_aligned  [n_frames, 16] float64: A @ E and its inverse (host, numpy.linalg.inv)
 *   yaw, pitch               [n_frames] float64 degrees, extract_yaw_pitch (CFR:86-100) on the host
 *   pairs                    [n_pairs, 2] int32 (i, j)
 *   out                      [n_pairs, 6] float64: ||t_j - t_i||, yaw_j - yaw_i, pitch_j - pitch_i,
 *                            then the translation column of inv(E_i) @ E_j (camera-i axes)
 */
int mspa_pair_pose(const double *E_aligned, const double *Einv_aligned, const double *yaw,
                   const double *pitch, int32_t n_frames, const int32_t *pairs, int64_t n_pairs,
                   double *out, mspa_stream_t stream);

/*
 * extract_yaw_pitch (CFR:86-100) for a batch of frames: yaw = degrees(atan2(z_y, z_x)), pitch = degrees(asin(z_z / |z|)) of
 * the rotated z axis z = E[:3, 2].   E_aligned [n_frames, 16] f64  ->  out_yaw, out_pitch [n_frames] f64 (degrees).
 * Device libm: within a few ulp of NumPy, not bit-identical (float64 quantity; the host layer keeps a NumPy path for
 * callers that need the reference's exact bits).
 */
int mspa_extract_yaw_pitch(const double *E_aligned, int32_t n_frames, double *out_yaw, double *out_pitch,
                           mspa_stream_t stream);

/*
 * K5a -- TAPVid-3D tracks: camera->world (OM_C:446-454) and the normalised pinhole projection with
 * its validity test (TwoFrameVideoQAEngine.project_point, OM_C:293-315) for every (frame, point).
 *
 *   tracks_xyz  [T, P, 3] float64 camera-space      c2w  [T, 16] float64 = inv(extrinsics_w2c) (host)
 *   fx_fy_cx_cy_host  4 float64 on the HOST          H, W image size
 *   out_world [T, P, 3] f64,  out_uvn [T, P, 2] f64 (u/W, v/H),  out_ok [T, P] u8
 *   (1 = the reference would return coordinates, 0 = it returns None); each optional.
 */
int mspa_track_to_world(const double *tracks_xyz, const double *c2w, int32_t T, int32_t P,
                        const double *fx_fy_cx_cy_host, int32_t H, int32_t W, double *out_world,
                        double *out_uvn, uint8_t *out_ok, mspa_stream_t stream);

/*
 * K5b -- object displacement between two frames of one track point (OM_C:324-356) and the pair
 * distance used for binning (OM_C:484-498), for a list of (frame1, frame2, point) triples.
 *
 *   world [T, P, 3] f64 (K5a),  w2c / c2w [T, 16] f64,  triples [n, 3] int32
 *   out   [n, 5] f64: distance (0 when below obj_threshold), displacement in camera-1 axes (x, y, z;
 *         zeroed likewise), np.linalg.norm(axis=1)-form pair distance
 *   out_flags [n, 2] u8: point_moving, cam_moving
 */
int mspa_track_displacement(const double *world, const double *w2c, const double *c2w, int32_t T,
                            int32_t P, const int32_t *triples, int64_t n, double obj_threshold,
                            double cam_threshold, double *out, uint8_t *out_flags, mspa_stream_t stream);

/*
 * K5c -- pair mining of the object-movement head (OM_C:484-498): for each selected track point, the distance
 * between its world positions in every two of the frames it is visible in,
 * np.linalg.norm(points2 - points1, axis=1) over frame_pairs = [(i, j) for i < j] in that order.
 *   world [T, P, 3] f64 (K5a);  points [n_selected] i32;  frames = the points' visible-frame lists, concatenated,
 *   frame_offsets [n_selected+1] i32 into it;  n_frames_max = longest list;
 *   out_offsets [n_selected+1] i64 with out_offsets[s+1]-out_offsets[s] = n_s (n_s - 1) / 2  ->  out f64
 */
int mspa_track_pair_distances(const double *world, int32_t T, int32_t P, const int32_t *points,
                              const int32_t *frame_offsets, const int32_t *frames, int32_t n_selected,
                              int32_t n_frames_max, const int64_t *out_offsets, double *out, mspa_stream_t stream);

/*
 * K7 -- the accumulation inside rigid_body_segmentation (OM_C:49-92): cumulative_loss[i, j] = sum over
 * frames t >= 1 of |d_t(i,j) - d_{t-1}(i,j)| where that change exceeds smoothing_factor, d_t the Euclidean
 * distance between track points i and j at frame t.  The linkage / fcluster step stays with SciPy on the
 * host.   tracks_xyz [T, P, 3] f64 -> out_loss [P, P] f64 (symmetric, zero diagonal)
 */
int mspa_track_rigidity_loss(const double *tracks_xyz, int32_t T, int32_t P, double smoothing_factor,
                             double *out_loss, mspa_stream_t stream);

/*
 * K8 -- per (object, image) extent of the object's visible vertices: what compute_coverage
 * (object_perception/single_object_coverage_finder.py:56-65) takes off `image mask & object mask`, for all
 * three axes at once.  The coverage of a union of images is max-of-max minus min-of-min of these, so the
 * minimal-combination search (COV:76-220) runs on them without ever forming a union mask.
 *   vis_bits [n_images, n_words] u64 (K1's bitsets, or rows packed from the visibility parquet),
 *   xyz [n_vertices, 3] f64 (axis-aligned scene points), objects as CSR: obj_offsets [n_objects+1] i32 into
 *   obj_vertices [n_list_entries] (vertex ids, each < n_vertices; n_list_entries = obj_offsets[n_objects])
 *   -> out_lo / out_hi [n_objects, n_images, 3] f64 (+inf / -inf where the image sees none of the object),
 *      out_count [n_objects, n_images] i32 (= intersection_count of compute_object_visibility.py:119-121)
 */
int mspa_object_extents(const uint64_t *vis_bits, int32_t n_images, int64_t n_words, const double *xyz,
                        int64_t n_vertices, const int32_t *obj_offsets, const int32_t *obj_vertices,
                        int64_t n_list_entries, int32_t n_objects, double *out_lo, double *out_hi, int32_t *out_count,
                        mspa_stream_t stream);

/*
 * K6a -- correspondence extraction on K1's bitsets: for every selection (image1, image2, j) the j-th
 * vertex (ascending index) visible in both images, i.e. element j of np.intersect1d(points1, points2)
 * (visual correspondence, VC_C:303-313); with image1 == image2 it is element j of that image's
 * visible-vertex list (depth heads, DE_C:190-198).  -1 when fewer than j+1 vertices qualify.
 *   bits [n_images, n_words] uint64,  selections [n, 3] int32,  out_vertex [n] int32
 */
int mspa_select_common_point(const uint64_t *bits, int32_t n_images, int64_t n_words,
                             const int32_t *selections, int64_t n, int32_t *out_vertex,
                             mspa_stream_t stream);

/*
 * K6b -- SceneInfoHandler.get_point_2d_coordinates_in_image (IH:291-305) for a batch of
 * (vertex, image) samples: un-rounded (u, v), camera depth and the check_point_visibility flag.
 *   xyz as in mspa_vertex_visibility; cam_mats [n_images, MSPA_CAM_MATS, 16]; depth [n_images, dh, dw]
 *   samples [n, 2] int32 (vertex, image);  out_uv [n, 2] f64, out_depth [n] f64, out_visible [n] u8
 */
int mspa_project_samples(const double *xyz, int64_t n_points, int64_t point_stride, int64_t comp_stride,
                         const double *cam_mats, int32_t n_images, const uint16_t *depth, int32_t dh,
                         int32_t dw, int32_t H, int32_t W, const int32_t *samples, int64_t n,
                         double *out_uv, double *out_depth, uint8_t *out_visible, mspa_stream_t stream);
/* ... with the handler's depth_value_scale (IH:368) instead of the default 0.001 */
int mspa_project_samples_ex(const double *xyz, int64_t n_points, int64_t point_stride, int64_t comp_stride,
                            const double *cam_mats, int32_t n_images, const uint16_t *depth, int32_t dh,
                            int32_t dw, int32_t H, int32_t W, double depth_value_scale, const int32_t *samples, int64_t n,
                            double *out_uv, double *out_depth, uint8_t *out_visible, mspa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MSPA_H */
