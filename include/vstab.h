/*
 * vstab.h -- C ABI of libvstab.so, the MI355X (gfx950) implementation of the
 * dense-flow stabilization hot path of nomadoor/ComfyUI-Video-Stabilizer.
 *
 * The reference is pure Python and has no FFI of its own; every entry point
 * below replaces a group of OpenCV/NumPy calls made by the reference's Python
 * (file:line cited per function, paths relative to the reference root).  The
 * Python host layer (comfyui-video-stabilizer_amd/) binds these with ctypes;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every call returns 0 on success, non-zero on failure;
 *     vstab_last_error() returns a thread-local message for the last failure.
 *     A failure detected by a kernel (e.g. an expired dependency wait inside the
 *     DIS patch search) is recorded in a host-visible status word and reported
 *     -- return value 3 -- by the next call that synchronises with the host
 *     (vstab_sample_fit_batch, vstab_synchronize).
 *   - "dev" pointers are HIP device pointers owned by the caller (e.g.
 *     torch.Tensor.data_ptr()); "host" pointers are ordinary host memory.
 *     The library never frees caller memory and never returns owned memory.
 *   - work is enqueued on the context's stream (vstab_set_stream; default: the
 *     null stream).  Calls that return host results synchronise that stream;
 *     the others are asynchronous.
 *   - images are row-major, channel-last: frames [N,H,W,3] f32 0..1,
 *     masks [N,H,W] f32, gray [N,h,w] u8, flow [P,h,w,2] f32 (x,y).
 *   - 3x3 matrices are row-major, 9 values.
 */
#ifndef VSTAB_H
#define VSTAB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSTAB_ABI_VERSION 1

typedef struct vstab_ctx vstab_ctx;

enum vstab_interp { VSTAB_INTERP_BILINEAR = 0, VSTAB_INTERP_BICUBIC = 1 };
/* sub-pixel model of the interpolating warp: Q5 = OpenCV legacy kernels (source
 * coordinates rounded to 1/32 px), EXACT = full f32 coordinates (OpenCV >= 4.11
 * INTER_LINEAR kernels; bilinear only). */
enum vstab_subpix { VSTAB_SUBPIX_Q5 = 0, VSTAB_SUBPIX_EXACT = 1 };
enum vstab_mode { VSTAB_MODE_TRANSLATION = 0, VSTAB_MODE_SIMILARITY = 1, VSTAB_MODE_PERSPECTIVE = 2 };

/* ---- context ------------------------------------------------------------ */
int vstab_abi_version(void);
const char* vstab_last_error(void);
/* device < 0: use the current HIP device */
/* 1 in the test build (-DVSTAB_TEST_HOOKS -> lib/libvstab_hooks.so: the fault injectors VSTAB_DEBUG_PLAN_PERTURB,
 * VSTAB_DEBUG_PIS_SPIN_LIMIT, VSTAB_DEBUG_XFER_SPAWN_FAIL are read from the environment), 0 in the shipped library,
 * which reads none of them. */
int vstab_test_hooks(void);
int vstab_create(vstab_ctx** out, int device);
int vstab_destroy(vstab_ctx* ctx);
/* hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = null stream */
int vstab_set_stream(vstab_ctx* ctx, void* hip_stream);
int vstab_synchronize(vstab_ctx* ctx);
/* Kernel timing with HIP events recorded on the call's own stream, per kind of call ("warp", "warp_blur",
 * "gray", "dis", "fit", "gftt", "lk", "phase").  vstab_set_timing(ctx, 1) enables it and clears the totals;
 * vstab_set_timing(ctx, 2) also brackets the stages INSIDE a DIS call ("dis_prep", "dis_pis4_L<level>",
 * "dis_level_L<level>", "dis_final": for a measurement pass of its own, the events lengthen the chain).
 * vstab_set_timing(ctx, 3) records events around the warp launches only ("warp", "warp_blur"): what a timed loop keeps.
 * vstab_last_kernel_ms: milliseconds of the most recent call of that kind (waits for it to finish).
 * vstab_kernel_ms_stats: sum and number of all calls of that kind since timing was enabled -- no
 * synchronisation is needed inside a timed loop, bench.py reads the totals after its closing fence
 * for the roofline line. */
int vstab_set_timing(vstab_ctx* ctx, int enabled);
int vstab_last_kernel_ms(vstab_ctx* ctx, const char* kind, float* ms_out);
int vstab_kernel_ms_stats(vstab_ctx* ctx, const char* kind, double* total_ms, int* launches);

/* ---- F0 / F16: the node boundary's bulk transfers ---------------------------------------------
 * ComfyUI hands CPU tensors in and takes CPU tensors back (nodes/stabilizer_utils.py:96-147, :200-221): 6.37 GB in and
 * 8.49 GB out for a 256 x 1080p clip.  vstab_upload / vstab_download move `bytes` between ordinary (pageable) host
 * memory and device memory through a ring of page-locked buffers filled / drained by several host threads
 * (VSTAB_XFER_THREADS, default 16, capped at the core count) while the DMA engine moves the previous chunk.
 * vstab_upload returns when every byte has left host_src (the copy into dev_dst completes asynchronously; work
 * enqueued afterwards on the context's stream is ordered behind it).  vstab_download starts after the work already
 * enqueued on the context's stream and returns when host_dst is complete. */
int vstab_upload(vstab_ctx* ctx, const void* host_src, void* dev_dst, size_t bytes);
int vstab_download(vstab_ctx* ctx, const void* dev_src, void* host_dst, size_t bytes);
/* The same two transfers with fewer bytes on PCIe where the DATA allow it; the destination holds the source's bits either way.
 * vstab_upload_f32_coded: `count` float32 values.  A ComfyUI IMAGE decoded from 8-bit video holds float32(k) / float32(255)
 * and nothing else (what nodes/stabilizer_utils.py:122-126 itself produces for uint8 input): a chunk (2^25 values) whose values
 * all have exactly the bits of such a quotient crosses as bytes and is expanded on the device by the same correctly rounded
 * division; the first chunk with a value of any other kind, and everything behind it, crosses as float32.  *coded_chunks
 * (may be NULL) = chunks that crossed as bytes.  dev_dst 16-byte aligned.
 * vstab_download_mask_coded: `count` float32 mask values (nodes/stabilizer_utils.py:1055-1077).  If every value is 0.0f or
 * 1.0f (the Flow node's mask, nodes/video_stabilizer_flow.py:583-586) they cross as bytes and the host threads expand them;
 * otherwise (Motion Apply's soft mask under motion blur) this is vstab_download.  *coded (may be NULL) = 1 / 0.
 * vstab_upload_u8_as_f32: `count` uint8 values k arrive on the device as float32(k) / float32(255) -- _to_numpy_frame's uint8
 * branch, `arr.astype(np.float32); arr /= 255.0` (nodes/stabilizer_utils.py:122-126), without the host ever holding the floats.
 */
int vstab_upload_f32_coded(vstab_ctx* ctx, const float* host_src, float* dev_dst, size_t count, size_t* coded_chunks);
int vstab_upload_u8_as_f32(vstab_ctx* ctx, const unsigned char* host_src, float* dev_dst, size_t count);
int vstab_download_mask_coded(vstab_ctx* ctx, const float* dev_src, float* host_dst, size_t count, int* coded);
/* The motion-blur warp's soft mask (nodes/motion_apply.py:195-199) holds 1 - c / S for c = 0 .. S covered samples, values below
 * 1e-3 set to 0: S + 1 different floats.  vstab_download_mask_levels(levels = S) sends c as a byte and the host threads look the
 * float up (formed with the same IEEE float32 subtraction and division); a mask with any other value takes vstab_download.
 * levels = 1 is vstab_download_mask_coded. */
int vstab_download_mask_levels(vstab_ctx* ctx, const float* dev_src, float* host_dst, size_t count, int levels, int* coded);

/* ---- F13 / A3: per-frame warp with padding mask ---------------------------
 * Replaces the loop at nodes/video_stabilizer_flow.py:560-588 and
 * nodes/motion_apply.py:92-120:
 *   cv2.warpPerspective(frame, M, (out_w,out_h), INTER_LINEAR|INTER_CUBIC, BORDER_CONSTANT, rgb)
 *   cv2.warpPerspective(ones,  M, (out_w,out_h), INTER_NEAREST, BORDER_CONSTANT, 0)
 *   mask = 1 - (content > 0.5); mask[mask < 1e-3] = 0
 * src        dev  [n, src_h, src_w, 3] f32
 * matrices   host [n, 9] f32, forward (source -> output) as the reference passes them
 * border_rgb host [3] f32 (padding_rgb / 255 computed in f32)
 * dst        dev  [n, out_h, out_w, 3] f32
 * mask       dev  [n, out_h, out_w] f32 or NULL (masks_zero path, motion_apply.py:103-106)
 * pad_count  dev  [n] u32 or NULL: number of mask==1 pixels per frame
 *                 (mask.mean() = count/(out_h*out_w), flow.py:587)
 */
int vstab_warp_batch(vstab_ctx* ctx, const float* src, int n, int src_h, int src_w,
                     const float* matrices, int out_h, int out_w, int interp,
                     const float* border_rgb, int subpix, float* dst, float* mask,
                     uint32_t* pad_count);

/* ---- A5: multi-sample motion-blur warp -------------------------------------
 * Replaces nodes/motion_apply.py:125-202 (_blurred_matrix_samples +
 * _warp_with_motion_blur).  matrices: host [n,9] f64 motion matrices; the
 * library forms the S sample matrices M[i] + (M[i+1]-M[i])*t_k, t = linspace(0,
 * blur, S) in f64, casts each to f32 and inverts in f64 exactly as the plain warp does.
 * ts: host [samples] f64 = numpy.linspace(0, blur, samples) supplied by the caller
 * (keeps NumPy's own linspace rounding on the Python side of the boundary).
 * Output frame = sum_k warp_k / float(samples); mask = 1 - coverage_sum/samples,
 * values < 1e-3 -> 0.  mask may be NULL.
 */
int vstab_warp_blur_batch(vstab_ctx* ctx, const float* src, int n, int src_h, int src_w,
                          const double* matrices, const double* ts, int samples, int out_h,
                          int out_w, int interp, const float* border_rgb, int subpix,
                          float* dst, float* mask);

/* Shardable form of vstab_warp_blur_batch ("blur needs the next frame's matrix (replicated JSON) but never
 * another rank's pixels"): src holds frames [clip_first, clip_first + n) of a clip whose motion_meta has clip_total
 * matrices; clip_matrices is the whole host [clip_total,9] f64 table, so the sample delta of a frame at a shard edge
 * uses its true neighbour (M[i+1] - M[i]; the clip's last frame: M[i] - M[i-1], motion_apply.py:129-132).
 * vstab_warp_blur_batch(…, matrices, …) == vstab_warp_blur_clip_batch(…, matrices, n, 0, …). */
int vstab_warp_blur_clip_batch(vstab_ctx* ctx, const float* src, int n, int src_h, int src_w,
                               const double* clip_matrices, int clip_total, int clip_first,
                               const double* ts, int samples, int out_h, int out_w, int interp,
                               const float* border_rgb, int subpix, float* dst, float* mask);

/* ---- A4: the shutter sample matrices themselves (host arithmetic only, no GPU involved) --------
 * nodes/motion_apply.py:125-134 (_blurred_matrix_samples) followed by the float32 cast of motion_apply.py:172, for
 * frames [first, first+count) of a clip of `total` f64 matrices: out host [count, S', 9] f32 with S' = samples, or 1
 * for a single-matrix clip (motion_apply.py:126-127).  This is the routine vstab_warp_blur_*batch runs internally;
 * exported so that the reference-generated golden vectors pin the shipped arithmetic (tests/test_abi_cpu.py). */
int vstab_blur_sample_matrices(const double* matrices, int total, int first, int count,
                               const double* ts, int samples, float* out);

/* ---- F2: grayscale + INTER_AREA downscale to the estimation size ------------
 * Replaces nodes/stabilizer_utils.py:236-242 (_make_gray: cv2.cvtColor RGB2GRAY
 * on f32, clip(gray*255,0,255).astype(uint8)) and :271-276 (cv2.resize INTER_AREA).
 * work_h/work_w == src_h/src_w means "no downscale" (working_size None).
 * frames dev [n,src_h,src_w,3] f32 -> gray dev [n,work_h,work_w] u8.
 */
int vstab_gray_downscale(vstab_ctx* ctx, const float* frames, int n, int src_h, int src_w,
                         int work_h, int work_w, uint8_t* gray);

/* ---- F0 + F2: the same pass with the value-range sniff of the input adaptation folded in ------
 * nodes/stabilizer_utils.py:127-131 decides PER FRAME whether float input is 0..255 (`float(arr.max()) > 1.5` -> /255):
 * one more full read of every frame.  The gray pass reads every sample anyway, so it can report the per-frame maximum
 * (NaN if the frame holds one, as numpy's max) in the same 24.9 MB read: frame_max dev [n] f32.  The caller runs the
 * estimation optimistically on the tensor as given and, in the rare case that a maximum exceeds 1.5, rescales those
 * frames and repeats it (host_math.resolve_value_range).
 * vstab_frame_range: the sniff alone (Motion Apply has no estimation pass), an HBM-bound read of the clip. */
int vstab_gray_downscale_range(vstab_ctx* ctx, const float* frames, int n, int src_h, int src_w,
                               int work_h, int work_w, uint8_t* gray, float* frame_max);
/* The per-frame maxima of the latest vstab_gray_downscale_range call, on the host (n = that call's frame count): the
 * kernel that forms them also writes them into coherent host memory, so this waits for that kernel only -- nothing is
 * copied and no event is recorded on the stream.  The host side of `float(arr.max()) > 1.5` per frame
 * (nodes/stabilizer_utils.py:127-131). */
int vstab_last_frame_peaks(vstab_ctx* ctx, int n, float* out);
int vstab_frame_range(vstab_ctx* ctx, const float* frames, int n, int h, int w, float* frame_max);
/* The rule itself, out of place: out[f] = frames[f] / 255 (IEEE float32 division, numpy's `arr /= 255.0`) where
 * frame_max[f] > 1.5, else a copy.  frames, frame_max, out: dev.  The caller's tensor is never modified. */
int vstab_apply_value_range(vstab_ctx* ctx, const float* frames, int n, int h, int w,
                            const float* frame_max, float* out);

/* ---- F3 (+F4): DIS dense optical flow over consecutive pairs ---------------
 * Replaces cv2.DISOpticalFlow (PRESET_MEDIUM, finestScale 2, patchSize 8,
 * patchStride 4, spatial propagation) created at nodes/video_stabilizer_flow.py:82-86
 * and called at :140, for pairs (i, i+1), i in [0, n-1).
 * gray        dev [n,h,w] u8
 * flow        dev [n-1,h,w,2] f32 or NULL (full field; tests / debugging)
 * grid_flow   dev [n-1,gh,gw,2] f32 or NULL: the flow sampled at y=0,step,.. x=0,step,..
 *             (gh = ceil(h/step), gw = ceil(w/step)) -- the only values the
 *             reference consumes (flow.py:141-147)
 */
int vstab_dis_flow_batch(vstab_ctx* ctx, const uint8_t* gray, int n, int h, int w,
                         float* flow, float* grid_flow, int sample_step);

/* OpenCV's DIS object is stateful for tiny images: its first calc() may auto-select patch size / scales
 * (autoSelectPatchSizeAndScales) and keeps the new finest scale for the following calls.  The reference
 * creates one object per clip (flow.py:316), so the first pair of a CLIP can use a different pyramid than
 * the rest.  A rank that processes a later shard of the clip clears this flag (default: 1). */
int vstab_dis_set_clip_start(vstab_ctx* ctx, int first_pair_is_clip_start);

/* ---- F4 + F5: model fit on the sampled flow -------------------------------
 * Replaces nodes/video_stabilizer_flow.py:141-210 for every pair: RANSAC
 * homography (cv2.findHomography 2.5 px / 2000 / 0.992, accept >= 0.15),
 * RANSAC similarity (cv2.estimateAffinePartial2D 2.0 px / 2000 / 0.992, accept
 * >= 0.1) and per-axis median translation.  All candidate models at or below
 * `requested_mode` are evaluated for every pair; the host applies the
 * sequential "sticky active_mode" rule (flow.py:324-339) to pick one.
 * grid_flow dev [pairs,gh,gw,2]; step = sample stride used to build the grid.
 * results host [pairs * 3] records indexed [pair*3 + mode].
 */
typedef struct vstab_fit_record {
    float matrix[9];   /* 3x3 f32 at working resolution */
    double confidence; /* inlier ratio (translation: valid/total), as the reference's Python float */
    double residual;   /* mean |model(p) - q| over all valid samples, both axes */
    int32_t accepted;  /* 1 if this mode's acceptance test passed */
    int32_t computed;  /* 1 if this mode was evaluated */
    int32_t valid_points; /* finite samples (flow.py:150-154) */
    int32_t total_points;
} vstab_fit_record;
int vstab_sample_fit_batch(vstab_ctx* ctx, const float* grid_flow, int pairs, int gh, int gw,
                           int step, int requested_mode, vstab_fit_record* results);
/* The same in two halves, so that work can be queued on the stream between the launch and the host's wait:
 * _begin launches the fits and queues the download of the records behind them; vstab_fit_records_device is the
 * device copy of those records ([pairs * 3], valid until the next fit call of this context); _end waits for the
 * download only (not for anything queued after _begin) and hands out the host records. */
int vstab_sample_fit_batch_begin(vstab_ctx* ctx, const float* grid_flow, int pairs, int gh, int gw,
                                 int step, int requested_mode);
const vstab_fit_record* vstab_fit_records_device(vstab_ctx* ctx);
int vstab_sample_fit_batch_end(vstab_ctx* ctx, int pairs, vstab_fit_record* results);

/* ---- N1 (crop framing): coverage analysis for the keep_fov crop solver ---------
 * Replaces the per-frame cv2 calls of nodes/stabilizer_utils.py:611-643
 * (finalize_with_masks: warpPerspective(ones, NEAREST) > 0.5, dilate 3x3, erode 3x3, bounding box of
 * the remaining content) and :763-787 (_refine_no_padding_crop: AND of the coverage masks of all
 * frames, erode 3x3).  One pass over the n matrices serves both.
 * matrices host [n,9] f32 (source -> output);  source src_h x src_w, output out_h x out_w.
 * bbox     host [n,4] i32: x_min, y_min, x_max, y_max of the closed coverage, or -1,-1,-1,-1 if empty
 * common   host [out_h*out_w] u8: 1 where every frame covers the pixel, after the 3x3 erosion
 *          (pixels outside the image do not constrain the erosion, as cv2.erode's default border)
 */
int vstab_crop_analysis(vstab_ctx* ctx, const float* matrices, int n, int src_h, int src_w,
                        int out_h, int out_w, int32_t* bbox, uint8_t* common);

/* Motion Apply's crop framing (nodes/motion_apply.py:205-227, _common_valid_mask): AND over all frames of
 * warpPerspective(ones, M, INTER_NEAREST) > 0.5 -- no morphology, no bounding boxes.
 * matrices host [n,9] f32;  common host [out_h*out_w] u8 (1 = covered by every frame). */
int vstab_common_coverage(vstab_ctx* ctx, const float* matrices, int n, int src_h, int src_w,
                          int out_h, int out_w, uint8_t* common);

/* ---- N1 (Classic estimator): sparse features + pyramidal LK ------------------
 * Replaces nodes/video_stabilizer_classic.py:76-96 (`_estimate_motion_pair`) for a whole clip.
 *
 * vstab_gftt_batch: cv2.goodFeaturesToTrack(gray, maxCorners, qualityLevel, minDistance, blockSize)
 * (classic.py:76-83: 400 / 0.01 / 7 / 21; no mask, min-eigenvalue score, 3x3 Sobel) on every frame.
 *   gray    dev [n,h,w] u8
 *   corners dev [n,max_corners,2] f32 (x, y), strongest first;  counts dev [n] i32
 *
 * vstab_lk_track_batch: cv2.calcOpticalFlowPyrLK(frame i, frame i+1, corners of frame i, None,
 * winSize=(win,win), maxLevel, criteria=(EPS|COUNT, max_count, epsilon)) (classic.py:88-96: 31 / 3 /
 * 50 / 0.01) for the n-1 consecutive pairs of the clip.
 *   points      dev [n-1,max_points,2] f32, counts dev [n-1] i32 (rows of vstab_gftt_batch's output)
 *   point_pairs dev [n-1,max_points,4] f32: prev.x, prev.y, next.x, next.y; next = NaN where the
 *               tracker's status is 0 -- the input layout of vstab_points_fit_batch
 *   next_points dev [n-1,max_points,2] f32 or NULL (raw tracker output, also for status 0)
 *   status      dev [n-1,max_points] u8 or NULL
 * vstab_lk_levels: number of pyramid levels above level 0 that buildOpticalFlowPyramid keeps.
 *
 * vstab_points_fit_batch: the model-fit cascade of classic.py:98-160 on the tracked pairs: fewer
 * than 12 features or fewer than 8 tracked points -> no candidate; otherwise the same three
 * estimators as vstab_sample_fit_batch (translation confidence = tracked / features).
 *   results host [pairs*3] records indexed [pair*3 + mode] (residual is filled but unused by Classic)
 */
int vstab_gftt_batch(vstab_ctx* ctx, const uint8_t* gray, int n, int h, int w, int max_corners,
                     double quality, double min_distance, int block_size, float* corners, int* counts);
int vstab_lk_levels(int h, int w, int win, int max_level);
int vstab_lk_track_batch(vstab_ctx* ctx, const uint8_t* gray, int n, int h, int w, const float* points,
                         const int* counts, int max_points, int win, int max_level, int max_count,
                         double epsilon, float* point_pairs, float* next_points, uint8_t* status);
int vstab_points_fit_batch(vstab_ctx* ctx, const float* point_pairs, const int* counts, int pairs,
                           int max_points, int requested_mode, vstab_fit_record* results);

/* ---- N2 (fallback estimator): phase correlation -------------------------------------
 * Replaces cv2.phaseCorrelate(prev_gray.astype(float32), curr_gray.astype(float32)) of
 * nodes/video_stabilizer_flow.py:110-130 (`_estimate_motion_phase_correlate`, the branch of the pair loop
 * flow.py:325-330 taken when no dense-flow backend could be created, flow.py:90-107) for the n-1 consecutive
 * pairs of a clip: zero-pad to the optimal DFT size, normalised cross-power spectrum, inverse DFT, first
 * maximum of the centred correlation surface, 5x5 weighted centroid.
 *   gray    dev [n,h,w] u8 (estimation images; padded sizes up to 2048)
 *   results host [pairs*3] records indexed [pair*3 + mode], or NULL: only the translation row is
 *           computed/accepted (matrix = [1 0 tx; 0 1 ty; 0 0 1], confidence = response, residual 0; a
 *           non-finite result becomes tx = ty = confidence = 0 as flow.py:115-120 does) -- the reference reports
 *           every pair of this estimator as "translation" whatever mode was requested
 *   shifts  host [pairs*3] f64 (tx, ty, response) as cv2.phaseCorrelate returns them, or NULL
 * Synchronises with the host (results are host values).  Timing kind: "phase".
 */
int vstab_phase_correlate_batch(vstab_ctx* ctx, const uint8_t* gray, int n, int h, int w,
                                vstab_fit_record* results, double* shifts);

/* ---- F7 + F8: trajectory (prefix sum, box smoothing, strength blend), fp64 ---
 * Replaces nodes/video_stabilizer_flow.py:356-371 and
 * nodes/stabilizer_utils.py:361-383 (_smooth_path: moving average, edge padded,
 * window from fps).  deltas host [n-1,p]; path/target host [n,p].
 * Host arithmetic since round 4 (a few thousand doubles; ctx is not used): the same operation order as the
 * device plan below, so the two agree bit for bit on equal deltas.
 */
int vstab_trajectory(vstab_ctx* ctx, const double* deltas, int n, int p, double smooth,
                     double fps, double strength, int camera_lock, double* path,
                     double* target);

/* ---- F6-F12 on the device, speculatively: fit records -> the warp's transform table, no host round trip ----
 * Replaces the stretch of nodes/video_stabilizer_flow.py:324-371 and :472-521 between the last model fit and the first
 * warp for framing_mode "crop_and_pad" / "expand" (any of the three models): sticky active_mode walk, rescale to full
 * resolution, parameter deltas of the requested model, path / _smooth_path / strength blend, _params_to_matrix,
 * _compute_bounding_boxes, the common region, the recentring shift and final = T @ M (float32), inverted as
 * cv2.warpPerspective inverts it.  One kernel on the context's stream.
 *   d_records  dev [pairs*3] (vstab_fit_records_device, or a gathered table of the whole clip)
 *   up, down   as for vstab_transitions_to_params (NULL: estimated at full size)
 * The plan stays on the device for vstab_warp_batch_planned; its float32 final matrices, path, target and the common
 * region (x0, y0, x1, y1) are downloaded behind the kernel and handed out by vstab_flow_plan_result (waits for that
 * download only).  The device forms atan2 / log / exp / cos / sin with its own fp64 library, the reference with the host's
 * libm (a perspective plan has no such call, but its final = T @ M holds sums of two inexact float32 terms, which NumPy's matmul
 * may or may not fuse): the caller computes the plan on the host as well (vstab_transitions_to_params ... vstab_bounding_boxes),
 * compares the final matrices bit for bit and warps a frame again where they differ -- see flow_pipeline.py. */
/* Optional, before vstab_flow_plan_device: the planned warp's padded-pixel count array (dev [n] u32); the plan kernel zeroes
 * it, and vstab_warp_batch_planned with the same pointer skips its own fill -- one launch less between plan and warp. */
int vstab_flow_plan_zero_counts(vstab_ctx* ctx, uint32_t* pad_count, int n);
/* framing (last argument of vstab_flow_plan_device): 0 = crop_and_pad -- `region` = the frames' common region, the matrices are
 * recentred on it (flow.py:500-529); 1 = expand -- `region` = the frames' union x_min, y_min, x_max, y_max, the matrices are
 * shifted by (-x_min, -y_min) (stabilizer_utils.py:386-406); the caller forms the canvas size ceil(x_max - x_min) x
 * ceil(y_max - y_min) from vstab_flow_plan_result's region before it queues the warp. */
int vstab_flow_plan_device(vstab_ctx* ctx, const vstab_fit_record* d_records, int pairs, int requested_mode,
                           const double* up, const double* down, double smooth, double fps, double strength,
                           int camera_lock, int width, int height, int segments, const int* seg_pairs, int seg_rows, int framing);
/* segments = 0: d_records is [pairs*3].  segments = world > 0 (multi-GPU): d_records is the receive buffer of the ranks'
 * all-gather as RCCL leaves it -- one block of seg_rows pairs per rank, of which the first seg_pairs[r] are valid -- so the
 * gathered table feeds the plan where it lands (no compaction pass, no host copy before the warp). */
/* Copies the records of the last vstab_sample_fit_batch_begin (device) to dst (device, >= pairs*3 records), stream-ordered:
 * how a rank places its records in its all-gather send buffer. */
int vstab_fit_records_copy(vstab_ctx* ctx, void* dst_dev, int pairs);
int vstab_flow_plan_result(vstab_ctx* ctx, int frames, float* final32, double* path, double* target, double* region);
/* vstab_warp_batch for frames [first, first + n) of the clip planned by the last vstab_flow_plan_device of this context
 * (src: those n frames). */
int vstab_warp_batch_planned(vstab_ctx* ctx, const float* src, int first, int n, int src_h, int src_w, int out_h,
                             int out_w, int interp, const float* border_rgb, int subpix, float* dst, float* mask,
                             uint32_t* pad_count);
/* The per-frame padded-pixel counts of the latest vstab_warp_batch / vstab_warp_batch_planned call that was given a
 * pad_count array, on the host (n = that call's frame count): a one-workgroup kernel behind the warp mirrors them into
 * coherent host memory, this call waits for it -- the host side of `mask.mean()` per frame (video_stabilizer_flow.py:583-588)
 * without a copy or a stream synchronisation of the caller's. */
int vstab_last_pad_counts(vstab_ctx* ctx, int n, uint32_t* out);

/* ---- F6 / F9 host helper: element-wise libm over fp64 arrays (host pointers, no GPU involved) ----
 * nodes/stabilizer_utils.py:300-358 (_matrix_to_params / _params_to_matrix) call math.sqrt/atan2/log and
 * math.exp/cos/sin per frame; this runs the same libm functions over a whole clip in one call.
 * b is only read by VSTAB_HOST_ATAN2 (out = atan2(a, b)). */
enum { VSTAB_HOST_SQRT = 0, VSTAB_HOST_ATAN2 = 1, VSTAB_HOST_LOG = 2, VSTAB_HOST_EXP = 3, VSTAB_HOST_COS = 4, VSTAB_HOST_SIN = 5 };
int vstab_host_math(int op, const double* a, const double* b, int n, double* out);

/* ---- F6 / F9 for a whole clip (host pointers, no GPU involved) ----
 * vstab_transitions_to_params: nodes/video_stabilizer_flow.py:340-346 per transition --
 *   _rescale_transform_to_full (stabilizer_utils.py:279-297; skipped when up == down == NULL: estimated at full size)
 *   then _matrix_to_params (stabilizer_utils.py:300-324) of the requested mode.
 *   work_mats host [count,9] f32 (working resolution); up = {1/sx, 1/sy, 1}, down = {sx, sy, 1} with
 *   sx = working_w / source_w, sy = working_h / source_h (fp64, as the reference forms them);
 *   full_mats host [count,9] f32; params host [count, 2 | 4 | 8] f64.
 * vstab_params_to_matrices: _params_to_matrix (stabilizer_utils.py:327-358): params host [count, 2|4|8] f64 ->
 *   mats host [count,9] f32.  mode = VSTAB_MODE_*.
 * Replicated on every rank of a multi-GPU run for the whole clip (the serial part of a rank's step): one call each
 * instead of ~25 NumPy / ctypes calls.  Same libm, same rounding points as the per-item Python forms. */
int vstab_transitions_to_params(const float* work_mats, int count, int mode, const double* up, const double* down,
                                float* full_mats, double* params);
int vstab_params_to_matrices(const double* params, int count, int mode, float* mats);

/* F10 for a whole clip (host pointers): _compute_bounding_boxes (stabilizer_utils.py:1010-1034) -- the four frame corners
 * through each f32 matrix in fp64, projective division, per-axis min / max (NaN-propagating, as numpy.minimum / maximum).
 * mats host [count,9] f32; mins, maxs host [count,2] f64. */
int vstab_bounding_boxes(const float* mats, int count, double width, double height, double* mins, double* maxs);

#ifdef __cplusplus
}
#endif
#endif /* VSTAB_H */
