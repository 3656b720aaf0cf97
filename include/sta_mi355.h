/*
 * sta_mi355.h - C ABI of the MI355X-native Symmetric Two-view Association (STA) frontend.
 *
 * Drop-in boundary for ONE hot path of ViSTA-SLAM: the STA forward pass.  Every entry point
 * below replaces a Python method of the reference `SymmetricTwoViewAssociation`
 * (vista_slam/sta_model/sta_model.py) or its only native sub-boundary (curope):
 *
 *   sta_create / sta_load_tensor / sta_finalize_weights
 *        <- STA() + load_state_dict(ckpt['model'], strict=True) + .to(device).eval()
 *           (vista_slam/slam.py:95-106).  Tensor names == reference state_dict keys.
 *   sta_encode        <- _encode_image(image, true_shape, normalize=False)
 *                        (sta_model.py:163-174, called from slam.py:144)
 *   sta_decode        <- _decode_stereo(feat1, feat2, pos1, pos2)   (positions = the patch grid; sta_decode_pos: any positions)
 *                        (sta_model.py:177-244, called from slam.py:162)
 *   sta_head_pose     <- head_pose_s(tok[:,0,:])        (heads/pose_head.py:109-120, slam.py:165)
 *   sta_head_pts      <- head_pts(list14, true_shape)   (heads/dpt_head.py:34-66 +
 *                        heads/postprocess.py:10-62 + utils/misc.py:36-78, slam.py:179-180)
 *   sta_forward_pair  <- forward({'main_view','neighbor_views':[b],'loop_views':[]})
 *                        (sta_model.py:247-291)
 *   sta_rope2d_inplace<- curope.rope_2d(tokens, positions, base, fwd)
 *                        (pos_embed/curope/curope.cpp:49-65, kernels.cu:84-108)
 *
 * Conventions
 *   - Plain C types only.  All *_dev pointers are device (HBM) pointers owned by the caller
 *     (the Python shim passes torch-ROCm tensor .data_ptr()).  The library owns only weights
 *     and an internal workspace per caller stream; a workspace grows on the first call of a new shape
 *     and is then reused (no allocation in steady state).
 *   - All work is enqueued on the caller-supplied hipStream_t (`stream`, passed as void*);
 *     no host synchronisation inside, so ordering with surrounding torch ops is preserved.
 *   - Return value: 0 on success, negative on error; sta_last_error() returns a thread-local
 *     message.  A handle is not thread-safe (one host thread at a time; several STREAMS are fine, see
 *     "Streams and concurrency" below); one handle per device.
 *   - Images are NCHW fp32 in [-1,1]; H and W must be multiples of 16.  Portrait frames (H > W) are tokenised row-major
 *     as they are (PatchEmbedDust3R, patch_embed.py:15-26) and every per-pixel output of this ABI is in IMAGE orientation
 *     [.., H, W, ..].  The reference's head wrapper returns portrait outputs as transposed VIEWS of exactly that memory
 *     (`transposed(head(decout, (H, W)))`, utils/misc.py:60-61,81); the Python shim applies the same swapaxes(1, 2), and
 *     sta_regress_views evaluates the shared intrinsics the way the reference does on those views (see there).
 *   - Token tensors are row-major fp32: encoder [B, N, enc_dim], decoder [B, N+1, dec_dim]
 *     (row 0 of every decoder sequence is the pose token, sta_model.py:206-219).
 */
#ifndef STA_MI355_H
#define STA_MI355_H

#include <stdint.h>

/* Every entry point carries STA_API = default ELF visibility; the library itself is compiled with -fvisibility=hidden, so its
 * dynamic symbol table is these declarations and nothing else (no kernel launch stubs, no helper functions:
 * tests/test_cabi_symbols.py compares `nm -D` with this header). */
#ifndef STA_API
#define STA_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sta_handle sta_handle;

/* Arithmetic policy of every GEMM / attention contraction (fp32 accumulate always).
 * gfx950 has no TF32/XF32 MFMA (the reference runs TF32: sta_model.py:5). */
enum {
    STA_PREC_F16   = 1,  /* fp16 x fp16 -> fp32 MFMA, one product  (10-bit mantissa == TF32 class) */
    STA_PREC_F16X3 = 3,  /* 2-term fp16 split of both operands, 3 products (~21-bit, fp32 class)   */
    /* (4 was STA_PREC_F16MX, rounds 1-2: every linear / convolution with its two correction products as ONE block-scaled fp8
     *  MFMA.  +10 % but above the 1e-3 bar on five of the ten stress goldens; retired in round 3, the value is rejected.) */
    STA_PREC_F16X3H = 5, /* f16x3 in the transformer (encoder, decoder, attention, embeddings, pose head); the DPT
                          * head's convolutions in the f16mx arithmetic: fp16 main product + ONE block-scaled fp8 MFMA that
                          * carries both correction products (activation bytes e5m2, weight bytes e4m3: GEMM error ~2e-5, 2 instead of 3 MFMA
                          * units).  The head is
                          * feed-forward and is not followed by any attention layer, so that error is not amplified */
    STA_PREC_F16X3M = 6  /* f16x3h + mlp.fc2 of both transformers in the f16mx arithmetic (mlp.fc1's GELU epilogue writes the
                          * f16mx rows).  Whether this or F16X3H is the default is decided by the written rule of DESIGN.md
                          * section 2 ("a layer class ships in f16mx iff ...") on the measured precision table, not by taste */
};

enum { STA_DTYPE_F32 = 0, STA_DTYPE_F16 = 1, STA_DTYPE_F64 = 2 };   /* weights: F32 only; sta_rope2d_inplace_dtype: all three */

typedef struct sta_config {
    int32_t patch_size;     /* 16 */
    int32_t enc_embed_dim;  /* 1024 */
    int32_t enc_depth;      /* 24 */
    int32_t enc_num_heads;  /* 16  (head_dim must be 64) */
    int32_t dec_embed_dim;  /* 768 */
    int32_t dec_depth;      /* 12  (> 9, heads/dpt_head.py:102) */
    int32_t dec_num_heads;  /* 12 */
    int32_t mlp_ratio;      /* 4 */
    float   rope_base;      /* 100.0 ('RoPE100', sta_model.py:44) */
    float   ln_eps;         /* 1e-6  (sta_model.py:43) */
    int32_t precision;      /* STA_PREC_* */
} sta_config;

/* Fill `cfg` with the reference constructor defaults (sta_model.py:33-52). */
STA_API void sta_default_config(sta_config* cfg);

STA_API int sta_create(const sta_config* cfg, int device, sta_handle** out);
STA_API int sta_destroy(sta_handle* h);

/* Change the arithmetic policy after creation (weights hold both split planes). */
STA_API int sta_set_precision(sta_handle* h, int precision);

/* Bit-reproducible mode.  At SLAM scale (a few hundred rows) the GEMMs and the low-resolution DPT convolutions split K
 * over workgroups.  Since round 2 every product path combines the slices in a FIXED order (each slice stores its partial
 * tile to its own fp32 slab; resid_ln_kernel / qkv_finish_kernel / splitk_finish_kernel sum them), so repeated runs of the
 * same binary give identical bits by default.  One fp32-atomics split-K form is left (in-place residual GEMMs reached
 * outside the fused GEMM + LayerNorm call, e.g. with a forced tile family): on != 0 disables it, at no cost on the product
 * path.  (Why it matters: a borderline `pose_conf < rel_pose_thres` decision, slam.py:169, must not flip run to run.)
 * Default: off. */
STA_API int sta_set_deterministic(sta_handle* h, int on);

/* Streams and concurrency.  Every call enqueues on the caller's stream; the handle keeps ONE scratch context (workspace +
 * split-K buffers) PER STREAM it has been called on, created on the first call on that stream (at most 8 live contexts: a
 * ninth stream takes over the least recently used one behind a device synchronisation).  Calls on
 * different streams therefore never share scratch memory and may overlap on the GPU - the intended use is the SLAM loop's
 * own independence: sta_encode of keyframe i+1 (add_view, slam.py:142-151, 258) on a second stream under
 * sta_regress_views of keyframe i (slam.py:263-277); at 224x224, batch 1 each of them alone leaves most of the chip idle
 * between its ~200 dependent dispatches.  Host-side the handle is still single-threaded (one call at a time).
 * Inside one call the library forks an internal SIDE stream off the caller's stream for the branches that do not lie on the
 * call's critical chain (DPT head: the reassembly of levels 0-2 under the level-3 / refinenet chain; decoder at SLAM scale:
 * the cross-attention K / V under the self-attention) and joins it back before the call's last kernels: the caller sees
 * ordinary stream order.  Same kernels, bit-identical results.
 * (Rounds 2-3 had sta_set_concurrency(h, n): batch slices of ONE forward on library-owned streams.  It stopped paying once
 * the epilogues no longer serialised - -1 % at the headline configuration in round 3 - and was removed in round 4.)
 *
 * sta_set_side_lanes: the application's switch for those internal side streams.  STA_LANES_AUTO (default): on, EXCEPT while
 * the application itself overlaps calls on several streams (another scratch context of this handle was used within its last 8
 * context switches: the chip is then filled across calls and more streams only compete for the runtime's few hardware queues),
 * and except when GPU_MAX_HW_QUEUES is set in the environment (the lanes are tuned for the runtime's default of 4).
 * STA_LANES_OFF / STA_LANES_ON make the schedule independent of either.  Results are bit-identical in all three. */
/* sta_reserve: everything calls of at most these sizes will need on these streams, allocated NOW (no hidden allocation
 * per call).  Without it the library sizes itself lazily - the first call on a new stream creates that stream's scratch context
 * (2 x 16 MiB of split-K scratch; later the side lane's stream, events and 16 MiB), the first call of a larger shape re-allocates
 * the stream's workspace behind a hipDeviceSynchronize, the first scheduler call creates its pinned confidence buffer, a larger
 * patch grid rebuilds the RoPE table.  sta_reserve runs the planning pass of the entry points (the same orchestration code, dry:
 * nothing is launched) and allocates the maximum:
 *   B > 0:          sta_forward_pair[_u8hwc] / sta_encode[_u8hwc] / sta_decode / sta_head_pose / sta_head_pts with batch <= B and
 *                   sta_estimate_intrinsics over <= 2 B maps, frames of H x W;
 *   max_edges > 0:  sta_regress_views[_begin / _finish] with k <= max_edges candidate edges and sta_encode of one H x W frame;
 *   streams[0 .. n_streams): the caller streams the calls will be enqueued on (NULL = the null stream), at most 8 per handle.
 * Afterwards such calls neither allocate nor synchronise the device; sta_alloc_stats proves it: out[0] = device / pinned
 * allocations, frees and stream / event creations, out[1] = device-wide synchronisations the compute entry points have made
 * since sta_create (weight loading, sta_range_report, sta_destroy and the timing tools are not compute entry points).  Not covered
 * (their sizes depend on other arguments): sta_preprocess_frame (tables per source geometry: the first frame of a geometry
 * allocates and synchronises), sta_world_pointcloud (workspace per view count), sta_decode_pos (its RoPE table grows with pos_max and
 * its plan holds the positions table on top of sta_decode's). */
STA_API int sta_reserve(sta_handle* h, int B, int H, int W, int max_edges, void* const* streams, int n_streams);
STA_API int sta_alloc_stats(const sta_handle* h, int64_t out[2]);

/* sta_pipeline_streams: n (<= 4) library-owned non-blocking streams that were MEASURED to overlap pairwise on this device.
 * The runtime maps streams onto a few hardware queues and two streams on one queue serialise - which streams those are is not
 * visible through the HIP API (round 4: the same three application streams were reproducibly 20 % slower or faster) - so the
 * library probes: a kernel that spins ~200 us on one stream, a stamp kernel on the other, overlap iff the second started before
 * the first ended; candidates are kept when they overlap every stream kept so far.  The streams belong to the handle: the FIRST
 * call creates and probes all four (a few ms) whatever n is, every later call returns a prefix of the same list - a stream that
 * was handed out stays valid until sta_destroy -, and a failure while probing leaves nothing behind (the next call starts
 * over).  They are meant for the application's lanes: add_view of keyframe i+1 | edges
 * of keyframe i | heads of keyframe i-1 (vista_slam_amd.keyframe_pipeline).  *n_verified_out (may be NULL): how many of the n
 * are verified mutually concurrent (n unless the runtime has fewer usable queues). */
STA_API int sta_pipeline_streams(sta_handle* h, int n, void** streams_out, int* n_verified_out);

#define STA_LANES_AUTO (-1)
#define STA_LANES_OFF 0
#define STA_LANES_ON 1
STA_API int sta_set_side_lanes(sta_handle* h, int mode);

/* Range report.  Activations travel between kernels as fp16 planes (hi + residual), the f16mx arithmetic of the DPT head adds
 * fp8 correction bytes (activations e5m2, weights e4m3): values beyond +-65504 (or NaN) SATURATE when they are written to a
 * plane, activation correction bytes saturate at +-57344, weight bytes at |w| > 28 (the result then degrades towards
 * single-fp16 accuracy for those elements).  Neither can be seen in the outputs, so the writers of every
 * tensor that is NOT a function of a LayerNorm output count them - the input and hook conversions, every plane of the DPT head
 * (which has no normalisation layers: convolutions, transposed convolutions, bilinear), the split-K finishers - and the
 * LayerNorm kernels count non-finite rows of the residual streams: counts[0] = fp16-range events, counts[1] = fp8
 * saturations of THIS handle's calls since the last reset (events = (lane, tile) pairs with at least one such value; the
 * counters live in the handle since round 4 - two handles on one GPU no longer see each other's events; the call
 * synchronises the device).  QKV / attention / mlp.fc1 outputs
 * are bounded by their LayerNorm inputs and are not counted (0.7 % of the step if they were).  A non-zero counts[0] means the
 * forward left the range the parity goldens cover - the reference (fp32) has no such limit.  reset != 0 clears them.
 * counts[0] also includes, permanently, the number of loaded MFMA-operand weight tensors whose EVERY value is below 2^-12 in
 * magnitude (not all zero): such a tensor would enter the matrix pipe as fp16 subnormals (the LOW end of the range). */
STA_API int sta_range_report(sta_handle* h, unsigned long long counts[2], int reset);

/* Number of state_dict entries the handle expects / has received so far. */
STA_API int sta_num_expected_tensors(const sta_handle* h);
STA_API int sta_num_loaded_tensors(const sta_handle* h);

/* Copy one state_dict entry from HOST memory.  `name` is the reference key
 * (e.g. "enc_blocks.3.attn.qkv.weight"); shape must match exactly; unknown names fail
 * (strict=True semantics).  Aliased keys (scratch.layerK_rn / scratch.layer_rn.{K-1}) and the
 * never-executed tensors (enc_norm.*, refinenet4.resConfUnit1.*) are accepted and dropped. */
STA_API int sta_load_tensor(sta_handle* h, const char* name, const void* host_ptr,
                    const int64_t* shape, int ndim, int dtype);

/* Verify every expected tensor arrived (strict) and build the packed fp16 hi/lo planes. */
STA_API int sta_finalize_weights(sta_handle* h);

/* img_dev [B,3,H,W] -> feat_dev [B, N, enc_dim], N = (H/16)*(W/16); no final norm. */
STA_API int sta_encode(sta_handle* h, const float* img_dev, int B, int H, int W,
               float* feat_dev, void* stream);

/* enc_norm (the encoder's final LayerNorm) on `rows` token rows of enc_dim floats: what
 * _encode_image(normalize=True) adds after the blocks (sta_model.py:172-173).  The forward / SLAM paths call
 * _encode_image(normalize=False) (sta_model.py:259,267; slam.py:144), so nothing on the hot path runs this. */
STA_API int sta_encoder_norm(sta_handle* h, const float* feat_dev, int64_t rows, float* out_dev, void* stream);

/* feat1/feat2 [B, N, enc_dim] (N = hp*wp tokens, hp x wp patch grid).
 * out1/out2: arrays of (dec_depth+1) device pointers, each [B, N+1, dec_dim] or NULL to skip
 * that layer.  Index 0 = decoder input (embed + pose token), index i = output of block i,
 * last index has dec_norm applied (sta_model.py:241-242). */
STA_API int sta_decode(sta_handle* h, const float* feat1, const float* feat2, int B, int hp, int wp,
               float* const* out1, float* const* out2, void* stream);

/* _decode_stereo with CALLER positions (sta_model.py:177-244 hands pos1 / pos2 to every decoder block, whose attentions rotate
 * q / k by them: sta_blocks.py:134-137,196-199): pos1 / pos2 are device int64 [B, N, 2] (y, x) as _encode_image returns them,
 * but need not be the patch grid - a window of a larger grid, a permuted token order, repeated positions.  Values must lie in
 * [-1, pos_max] (out-of-range values are clamped); the RoPE table grows to pos_max on first use.  N = tokens per view (both
 * views: the two sides run as one batch).  Implementation: the QKV epilogues of the call rotate by the identity and one small kernel
 * per Q / K buffer rotates it in place from the positions table - the grid form (sta_decode) stays the fast path, and with the patch
 * grid's own positions the two agree to the rounding of one more fp16-plane split (~1e-7), not bit for bit. */
STA_API int sta_decode_pos(sta_handle* h, const float* feat1, const float* feat2, const int64_t* pos1, const int64_t* pos2,
                   int B, int N, int pos_max, float* const* out1, float* const* out2, void* stream);

/* tok: B rows of dec_dim floats, consecutive rows `tok_stride` floats apart.
 * pose [B,16] row-major 4x4, conf [B]. */
STA_API int sta_head_pose(sta_handle* h, const float* tok, int B, int64_t tok_stride,
                  float* pose, float* conf, void* stream);

/* DPT pointmap head + postprocess.  enc_feat [B,N,enc_dim] (batch stride enc_bstride floats);
 * hookX point at the FIRST PATCH TOKEN (pose token already skipped) of the decoder outputs
 * selected by hooks [0, d/2+1, 3d/4+1, d+1] (dpt_head.py:112); batch strides in floats.
 * pts [B,H,W,3], conf [B,H,W]. */
STA_API int sta_head_pts(sta_handle* h,
                 const float* enc_feat, int64_t enc_bstride,
                 const float* hook1, int64_t hook1_bstride,
                 const float* hook2, int64_t hook2_bstride,
                 const float* hook3, int64_t hook3_bstride,
                 int B, int H, int W, float* pts, float* conf, void* stream);

/* Monolithic two-view forward.  Outputs index 0 = main view (img_a), 1 = support (img_b):
 * pts[k] [B,H,W,3], conf[k] [B,H,W], pose[k] [B,16], pose_conf[k] [B]. */
STA_API int sta_forward_pair(sta_handle* h, const float* img_a, const float* img_b, int B, int H, int W,
                     float* const pts[2], float* const conf[2],
                     float* const pose[2], float* const pose_conf[2], void* stream);

/* Camera-format input (the step before the path): uint8 HWC images [B,H,W,3], 16-byte
 * aligned.  The reference normalisation ImgNorm = ToTensor + Normalize(0.5,0.5)
 * (vista_slam/utils/image.py:13; datasets/slam_images_only.py:19,30) is fused into the patch gather;
 * results are bit-identical to sta_encode / sta_forward_pair on the normalised fp32 NCHW tensor. */
STA_API int sta_encode_u8hwc(sta_handle* h, const uint8_t* img_dev, int B, int H, int W, float* feat_dev, void* stream);
STA_API int sta_forward_pair_u8hwc(sta_handle* h, const uint8_t* img_a, const uint8_t* img_b, int B, int H, int W,
                           float* const pts[2], float* const conf[2],
                           float* const pose[2], float* const pose_conf[2], void* stream);

/* reductions that consume the path's output for every accepted pair, one fused pass.
 * sta_estimate_intrinsics <- estimate_intrinsic_from_pts3d(pts3d, confidence, shared_intrinsic)
 * (vista_slam/utils/slam_utils.py:8-79; slam.py:184) and, in the same read, depths = pts[...,2]
 * (slam.py:185) and conf.mean() per image (pose_graph.py:37).  shared = 0: K_out [B,3,3]; shared = 1: one K [3,3] over
 * all B images; shared = g >= 2: one K per group of g consecutive images, K_out [B/g,3,3] (g = 2: the two views of a
 * pair).  depth_out [B,H,W] and conf_mean_out [B] may be NULL.
 * sta_estimate_scale <- estimate_scale_with_depth_and_confidence(Di, Dj, ci, cj) (slam_utils.py:168-190),
 * s_out is one device float. */
STA_API int sta_estimate_intrinsics(sta_handle* h, const float* pts, const float* conf, int B, int H, int W, int shared,
                            float* K_out, float* depth_out, float* conf_mean_out, void* stream);
STA_API int sta_estimate_scale(sta_handle* h, const float* Di, const float* Dj, const float* ci, const float* cj, int64_t n,
                       float* s_out, void* stream);

/* input step = SLAM_image_only.process_image (vista_slam/datasets/slam_images_only.py:19-33):
 * `_crop_resize_if_necessary_image_only(rgb, resolution, w_edge, h_edge)` (datasets/base/base_view_graph_dataset.py:
 * 171-225: centre crop with an edge margin, cropping.rescale_image_depthmap LANCZOS rescale to cover the resolution
 * - vista_slam/utils/cropping.py:54-81 -, centre crop to the resolution) followed by ImgNorm (ToTensor +
 * Normalize(0.5,0.5)) and ImgGray (ToTensor + Grayscale).  src: one uint8 RGB frame [Hs,Ws,3] on the device.
 * (res_H, res_W) is the configured resolution (landscape or square, like the reference's `resolution`, which asserts
 * resolution[0] >= resolution[1]); the OUTPUT size (out_H, out_W) is the resolution, transposed when the first crop is
 * portrait (crop height > 1.1 x crop width; base_view_graph_dataset.py:200-205) - sta_preprocess_geometry returns it
 * (host only, no device work) so the caller can size the buffers.  A square crop (0.9 < h/w < 1.1) with a non-square
 * resolution is an error: the reference draws the orientation from an rng there.
 * Outputs (device, any may be NULL): u8_out [out_H,out_W,3] uint8 = the PIL image the reference hands to its
 * transforms, bit-exact to Pillow's 8-bit LANCZOS resampler (feeds sta_encode_u8hwc directly); rgb_out
 * [3,out_H,out_W] fp32 = value['rgb']; gray_out [out_H,out_W] fp32 = value['gray'].  The coefficient tables of one
 * geometry are cached in the handle; a geometry change synchronises `stream` once. */
STA_API int sta_preprocess_geometry(int Hs, int Ws, int res_H, int res_W, int w_edge, int h_edge, int* out_H, int* out_W);
STA_API int sta_preprocess_frame(sta_handle* h, const uint8_t* src, int Hs, int Ws, int res_H, int res_W, int w_edge, int h_edge,
                         uint8_t* u8_out, float* rgb_out, float* gray_out, void* stream);

/* output step of OnlineSLAM.save_data_all (vista_slam/slam.py:338-421).
 * sta_world_pointcloud <- slam.py:396-408: local = K^-1 [x,y,1] * depth * scale (compute_local_pointclouds,
 * vista_slam/utils/slam_utils.py:82-121), world = pose * [local,1], keep conf > conf_thres, colour = (img+1)/2;
 * order = torch boolean-mask order (view-major, row-major).  Inputs (device): depths [N,H,W], scales [N], K [N,3,3],
 * poses [N,4,4], confs [N,H,W], imgs [N,3,H,W] in [-1,1] (slam.imgs; may be NULL -> colour 0).  Outputs (device, any may
 * be NULL, each sized for N*H*W points): pts_out [M,3] fp32, col_out [M,3] fp32, ply_records_out [M,27] bytes = the
 * binary_little_endian vertex records (double x,y,z + uchar r,g,b) Open3D writes for pointcloud.ply.
 * *count_host = M on return (the call synchronises `stream`).
 * sta_mat_to_se3 <- pp.mat2SE3(pose) (slam.py:166): [B,4,4] -> [B,7] (tx,ty,tz,qx,qy,qz,qw), qw >= 0. */
STA_API int sta_world_pointcloud(sta_handle* h, const float* depths, const float* scales, const float* K, const float* poses,
                         const float* confs, const float* imgs, int N, int H, int W, float conf_thres,
                         float* pts_out, float* col_out, uint8_t* ply_records_out, int64_t* count_host, void* stream);
STA_API int sta_mat_to_se3(sta_handle* h, const float* poses, int B, float* se3_out, void* stream);

/* keyframe scheduler = OnlineSLAM.regress_two_views (vista_slam/slam.py:153-189) for ALL k candidate
 * edges (i, j_e) of a new keyframe i (the neighbour loop slam.py:263-265 and the loop-closure loop :273-277) in one
 * batched launch sequence instead of k sequential B=1 calls, with the reference's early reject kept:
 *   decode (i, j_e) for e < k  ->  pose head on the ij side (slam.py:165)  ->  one k-float D2H read (the reference
 *   synchronises at the same point, slam.py:169)  ->  edge e is REJECTED iff pose_conf[e] < rel_pose_thres and
 *   !adjacent[e] (adjacent[e] = (i - j_e == 1), slam.py:169)  ->  DPT heads for both views of the accepted edges only
 *   ->  shared-per-pair intrinsics and depths (slam.py:182-185).
 * feat_i [N,1024] device, feat_j[e] [N,1024] device (encoder features cached by add_view, slam.py:142-151).
 * pose [k,16] device (pose_ij, 4x4 row-major, every edge).  pose_conf_host [k], slot_host [k], n_accepted: HOST
 * outputs, valid on return (the call synchronises `stream` once): slot_host[e] = -1 for a rejected edge, else the
 * compact index s of edge e in the per-accepted-edge outputs, all device:
 *   pts [n_acc,2,H,W,3] (view order [ij, ji] = torch.cat order of slam.py:182), conf [n_acc,2,H,W],
 *   K [n_acc,3,3] (shared over the pair's two views), depth [n_acc,2,H,W].
 * The buffers must be sized for k edges.  k <= 16.
 * Portrait frames (H > W): pts / conf / depth stay in image orientation [..,H,W,..]; the reference computes K on the
 * transposed views its head wrapper returns (utils/misc.py:60-61), i.e. with u = row - H/2 against X, v = col - W/2
 * against Y and the principal point (H/2, W/2) - K holds exactly that. */
STA_API int sta_regress_views(sta_handle* h, const float* feat_i, const float* const* feat_j, int k,
                      const uint8_t* adjacent, float rel_pose_thres, int H, int W,
                      float* pose, float* pose_conf_host, int* slot_host, int* n_accepted,
                      float* pts, float* conf, float* K, float* depth, void* stream);

/* The same call in two phases, so that the host can enqueue other work between them (e.g. the decode of the NEXT keyframe's
 * edges on another stream while this keyframe's DPT heads run; the next keyframe's edges need only encoder features):
 *   sta_regress_views_begin : gather + batched decode + pose head on `stream`; the k confidences travel to a pinned host
 *                             buffer behind an event.  No host synchronisation.  `pose` [k,16] device as above.
 *   sta_regress_views_finish: waits for that event only, takes the accept / reject decisions (slam.py:169), and enqueues the
 *                             DPT heads + intrinsics + depths of the accepted edges; host outputs valid on return.
 * The pending call owns its stream's scratch context: between begin and finish no other call of this handle may run on
 * THAT stream (it fails loudly); calls on other streams are fine.  sta_regress_views == begin immediately followed by finish. */
STA_API int sta_regress_views_begin(sta_handle* h, const float* feat_i, const float* const* feat_j, int k, int H, int W,
                            float* pose, void* stream);
STA_API int sta_regress_views_finish(sta_handle* h, const uint8_t* adjacent, float rel_pose_thres,
                             float* pose_conf_host, int* slot_host, int* n_accepted,
                             float* pts, float* conf, float* K, float* depth, void* stream);
/* Give up a call that was begun on `stream` and will not be finished (a host-side error between the phases): waits for
 * phase A's confidence copy, clears the pending state, the stream's scratch context is usable again.  Nothing pending on
 * `stream`: returns 0.  (vista_slam_amd.slam_scheduler.PendingEdges calls it from close() / __del__ / its context manager.) */
STA_API int sta_regress_views_abort(sta_handle* h, void* stream);

/* the compact per-pair record of one step's all-gather (vista_slam_amd/parallel.py; what a SLAM consumer
 * reads of a pair, slam.py:165-185), packed from the outputs of sta_forward_pair* in one launch.  Row b of out_dev
 * (rows `row_stride` floats apart, >= 2 * (17 + 2*H*W)) = for view 0 (main) then view 1 (support):
 * pose[16] | pose_conf | depth = pts[..., 2] [H*W] | conf [H*W].  Inputs as sta_forward_pair wrote them (image orientation). */
STA_API int sta_pack_compact(sta_handle* h, const float* const pts[2], const float* const conf[2], const float* const pose[2],
                     const float* const pose_conf[2], int B, int H, int W, float* out_dev, int64_t row_stride, void* stream);

/* In-place 2-D RoPE on fp32 tokens (B,N,Hh,D) with element strides (stride of D must be 1,
 * stride of Hh must be D; same contract as kernels.cu:91-94); pos int64 [B,N,2] contiguous. */
STA_API int sta_rope2d_inplace(float* tokens_dev, int64_t stride_b, int64_t stride_n,
                       const int64_t* pos_dev, int B, int N, int Hh, int D,
                       float base, float fwd, void* stream);
/* The same for the token dtypes curope dispatches on (AT_DISPATCH_FLOATING_TYPES_AND_HALF, kernels.cu:101): STA_DTYPE_F16,
 * STA_DTYPE_F32, STA_DTYPE_F64.  Like the reference kernel (kernels.cu:33-79: float shared memory, float cos / sin) the
 * rotation is evaluated in fp32 whatever the storage type; strides are in ELEMENTS of that type. */
STA_API int sta_rope2d_inplace_dtype(void* tokens_dev, int dtype, int64_t stride_b, int64_t stride_n,
                             const int64_t* pos_dev, int B, int N, int Hh, int D,
                             float base, float fwd, void* stream);

/* Algorithmic FLOPs of one pair at H x W for this handle's config (SURVEY.md 8d closed form). */
STA_API double sta_flops_per_pair(const sta_handle* h, int H, int W);

/* Bytes currently held by the internal workspace / by packed weights. */
STA_API int64_t sta_workspace_bytes(const sta_handle* h);
STA_API int64_t sta_weight_bytes(const sta_handle* h);

/* Per-stage device timing (hipEvent) of the most recent sta_forward_pair when enabled:
 * ms[0]=encode(both views) ms[1]=decode ms[2]=pose heads ms[3]=dpt heads.  Enabling inserts event
 * records on the stream; reading synchronises on the last event. */
STA_API int sta_enable_stage_timing(sta_handle* h, int on);
STA_API int sta_get_stage_ms(sta_handle* h, float ms[4]);

/* Per-launch hipEvent timing of the dominant kernel (gemm_kernel<.., dense, fp32 epilogue>: the
 * proj / fc2 / embed GEMMs) on the stream it is launched on.  enable!=0 resets the counters and
 * starts recording; sta_kernel_timing_read synchronises on the recorded events and returns the
 * number of launches, the summed kernel time, the summed algorithmic FLOPs (2*M*N*K) and bytes.
 * tile_family selects ONE kernel symbol: 1 = gemm_kernel (128x128), 2..5 = gemm2_kernel 256x256,
 * 256x128, 192x256, 192x128; 0 = all of them. */
STA_API int sta_kernel_timing(sta_handle* h, int enable);
STA_API int sta_kernel_timing_read(sta_handle* h, int tile_family, int* launches, double* total_ms, double* total_flops,
                           double* total_algorithmic_bytes);
/* Effective shader clock (GHz) inside the timed launches of the dominant kernel since sta_kernel_timing(h, 1):
 * s_memtime cycles per 100 MHz s_memrealtime tick, summed over every 64th workgroup.  The chip clocks to its power
 * budget, so the MFMA peak actually available to a kernel is 2.5 PF x clock / 2.4 GHz. */
STA_API int sta_kernel_clock_read(sta_handle* h, float* ghz_out);

/* Every GEMM / convolution launch after sta_kernel_timing(h, 2) as a record (bench.py's survey step and roofline block;
 * tools/): shape6 = {M, N, K, epilogue id, A-loader id (0 dense, 1 conv3x3), 1 if the launch ran in the f16mx arithmetic};
 * variant = tile family (1 = 128x128 register-staged, 2 = 256x256 / 16 waves, 3 = 192x256 / 12 waves, 5 = 192x128 / 8 waves,
 * 6 = 128x64 small-grid ring, 7 = gemm2_pair_kernel: two 192x128 GEMMs in one launch, 8 = halo-tiled 3x3 convolution). */
STA_API int sta_kernel_timing_dump_shapes(sta_handle* h, int cap, int* shape6, float* ms, int* variant, int* n_out);
/* Restrict the per-launch timing to ONE kernel symbol {epilogue id, A-loader id, tile family, f16mx flag}; then
 * sta_kernel_timing(h, 3) times every `every`-th launch of that symbol (bench.py: the dominant kernel inside the timed
 * region.  An event pair costs ~9 us of dispatch - tools/probes/boundary_probe.hip: 11.4 vs 2.7 us per launch -, so the
 * timed region samples one launch in four instead of paying that on all 36 per step). */
STA_API int sta_kernel_timing_filter(sta_handle* h, int epilogue, int a_mode, int family, int mx, int every);

STA_API const char* sta_last_error(void);
STA_API const char* sta_version(void);

#ifdef __cplusplus
}
#endif
#endif /* STA_MI355_H */
