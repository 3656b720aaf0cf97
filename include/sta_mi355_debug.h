/*
 * sta_mi355_debug.h - kernel-level TEST and micro-benchmark entry points.  They are NOT part of the product ABI: the
 * product library libsta_mi355.so exports include/sta_mi355.h only; these symbols exist in libsta_mi355_test.so, the same
 * translation unit compiled with -DSTA_TEST_HOOKS (vista_slam_amd/build.py builds both; tests/ and tools/ load the second).
 *
 * Each sta_debug_* function runs exactly one product kernel (the same template instantiation the product path
 * launches) on fp32 device tensors so that tests/ can compare it with a plain fp32 reference of the
 * same op (reference ops cited per function).  Nothing in the product path calls these.
 * All pointers are device pointers unless noted; return 0 / negative + sta_last_error().
 */
#ifndef STA_MI355_DEBUG_H
#define STA_MI355_DEBUG_H
#include "sta_mi355.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Force the GEMM tile family: 0 = automatic (product behaviour), 1 = 128x128 register-staged kernel,
 * 2 = 256-row direct-to-LDS kernels, 3 = 192-row ones (whenever N % 128 == 0), 4 = 192x128 everywhere, 8 = the halo-tiled
 * 3x3 convolution kernel (conv3h.h) wherever it is legal (stride 1, Cout 128 / 256), 9 = automatic WITHOUT that kernel
 * (same-box A/B), 10 / 11 = automatic with the small-grid family switched off / extended to 4x its threshold (process-wide;
 * tools/tile_table.py only).  Lets the tests cover the families on small shapes. */
STA_API int sta_set_gemm_variant(sta_handle* h, int variant);

/* The tile family launch_gemm's cost model picks for a GEMM / convolution (pure host function, no handle, no GPU:
 * tests/test_tile_table.py replays profiles/r03_tile_table.txt through it).  amode: 0 dense, 1 3x3 convolution; epi: 0 f32,
 * 1 f16 planes, 2 qkv, 3 convT, 4 gelu, 5 f32 in-place residual, 6 fused head; M without the pose-token tail rows; split: 1 for
 * the f16x3 / f16x3h precisions; cstride / Ho / Wo: convolutions only (0 otherwise).  Returns 1, 2, 3, 5, 6 or 8
 * (sta_launch.inc: pick_family). */
STA_API int sta_debug_pick_family(int amode, int epi, long long M, int N, int K, int split, int cstride, int Ho, int Wo);

/* nn.Linear (+GELU/ReLU, +residual): out[M,N] = act(A[M,K] W[N,K]^T + bias) (+resid).
 * act: 0 none, 1 erf-GELU, 2 ReLU.  via_f16 != 0 uses the fp16-plane epilogue (sta_blocks.py:73-79). */
STA_API int sta_debug_gemm(sta_handle* h, const float* A, const float* W, const float* bias, int M, int N, int K,
                   int act, int via_f16, const float* resid, float* out, void* stream);

/* qkv = Linear(x); RoPE2D(q), RoPE2D(k) (sta_blocks.py:132-138, pos_embed.py:169-185).
 * x [S*ntok,K], W [3C,K]; q,k out [S,C/64,ntok,64]; v out is the TRANSPOSED buffer
 * [S*C/64*64, roundup(ntok,64)] exactly as the attention kernel consumes it.
 * has_pose_tok: 0 none, 1 = token 0 of every sequence (reference order), 2 = the decoder's row order: x = [S*ntok patch rows |
 * S pose rows], outputs hold ntok + 1 tokens per sequence with the pose token last (roundup(ntok + 1, 64) columns of V^T). */
STA_API int sta_debug_qkv_rope(sta_handle* h, const float* x, const float* W, const float* bias, int S, int ntok, int K, int C,
                       int wp, int has_pose_tok, float* q, float* k, float* v, void* stream);

/* softmax(q k^T / 8) v, K/V taken from sequence (s+kv_shift)%S (sta_blocks.py:143,201-205).
 * q [S,heads,nq,64], k,v [S,heads,nk,64] -> out [S,nq,heads*64]. */
STA_API int sta_debug_attention(sta_handle* h, const float* q, const float* k, const float* v, int S, int heads,
                        int nq, int nk, int kv_shift, float* out, void* stream);

/* The decoder form of the same kernel: q, k, v [S,heads,n+1,64] with the pose token LAST (as a key it is folded into the
 * initial softmax state, as a query it is served by the pose blocks) -> out [S*n + S, heads*64] in the decoder's row order
 * (patch rows sequence-major, then the S pose rows).  sta_blocks.py:129-148,201-205 on n + 1 tokens. */
STA_API int sta_debug_attention_pose(sta_handle* h, const float* q, const float* k, const float* v, int S, int heads,
                             int n, int kv_shift, float* out, void* stream);

/* Switches of the tests / tools (0 everywhere = product behaviour; see tools/ab_option.py; settable as STA_OPT<idx> in the
 * environment at sta_create only in -DSTA_BENCH_EXPERIMENTS builds).  idx 4 = 1: sta_debug_gemm (plane epilogue) / conv3x3 /
 * convt / up2 run in the DPT head's f16mx arithmetic (f16mx rows in and out, f16mx weights) when the handle's precision is
 * f16x3h - the kernels that precision uses inside the head.  A/B switches of round-4 choices: 1 = 1 small-grid K slices by the
 * old rule; 2 = 1 small-grid GEMMs always on 4 waves; 5 = 1 attention without the 4-stage prefetch schedule; 6 = 1 no side
 * lanes (2: always); 7 = 1 bilinear one output row per workgroup.  Indices 0 and 3 are free. */
STA_API int sta_debug_set_option(sta_handle* h, int idx, int value);

/* Row-tail hint for the dense GEMMs (what the decoder sets to its 2B pose-token rows): the last `rows` (<= 32) rows of the
 * following sta_debug_gemm calls are computed by skinny tail blocks when the shape qualifies.  Sticky; 0 resets. */
STA_API int sta_debug_set_tail_hint(sta_handle* h, int rows);

/* nn.Conv2d 3x3 pad 1 stride 1|2 on NHWC data, weights in the reference [Co,Cin,3,3] layout;
 * optional ReLU on the input, activation on the output, residual add (dpt_block.py:94-142). */
STA_API int sta_debug_conv3x3(sta_handle* h, const float* x, const float* w, const float* bias, int n, int H, int W, int Cin, int Co,
                      int stride, int relu_in, int act, const float* resid, float* out, void* stream);

/* nn.ConvTranspose2d kernel=stride=k on NHWC data, weights [C,C,k,k] (dpt_block.py:369-390). */
STA_API int sta_debug_convt(sta_handle* h, const float* x, const float* w, const float* bias, int n, int H, int W, int C, int k,
                    float* out, void* stream);

/* F.interpolate(scale_factor=2, bilinear, align_corners=True), NHWC, cropped to Hc x Wc. */
STA_API int sta_debug_up2(sta_handle* h, const float* x, int n, int H, int W, int C, int Hc, int Wc, float* out, void* stream);

/* nn.LayerNorm(eps) rows; out32 = direct fp32 output, out_planes = value carried by the fp16 planes. */
STA_API int sta_debug_layernorm(sta_handle* h, const float* x, const float* g, const float* b, int M, int C, float eps,
                        float* out32, float* out_planes, void* stream);

/* head.4 (1x1 128->4) + postprocess (postprocess.py:10-62) on [npix,128] features. */
STA_API int sta_debug_head_final(sta_handle* h, const float* x, const float* w, const float* bias, int64_t npix,
                         float* pts, float* conf, void* stream);

/* PoseHead_small.svd_orthogonalize (pose_head.py:38-57) of B row-major 3x3 matrices. */
STA_API int sta_debug_svd_orthogonalize(sta_handle* h, const float* m, float* r, int B, void* stream);

/* Per-launch record of the timed dominant-kernel family since sta_kernel_timing(h, 1): algorithmic FLOPs, HIP-event
 * duration (ms) and tile family of up to `cap` launches (superseded by sta_kernel_timing_dump_shapes, sta_mi355.h). */
STA_API int sta_kernel_timing_dump(sta_handle* h, int cap, double* flops, float* ms, int* variant, int* n_out);

/* In-kernel stamps of EVERY GEMM / convolution launch of the calls made since sta_kernel_timing(h, 4) (= mode 2 + stamps; the
 * first 512 launches, 2048 workgroups each): per launch out6 = {workgroups, span, median entry -> first K tile, median main loop,
 * median epilogue, spread of the exits} in us; pairs with sta_kernel_timing_dump_shapes (same launch order). */
STA_API int sta_kernel_stamps_dump(sta_handle* h, int cap, double* out6, int* n_out);

/* In-kernel timeline of ONE launch of the product's GEMM for M x N x K (tools/gemm_stamps.py): every workgroup stores four
 * 100 MHz stamps (entry, first K tile landed, main loop done, epilogue acknowledged).  resid != 0: the in-place residual form
 * (at SLAM scale: K slices to slabs).  out[10] (us): workgroups, kernel span (first entry -> last exit), median entry -> first
 * tile, median main loop, median epilogue, spread of the entries, spread of the exits, HIP-event duration of the same launch,
 * K slices, median lifetime of a workgroup.  resid == 2: the specialised in-place residual epilogue of the throughput families.
 * raw_host (may be NULL): the four stamps of the first raw_cap workgroups (block id order; block b runs on XCD b % 8). */
STA_API int sta_bench_gemm_stamps(sta_handle* h, int M, int N, int K, int resid, double* out, unsigned long long* raw_host, int raw_cap, void* stream);

/* Time `iters` back-to-back launches of the dominant GEMM kernel (M x N x K, this handle's
 * precision, random operands) with hipEvents on `stream`; average ms per launch in *ms_out.
 * tile: 0 = product selection, 1 = 128x128, 2 = 256x256, 3 = 256x128.  ablation (tile 2/3 only,
 * bench-only kernel variants): 0 none, 1 no DMA in the K loop, 2 DMA+barriers only, 3 MFMA only. */
STA_API int sta_bench_gemm(sta_handle* h, int M, int N, int K, int iters, int tile, int ablation, float* ms_out, void* stream);
/* Effective shader clock (GHz) observed inside the kernel of the last sta_bench_gemm call (s_memtime cycles per
 * 100 MHz s_memrealtime tick, sampled on every 64th workgroup): the chip clocks to its power budget (DVFS). */
STA_API float sta_bench_gemm_last_ghz(void);
/* The attention kernel alone on random operands (tools): ms per launch over `iters` back-to-back launches.  pose != 0: the
 * decoder form (nq == nk patch tokens + the pose token).  which: reserved for kernel variants under test, pass 0. */
STA_API int sta_bench_attention(sta_handle* h, int S, int heads, int nq, int nk, int pose, int iters, int which, float* ms_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
