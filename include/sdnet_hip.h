/*
 * sdnet_hip.h -- C ABI of libsdnet_hip.so: the MI355X (gfx950) implementation of the SDNet
 * hot path of laclouis5/StructureDetector.
 *
 * The reference has no FFI layer: this path sits behind plain Python classes
 * (SURVEY.md 8b).  Each entry point below therefore cites the reference *Python* interface it
 * replaces (paths relative to the reference repository root); INTEGRATION.md shows the ctypes
 * stub a reference maintainer would add at each of those sites.
 *
 * Conventions
 *   - plain pointers and sizes only, no torch / C++ types;
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); every call is
 *     asynchronous with respect to the host and performs no allocation and no synchronisation;
 *   - scratch memory comes from the caller: ask sd_*_workspace_bytes(), pass a device buffer;
 *   - return value: 0 = ok, <0 = invalid argument (SD_ERR_*), >0 = hipError_t;
 *     sd_last_error() returns a thread-local message for the last non-zero return;
 *   - maps are fp32, rows contiguous (x stride 1, y stride w); batch / channel strides are
 *     explicit (in elements) so that channel-slice views of the head output
 *     (src/sdnet/model/network.py:77-84) are consumed without a copy.
 */
#ifndef SDNET_HIP_H
#define SDNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD_ERR_INVALID   (-1)   /* bad shape / null pointer / unsupported size */
#define SD_ERR_WORKSPACE (-2)   /* workspace too small */
#define SD_ERR_ALIGN     (-3)   /* pointer or stride not 16-byte aligned where required */

#define SD_MAX_TOPK 1024        /* max_objects / max_parts upper bound supported by the select kernel */

typedef void* sd_stream_t;

int         sd_version(void);
const char* sd_last_error(void);
/* Bit mask of timing-only experiment switches compiled into this library (csrc/sd_conv.hip SD_ABLATE_HOT = 1, _STORE = 2,
 * _PATCH = 4: each makes the conv kernels compute WRONG results on purpose).  0 for every product build; the Python loader
 * (structuredetector_amd/_lib.py) refuses a non-zero library.  No reference counterpart (build hygiene). */
int         sd_build_flags(void);

/* ---- tensor primitives: src/sdnet/utils/utils.py ------------------------------------------ */

/* clamped_sigmoid, utils.py:355-361: y = clamp(sigmoid(x), 1e-6, 1-1e-6); n contiguous floats. */
int sd_clamped_sigmoid(const float* x, float* y, int64_t n, sd_stream_t stream);

/* nms, utils.py:441-443: out = (hm == maxpool5x5(hm)) * hm, -inf padding.  in strided
 * (sb, sc), out contiguous (B,C,h,w).  apply_sigmoid != 0 fuses clamped_sigmoid first. */
int sd_nms5(const float* hm, int64_t sb, int64_t sc, float* out, int B, int C, int h, int w,
            int apply_sigmoid, sd_stream_t stream);

/* topk, utils.py:447-467 on an arbitrary (B,C,h,w) score map (no sigmoid, no NMS): global top-k
 * over the class-major flattened map; order: score desc, class asc, flat index asc.
 * Outputs (B,k): score f32, ind i64, cls f32, ys f32, xs f32. */
size_t sd_topk_workspace_bytes(int B, int C, int h, int w, int k);
int sd_topk(const float* scores, int64_t sb, int64_t sc, int B, int C, int h, int w, int k,
            float* out_score, int64_t* out_ind, float* out_cls, float* out_ys, float* out_xs,
            void* workspace, size_t workspace_bytes, sd_stream_t stream);

/* transpose_and_gather, utils.py:347-351: feat (B,C,h*w) strided, ind (B,n) i64 -> out (B,n,C). */
int sd_transpose_and_gather(const float* feat, int64_t sb, int64_t sc, int B, int C, int64_t hw,
                            const int64_t* ind, int n, float* out, sd_stream_t stream);

/* hypot, utils.py:422-437: out[i] = sqrt(fl(fl(x0*x0) + fl(x1*x1))) over n pairs (no FMA). */
int sd_hypot(const float* in_pairs, float* out, int64_t n, sd_stream_t stream);

/* ---- decoder: src/sdnet/data/decoders.py:29-179 ------------------------------------------- */

/* D1-D3 (decoders.py:44-48 / 60-64): clamped sigmoid + 5x5 NMS + top-k of one heatmap group,
 * fused (logits are read once).  Same outputs as sd_topk. */
size_t sd_decode_peaks_workspace_bytes(int B, int C, int h, int w, int k);
int sd_decode_peaks(const float* logits, int64_t sb, int64_t sc, int B, int C, int h, int w, int k,
                    float* out_score, int64_t* out_ind, float* out_cls, float* out_ys, float* out_xs,
                    void* workspace, size_t workspace_bytes, sd_stream_t stream);

/* Whole device stage of Decoder.__call__ (decoders.py:41-100) in two launches (+1 memset node):
 * peaks of the anchor group (M maps, top K) and of the part group (N maps, top P), offset /
 * embedding gather, refinement, masking, K x P association.
 * `packed` receives B*(6K+11P) 4-byte words, structure-of-arrays over the batch:
 *   anchor_out   f32 (B,K,4)  x, y, raw score, label           decoders.py:55-57
 *   part_out     f32 (B,P,6)  x, y, raw score, kind, ox, oy    decoders.py:72-75
 *   part_emb     f32 (B,P,2)                                   decoders.py:66
 *   anchor_smask f32 (B,K)    score, or -1 where score <= conf  decoders.py:84
 *   part_smask   f32 (B,P)                                     decoders.py:79
 *   anchor_ind   i32 (B,K)    flat y*w+x                        decoders.py:46
 *   part_ind     i32 (B,P)
 *   assign       i32 (B,P)    anchor rank, or -1 when min dist >= dist_px  decoders.py:98-100
 *   status       i32 (B)      0 = ok; 1 = sd_decode_fused gave up waiting for a tile block (that image's results are invalid)
 * conf and dist_px are the fp32-rounded thresholds.
 * exact_topk = 1: every slot equals the reference's top-k, including peaks below conf (needed by return_metadata=True);
 * exact_topk = 0: peaks with score <= conf are dropped at compaction -- the assembled annotations are identical (the
 * reference skips / masks those entries, decoders.py:78-86,115-117) and the selection has far fewer candidates to sort;
 * slots past the surviving peaks are then zero-score fillers. */
size_t sd_decode_workspace_bytes(int B, int M, int N, int h, int w, int K, int P);
size_t sd_decode_packed_words(int B, int K, int P);
int sd_decode(const float* anchor_hm, int64_t a_sb, int64_t a_sc,
              const float* part_hm, int64_t p_sb, int64_t p_sc,
              const float* offsets, int64_t o_sb, int64_t o_sc,
              const float* embeddings, int64_t e_sb, int64_t e_sc,
              int B, int M, int N, int h, int w, int K, int P, float conf, float dist_px, int exact_topk,
              void* packed, void* workspace, size_t workspace_bytes, sd_stream_t stream);

/* The same decoder in ONE launch (no second kernel, no memset): the last tile block to arrive for an image runs the selection
 * and association for it.  Results are bit-identical to sd_decode.  max_objects and max_parts up to 512, B up to 256.
 * `state`: sd_decode_state_bytes(B) bytes of device memory owned by the caller and used by nothing else, which must be ZERO
 * before the first call; every call leaves it zero again (so back-to-back calls and hipGraph replays need no memset).  After a
 * failed or aborted launch re-zero it.  `workspace`: sd_decode_fused_workspace_bytes() bytes of scratch (per-tile candidate
 * slots and counts; contents need no initialisation).  Images whose (M+N) x tiles bookkeeping does not fit 64 KB of LDS
 * (e.g. 2048x2048 inputs with 16 maps) are rejected with SD_ERR_INVALID: use sd_decode for those.
 * Replaces decoders.py:41-100 like sd_decode; exists because bs = 1 inference is launch-latency-bound. */
size_t sd_decode_state_bytes(int B, int M, int N, int h, int w);
int    sd_decode_fused_supported(int B, int M, int N, int h, int w, int K, int P);   /* 1 when sd_decode_fused can compute this geometry */
/* 1 where ONE launch is also the FASTER decoder (at most 256 tile blocks per image: 512x512 with 2 + 1 maps has 48; exact top-k: batches
 * up to 8).  Elsewhere call sd_decode: at 1024x1024 / 8 + 8 maps / K = 128 / P = 512 the one-launch form takes 451 us against 116 us.
 * sd_decode_fused REFUSES (SD_ERR_INVALID) image geometries beyond 256 tile blocks unless bit 1 of `exact_topk` (value 2) is set. */
int    sd_decode_fused_recommended(int B, int M, int N, int h, int w, int K, int P, int exact_topk);
size_t sd_decode_fused_workspace_bytes(int B, int M, int N, int h, int w, int K, int P);
int sd_decode_fused(const float* anchor_hm, int64_t a_sb, int64_t a_sc,
                    const float* part_hm, int64_t p_sb, int64_t p_sc,
                    const float* offsets, int64_t o_sb, int64_t o_sc,
                    const float* embeddings, int64_t e_sb, int64_t e_sc,
                    int B, int M, int N, int h, int w, int K, int P, float conf, float dist_px, int exact_topk,
                    void* packed, void* state, size_t state_bytes, void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* Tuning knob of sd_decode_fused (not a result-changing setting; THREAD-LOCAL: it applies to calls made by the host thread that set
 * it, every thread starts from the default, so two engines in two threads of one process cannot disturb each other):
 * "tall_tiles_from" = number of 64x16-pixel tile blocks of a launch (B x (M+N) x tiles) from which the NMS runs on 64x32 tiles instead
 * (default 2688: batches of 56 and more at 128x128 maps; 1 = always, 1 << 30 = never).  The sizes returned by sd_decode_state_bytes /
 * _workspace_bytes cover both.
 * Knobs of sd_decode (same rules): "map_parallel_from" = number of 64x16-pixel tile blocks of a call from which sd_decode takes its
 * map-parallel path -- tile pass without global atomics, one selector block per (image, map), one merge + association block per image;
 * bit-identical results -- instead of the launch pair with one selector block per image (default, value -1: by geometry and mode -- on maps up to 128
 * columns wide sd_decode always takes it, and the one-launch kernel is recommended below 960 tile blocks without the exact top-k only;
 * wider maps from 2560; always for images of 1024+ tile blocks such as
 * 1024x1024 inputs with 8 + 8 maps; sd_decode_fused_recommended follows the same rule; a value >= 0 holds for every geometry: 1 = always,
 * 1 << 30 = never); "map_tile_height" = 16 / 32 / 0 (by size) rows per NMS tile there;
 * "map_scalar_nms" = 1 keeps the per-pixel-sigmoid tile kernel where the logit-domain one (w % 4 == 0, aligned planes) applies;
 * "map_rows11" = bands of 128-row maps: 1 (default) 8-row bands with two bands per wave on launches of < 1024 maps, else 11-row bands; 0 = 16-row bands;
 * 8 / 11 = forced. */
int sd_decode_set_option(const char* name, int value);

/* Device self-check of the two properties the logit-domain NMS tile pass of sd_decode rests on, over ALL 2^32 fp32 bit patterns:
 * out3[0] = violations of "clamped sigmoid is monotone non-decreasing", out3[1] = violations of the near-tie margin table
 * (a logit further below the window maximum than the margin has a strictly smaller sigmoid), out3[2] = values visited
 * (2^32 - NaNs = 4 278 190 082).  out3: three 64-bit words of device memory.  ~10 ms. */
int sd_selfcheck_sigmoid(unsigned long long* out3, sd_stream_t stream);

/* Explicit host wait for everything queued on `stream` (hipStreamSynchronize): the ONE blocking call of the decoder's host side,
 * after which a `packed` buffer that lives in pinned, device-mapped host memory may be read (decoders.py:103-139 reads its
 * tensors through ~200 implicit synchronisations instead).  No other function of this library blocks. */
int sd_stream_synchronize(sd_stream_t stream);

/* D4-D5 alone (decoders.py:49-100) from already selected peaks (outputs of sd_decode_peaks):
 * same `packed` layout as sd_decode. */
int sd_decode_group(const float* a_score, const int64_t* a_ind, const float* a_cls,
                    const float* p_score, const int64_t* p_ind, const float* p_cls,
                    const float* offsets, int64_t o_sb, int64_t o_sc,
                    const float* embeddings, int64_t e_sb, int64_t e_sc,
                    int B, int h, int w, int K, int P, float conf, float dist_px,
                    void* packed, sd_stream_t stream);

/* ---- target rendering: src/sdnet/data/transforms.py:130-205 (Encode) ---------------------- */

/* Heatmaps of a batch (transforms.py:143,160-161,173-174; utils.py:418-419): for every pixel of
 * channel c of image b, exp(fp32(-min d^2) / two_sigma2) over that channel's keypoint centres,
 * 0 when the channel has none.  Centres are the truncated output-pixel coordinates
 * (cx[i], cy[i]) sorted by (image, channel); chan_ptr (B*C+1) is the CSR row pointer.
 * out: (B,C,h,w) contiguous fp32. */
int sd_render_targets(const int32_t* cx, const int32_t* cy, const int32_t* chan_ptr,
                      int B, int C, int h, int w, float two_sigma2, float* out, sd_stream_t stream);

/* ---- input pipeline: src/sdnet/data/transforms.py:9-35,47-60,108-118 (flips, Resize, Normalize) ------------------------ */

/* Batched Resize (PIL bilinear, bit-identical bytes: Pillow's 22-bit fixed-point separable resampling, horizontal pass first)
 * + optional horizontal / vertical flip of the RESIZED image (transforms.py:217-226 order) + to_tensor + Normalize.
 * images: (B, Hin, Win, 3) u8 on the device; out: (B, 3, Hout, Wout) fp32 NCHW.  h_bounds / h_kk (Wout x {first source
 * column, count} and Wout x h_ksize fixed-point weights) and v_bounds / v_kk (per output row) are the coefficient tables of
 * Pillow's precompute_coeffs + normalize_coeffs_8bpc, computed by the host (structuredetector_amd/data/augment.py) and
 * resident on the device; flips: B bytes (bit 0 = horizontal, bit 1 = vertical) or NULL; mean3 / std3: HOST pointers to
 * three floats.  workspace: sd_preprocess_workspace_bytes() bytes (the 8-bit intermediate of the horizontal pass). */
size_t sd_preprocess_workspace_bytes(int B, int Hin, int Win, int Wout);
int sd_preprocess_images(const uint8_t* images, int B, int Hin, int Win, int Hout, int Wout,
                         const int* h_bounds, const int* h_kk, int h_ksize, const int* v_bounds, const int* v_kk, int v_ksize,
                         const uint8_t* flips, const float* mean3, const float* std3, float* out,
                         void* workspace, size_t workspace_bytes, sd_stream_t stream);

/* The same with the reference's RandomColorJitter (src/sdnet/data/transforms.py:37-47: torchvision ColorJitter(0.25, 0.25, 0.15, 0.05) on the
 * RESIZED PIL image, before the flips and Normalize) applied per image on the device, byte for byte what Pillow computes
 * (ImageEnhance.Brightness / Contrast / Color and the HSV round trip of adjust_hue).  jitter_order (B) int32: bits 0-7 = the four op ids in
 * application order, 2 bits each (0 brightness, 1 contrast, 2 saturation, 3 hue), bits 8-15 = the hue shift byte uint8(hue_factor * 255);
 * jitter_factors (B, 3) fp32 = brightness, contrast, saturation factors.  The random draws stay with the caller. */
size_t sd_preprocess_jitter_workspace_bytes(int B, int Hin, int Win, int Hout, int Wout);
int sd_preprocess_images_jitter(const uint8_t* images, int B, int Hin, int Win, int Hout, int Wout, const int* h_bounds, const int* h_kk,
                                int h_ksize, const int* v_bounds, const int* v_kk, int v_ksize, const uint8_t* flips, const int* jitter_order,
                                const float* jitter_factors, const float* mean3, const float* std3, float* out, void* workspace,
                                size_t workspace_bytes, sd_stream_t stream);

/* ---- loss: src/sdnet/model/loss.py:17-64,91-117 ------------------------------------------- */

#define SD_HM_MSE   0
#define SD_HM_FOCAL 1

typedef struct sd_loss_desc {
    /* head output, network.py:77-84 (strided channel-slice views allowed) */
    const float* anchor_hm;  int64_t a_sb, a_sc;      /* (B,M,h,w) logits */
    const float* part_hm;    int64_t p_sb, p_sc;      /* (B,N,h,w) logits */
    const float* offsets;    int64_t o_sb, o_sc;      /* (B,2,h,w) */
    const float* embeddings; int64_t e_sb, e_sc;      /* (B,2,h,w) */
    /* collated Encode output, dataset.py:58-87 */
    const float* t_anchor_hm; int64_t ta_sb, ta_sc;   /* (B,M,h,w) */
    const float* t_part_hm;   int64_t tp_sb, tp_sc;   /* (B,N,h,w) */
    const int64_t* anchor_inds;   /* (B,K) */
    const int64_t* part_inds;     /* (B,P) */
    const float* anchor_offsets;  /* (B,K,2) */
    const float* part_offsets;    /* (B,P,2) */
    const float* t_embeddings;    /* (B,P,2) */
    const uint8_t* anchor_mask;   /* (B,K) bool */
    const uint8_t* part_mask;     /* (B,P) bool */
    int B, M, N, h, w, K, P;
    int hm_loss_fn;               /* SD_HM_MSE | SD_HM_FOCAL (args.py:96-102) */
    float hm_weight, offset_weight, embedding_weight;   /* args.py:118-132 */
} sd_loss_desc;

/* Loss.forward (loss.py:17-50).  out[0..3] = total, hm, offset, embedding (weighted, as
 * LossStats loss.py:120-165); out[4..7] = num_pos(anchor), num_pos(part), #valid anchors,
 * #valid parts (kept on device for the backward; no host branch, cf. loss.py:59-61,110). */
size_t sd_loss_workspace_bytes(int B, int M, int N, int h, int w);
int sd_loss_fwd(const sd_loss_desc* d, float* out8, void* workspace, size_t workspace_bytes, sd_stream_t stream);

/* d total / d head for all M+N+4 channels.  grad_out: device scalar (upstream gradient).
 * dhead: (B,M+N+4,h,w) contiguous, fully overwritten. */
int sd_loss_bwd(const sd_loss_desc* d, const float* out8, const float* grad_out, float* dhead, sd_stream_t stream);


/* ---- network: src/sdnet/model/network.py:6-87 (+ torchvision resnet34 BasicBlocks) --------
 * Activations are NHWC fp32 (torch channels_last), weights [Cout][R][S][Cin] (torch
 * channels_last OIHW).  These entry points replace the torch.nn.Conv2d / BatchNorm2d / ReLU /
 * MaxPool2d / Upsample modules the reference composes, and torch.optim.Adam
 * (src/sdnet/model/trainer.py:53,124). */

typedef struct sd_conv_desc {
    int B, Hi, Wi, Cin, Ho, Wo, Cout, R, S, stride, pad;
} sd_conv_desc;

/* y = [relu]( conv(x, w) * scale[co] + shift[co] [+ residual] ), fp32 MFMA implicit GEMM.
 * scale/shift/residual nullable.  res_up2: residual is the (Ho/2, Wo/2) map, nearest-upsampled x2
 * (Fpn.forward, network.py:18-19).  Needs Cin % 32 == 0, Cout % 64 == 0.  When the tile grid cannot fill
 * the chip (small batch) and a workspace is supplied, K is split over blocks and a second pass applies the epilogue. */
size_t sd_conv2d_fwd_workspace_bytes(const sd_conv_desc* d);   /* > 0 only when split-K pays (small batch) */
int sd_conv2d_fwd(const float* x_nhwc, const float* w_krsc, float* y_nhwc, const sd_conv_desc* d,
                  const float* scale, const float* shift, const float* residual, int res_up2, int relu,
                  void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* stem: 7x7/2 conv 3 -> 64 reading the NCHW image directly (network.py:43; resnet.conv1). */
size_t sd_conv2d_stem_fwd_workspace_bytes(const sd_conv_desc* d);
int sd_conv2d_stem_fwd(const float* x_nchw, const float* w_krsc, void* y_nhwc, const sd_conv_desc* d,
                       const float* scale, const float* shift, int relu, int out_bf16, void* workspace,
                       size_t workspace_bytes, sd_stream_t stream);

/* Small-batch inference forward (the batch-1 evaluate loop, src/sdnet/cli/evaluate.py:34-45 -> network.py:59-84): same
 * arithmetic and epilogue as sd_conv2d_fwd / sd_conv2d_fwd_bf16 (bf16 != 0: x, w, y, residual are bf16, accumulation and epilogue
 * fp32) on 64-pixel x 64-channel tiles with the reduction split over blocks so that tiles x slices is about one block per CU; the
 * partial tiles are combined INSIDE the launch by the block that arrives last for a tile (summed in slice order: results do not
 * depend on the arrival order), so a conv is ONE launch.  sd_conv2d_fwd_sb_supported(): 1 for the geometries it is meant for
 * (the 128-row tile grid of sd_conv2d_fwd would not fill the chip twice: batches up to ~16 at 512x512); any R == S conv with Cin % 32 (bf16: 64) == 0, Cout % 64 == 0
 * is computed correctly.  `workspace`: sd_conv2d_fwd_sb_workspace_bytes() of scratch (partial tiles; no initialisation).
 * `state`: sd_conv2d_fwd_sb_state_bytes() bytes owned by the caller, used by nothing else that may run concurrently, ZERO before
 * the first call; every call leaves it zero again (back-to-back launches and hipGraph replays need no memset).  After a failed
 * or aborted launch re-zero it.  Both may be null / 0 when sd_conv2d_fwd_sb_workspace_bytes() is 0 (no split). */
int    sd_conv2d_fwd_sb_supported(const sd_conv_desc* d, int bf16);
size_t sd_conv2d_fwd_sb_workspace_bytes(const sd_conv_desc* d, int bf16);
size_t sd_conv2d_fwd_sb_state_bytes(const sd_conv_desc* d, int bf16);
int sd_conv2d_fwd_sb(const void* x_nhwc, const void* w_krsc, void* y_nhwc, const sd_conv_desc* d, const float* scale, const float* shift,
                     const void* residual, int res_up2, int relu, int bf16, void* workspace, size_t workspace_bytes, void* state,
                     size_t state_bytes, sd_stream_t stream);

/* The stem conv (7x7 / 2 / 3, NCHW image -> NHWC fp32) with the batch statistics of its output from the same launch
 * (as sd_conv2d_fwd_bn_stats; one pass over the 16.8 MB/img stem output less). */
size_t sd_conv2d_stem_fwd_bn_stats_workspace_bytes(const sd_conv_desc* d);
int sd_conv2d_stem_fwd_bn_stats(const float* x_nchw, const float* w, float* y, const sd_conv_desc* d, float eps, float momentum,
                                float* running_mean, float* running_var, float* mean, float* invstd, void* workspace,
                                size_t workspace_bytes, sd_stream_t stream);

/* Training forward of a conv that feeds a BatchNorm2d: y = conv(x, w) and, from the accumulators of the same launch, the batch
 * statistics of y (mean, invstd = 1/sqrt(biased var + eps), running-stat update as sd_bn_train_stats) -- one pass over y less
 * than sd_conv2d_fwd + sd_bn_train_stats.  Partial sums per (tile, wave row) go through `workspace` and are finished in double
 * precision in a fixed order (deterministic).  Falls back to the two-pass form when the forward splits K (tiny batches). */
size_t sd_conv2d_fwd_bn_stats_workspace_bytes(const sd_conv_desc* d);
int sd_conv2d_fwd_bn_stats(const float* x, const float* w, float* y, const sd_conv_desc* d, float eps, float momentum,
                           float* running_mean, float* running_var, float* mean, float* invstd, void* workspace,
                           size_t workspace_bytes, sd_stream_t stream);
/* second half of sd_bn_train_stats on caller-provided partial sums [rows][2][C] (sum, sum of squares per channel); `scratch`
 * (nullable) = sd_bn_finalize_scratch_rows(rows) * 2 * C floats for the coalesced first folding level used with many rows */
int sd_bn_finalize_scratch_rows(int rows);
int sd_bn_finalize_stats(const float* partial, int rows, int64_t M, int C, float eps, float momentum, float* running_mean,
                         float* running_var, float* mean, float* invstd, float* scratch, sd_stream_t stream);

/* Data-gradient of a conv whose INPUT was the output of a BatchNorm2d (+ReLU): dx = dgrad(dy) [+ residual] as sd_conv2d_dgrad,
 * and from the values of the same epilogue the reduction of that BatchNorm's backward (sum g, sum g * xhat per channel with
 * g = dx * relu mask, xhat = (bn_x - mean) * invstd; relu as in sd_bn_bwd) -- one pass over dx and bn_x less than
 * sd_conv2d_dgrad + sd_bn_bwd.  Writes dgamma / dbeta (+= when accumulate) and means_out = [mean(g) (C), mean(g xhat) (C)],
 * C = Cin; finish with sd_bn_bwd_apply(dy = dx, x = bn_x, ...).  Deterministic (partials per tile, fixed-order finish). */
size_t sd_conv2d_dgrad_bn_reduce_workspace_bytes(const sd_conv_desc* d);
int sd_conv2d_dgrad_bn_reduce(const float* dy, const float* w_t, float* dx, const sd_conv_desc* d, const float* residual,
                              const float* bn_x, const float* bn_y, int relu, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, float* dgamma, float* dbeta, int accumulate,
                              float* means_out, void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* the two halves of sd_bn_bwd after its reduction pass */
int sd_bn_bwd_finalize(const float* partial, int rows, int64_t M, int C, float* dgamma, float* dbeta, int accumulate,
                       float* means_out, float* scratch, sd_stream_t stream);
int sd_bn_bwd_apply(const float* dy, const float* x, const float* y, int relu, int64_t M, int C, const float* mean,
                    const float* invstd, const float* gamma, const float* beta, const float* means, float* dx, float* g_out,
                    sd_stream_t stream);

/* Kernel-selection thresholds (tuning / test switches; they never change results beyond fp32 summation order).  THREAD-LOCAL: a value
 * applies to the calls made by the host thread that set it and every thread starts from the defaults -- two engines driven from two
 * threads of one process (training + evaluation) cannot change each other's kernel choice; the library keeps no process-global
 * mutable state.  sd_conv2d_kernel_name() reports the choice the CALLING thread would get.  "conv_patch_min_tiles": smallest tile grid for which the 3x3 stride-1 convs take the
 * patch-staging kernel (default 512 = two resident blocks per CU; 1 = always, for tests; 1 << 30 = never);
 * "conv_patch_bn64": 1 = also for layers with 64 output channels (default 0: slower inside the training step);
 * "conv_pp_min_tiles": smallest grid of 512-pixel x 128-channel tiles for which the bf16 3x3 stride-1 convs take the two-group
 * kernel k_conv3x3_bf16_pp (default 200; 1 = always, for tests; 1 << 30 = never); "conv_patch_narrow": layers whose 128-channel patch
 * tiles do not fill the chip fall back to 64-channel patch tiles (0 never, 1 fp32 only, 2 = default: fp32 and bf16); "conv_pp_strips": 0 = maps of 128 pixels and wider are
 * not cut into 64-pixel column strips for that kernel (default 1; A/B measurements); "conv_fwd_split_k": 0 = the forward convs never
 * split K over blocks (default 1: small grids do), so that tests can put small problems on the single-pass kernels;
 * "conv_rows64_min_units": smallest number of (image, 128-pixel strip, row range) units for which the bf16 64 -> 64 channel 3x3 convs take
 * the row-stream kernel k_conv3x3_c64_rows_bf16 (default 192, and at least 16 rows per unit; 1 = always, for tests; 1 << 30 = never);
 * "conv_rows_f32_min_units": the same for the fp32 row-stream kernel k_conv3x3_c64_rows_f32 (units = image x 64-pixel strip x row range).
 * "conv1x1_stream_min_pixels": smallest number of output pixels for which a bf16 1x1 / stride 1 conv onto 128 channels (Cin 64 or 128, no
 *   scale, no statistics: the FPN laterals and the 128 -> 128 1x1 data-gradient) takes the stream kernel k_conv1x1_stream_bf16 (default
 *   65536; 32 = always, for tests; 1 << 30 = never).
 * "wgrad_f32_ring": form of the fp32 weight gradient of the 3x3 / stride 1 layers with maps a multiple of 32 pixels wide: 2 (default) =
 *   k_wgrad3x3_ring2 (row ring: one new patch row per chunk; two groups of four waves per 512-thread block share a tile and store ONE partial),
 *   1 = k_wgrad3x3_ring (one group per block), 0 = the first form k_wgrad3x3<32> (A/B, tests).
 * "wgrad_bf16_ring": form of the bf16 weight gradient of the 3x3 / stride 1 layers with maps a multiple of 32 pixels wide: 5 (default) =
 *   k_wgrad3x3_bf16_ring2 (row ring, two groups of four waves half a chunk apart in one 512-thread block), 2 .. 4 = k_wgrad3x3_bf16_ring with
 *   that prefetch distance, 0 = the first form k_wgrad3x3_bf16<32> (A/B, tests).
 * "stem_fwd_blocks": persistent blocks of the fp32 training stem forward (default 512 = two per CU; 256 = one per CU, the setting the
 *   in-kernel phase trace of tools/stem_trace_f32.py compares against). */
int sd_set_option(const char* name, int value);

/* Name of the device kernel the launchers pick for this geometry (pass 0 = sd_conv2d_fwd, 1 = sd_conv2d_dgrad,
 * 2 = sd_conv2d_wgrad; 16 = sd_conv2d_fwd_bf16, 17 = sd_conv2d_dgrad_bf16), as rocprofv3 prints it without the sd:: namespace -- lets a profiler label its event timings
 * with the same names as the kernel trace.  Thread-local storage, valid until the next call on the thread. */
const char* sd_conv2d_kernel_name(const sd_conv_desc* d, int pass);

/* bf16 backbone (inference; BASELINE stress config "bf16 backbone + fp32 decode"): activations and weights bf16
 * NHWC / [Cout][R][S][Cin], v_mfma_f32_32x32x16_bf16 with fp32 accumulation and fp32 epilogue (folded BN, bias,
 * residual, ReLU), bf16 stores.  Needs Cin % 64 == 0. */
int sd_cast_f32_to_bf16(const float* x, void* y_bf16, int64_t n, sd_stream_t stream);
size_t sd_conv2d_fwd_bf16_workspace_bytes(const sd_conv_desc* d);
int sd_conv2d_fwd_bf16(const void* x_nhwc_bf16, const void* w_krsc_bf16, void* y_nhwc_bf16, const sd_conv_desc* d,
                       const float* scale, const float* shift, const void* residual_bf16, int res_up2, int relu,
                       void* workspace, size_t workspace_bytes, sd_stream_t stream);
int sd_maxpool3x3s2_fwd_bf16(const void* x_bf16, void* y_bf16, int B, int Hi, int Wi, int C, sd_stream_t stream);
/* network.py:59-63 `adpater` (conv1 7x7/s2 -> bn1 -> relu -> maxpool 3x3/s2) in ONE launch for the bf16 backbone: fp32 NCHW image, fp32
 * weights [64][7][7][3], folded BatchNorm scale / shift, pooled NHWC bf16 output [B][Ho/2][Wo/2][64] (d describes the conv: Ho, Wo even).
 * The full-resolution activation is never written.  Same values as sd_conv2d_stem_fwd(out_bf16 = 1) + sd_maxpool3x3s2_fwd_bf16 up to
 * the fp32 summation order of the conv (<= 1 bf16 ulp). */
int sd_stem_bn_relu_maxpool_fwd_bf16(const float* x_nchw, const float* w, const float* scale, const float* shift, void* y_bf16,
                                     const sd_conv_desc* d, sd_stream_t stream);

/* ---- mixed-precision training: the step the reference runs under `--amp` (src/sdnet/model/trainer.py:115-121: the forward and
 * the loss inside torch.autocast, backward in the dtypes the forward used, fp32 master weights updated by Adam).  Activations and
 * the weights handed to the convs are bf16, every accumulation (MFMA, BatchNorm statistics, gradients of parameters) is fp32.
 * BatchNorm statistics are those of the bf16-ROUNDED conv output (the tensor that is normalised, as under autocast). */
int sd_cast_bf16_to_f32(const void* x_bf16, float* y, int64_t n, sd_stream_t stream);
size_t sd_conv2d_fwd_bf16_bn_stats_workspace_bytes(const sd_conv_desc* d);
int sd_conv2d_fwd_bf16_bn_stats(const void* x_nhwc_bf16, const void* w_krsc_bf16, void* y_nhwc_bf16, const sd_conv_desc* d, float eps,
                                float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                                void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* sd_conv2d_stem_fwd_bn_stats with the product on the bf16 MFMA (operands rounded on the way in; fp32 output and statistics). */
int sd_conv2d_stem_fwd_bn_stats_bf16mm(const float* x_nchw, const float* w, float* y, const sd_conv_desc* d, float eps, float momentum,
                                       float* running_mean, float* running_var, float* mean, float* invstd,
                                       void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* ... and with a bf16 NHWC output (statistics of the rounded values): the mixed-precision step's stem; its tail reads that tensor through
 * sd_bn_relu_maxpool_fwd_bf16 / sd_maxpool_bn_relu_bwd_bf16 (fp32 arithmetic, bf16 conv output / pooled map / pooled gradient, fp32 dx). */
int sd_conv2d_stem_fwd_bn_stats_bf16(const float* x_nchw, const float* w, void* y_bf16, const sd_conv_desc* d, float eps, float momentum,
                                     float* running_mean, float* running_var, float* mean, float* invstd, void* workspace,
                                     size_t workspace_bytes, sd_stream_t stream);
int sd_bn_relu_maxpool_fwd_bf16(const void* x_bf16, int B, int Hi, int Wi, int C, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, void* y_pool_bf16, uint8_t* idx, sd_stream_t stream);
int sd_maxpool_bn_relu_bwd_bf16(const void* dpool_bf16, const uint8_t* idx, const void* x_bf16, int B, int Hi, int Wi, int C, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta, int accumulate,
                                void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* ... and with the input gradient stored as bf16 (what autocast hands the stem conv's backward: trainer.py:116-121 runs conv1 in bf16):
 * half the bytes written here and read by sd_conv2d_stem_wgrad_bf16.  dgamma / dbeta as above (fp32 sums of the fp32 values). */
int sd_maxpool_bn_relu_bwd_bf16_dx16(const void* dpool_bf16, const uint8_t* idx, const void* x_bf16, int B, int Hi, int Wi, int C, const float* mean,
                                     const float* invstd, const float* gamma, const float* beta, void* dx_bf16, float* dgamma, float* dbeta,
                                     int accumulate, void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* [Cout][taps][Cin] fp32 -> [Cin][taps][Cout] bf16 in one pass (the data-gradient's weights under --amp). */
int sd_conv2d_transpose_weights_bf16(const float* w, void* w_t_bf16, int Cout, int taps, int Cin, sd_stream_t stream);

/* Inference: the last FPN convolution (network.py:17-18: conv3x3 + folded BatchNorm + ReLU onto fpn_depth = 128 channels) with the network's
 * 1x1 head (network.py:22-29) applied to every output tile in the kernel's epilogue -- the FPN output tensor (268 MB in bf16 at bs = 64,
 * 512x512) is neither written nor read back, and the head launch disappears.  head_y = fp32 NCHW (B, head_co, Ho, Wo), the tensor
 * Network.forward slices into its four views (network.py:77-84).  `head_prepared` = sd_head_split_bf16_bytes() bytes written by
 * sd_head_split_bf16 from the fp32 head weights [head_co][128] and bias (hi / lo bf16 halves, zero padded to 32 rows: once per set of
 * weights).  sd_conv2d_fwd_bf16_head_supported = 1 where the convolution takes k_conv3x3_bf16_pp (bs = 64 at 512x512, bs = 16 at
 * 1024x1024, ...); elsewhere call sd_conv2d_fwd_bf16 + sd_head_fwd_bf16. */
int sd_conv2d_fwd_bf16_head_supported(const sd_conv_desc* d, int head_co);
size_t sd_head_split_bf16_bytes(void);
int sd_head_split_bf16(const float* head_w, const float* head_bias, int head_co, void* prepared, sd_stream_t stream);
int sd_conv2d_fwd_bf16_head(const void* x_bf16, const void* w_bf16, const sd_conv_desc* d, const float* scale, const float* shift, int relu,
                            const void* head_prepared, int head_co, float* head_y, sd_stream_t stream);
/* dx = dgrad(dy) [+ residual]: bf16 dy / transposed weights [Cin][R][S][Cout] / dx; res_mode 0 none, 1 bf16 tensor of dx's shape,
 * 2 bf16 half-size map added at the even pixels (the 1x1 / stride-2 downsample branch). */
int sd_conv2d_dgrad_bf16(const void* dy_bf16, const void* w_t_bf16, void* dx_bf16, const sd_conv_desc* d, const void* residual_bf16,
                         int res_mode, sd_stream_t stream);
/* dW (fp32, into the flat gradient buffer) from bf16 dy and bf16 x: 3x3 / stride 1 / pad 1 layers on the bf16 MFMA (transposed LDS
 * reads, all nine taps per block), the strided and 1x1 convs through the fp32 kernels on widened operands. */
size_t sd_conv2d_wgrad_bf16_workspace_bytes(const sd_conv_desc* d);
int sd_conv2d_wgrad_bf16(const void* dy_bf16, const void* x_nhwc_bf16, float* dw_krsc, const sd_conv_desc* d, int accumulate,
                         void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* sd_bn_apply / sd_bn_bwd / sd_col_sum / sd_upsample2x_bwd on bf16 activations (parameters, statistics and parameter gradients fp32;
 * arithmetic in fp32, one rounding at the store). */
int sd_bn_apply_bf16(const void* x, void* y, int64_t M, int C, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, const void* residual, int relu, uint8_t* relu_mask_out, sd_stream_t stream);
int sd_bn_bwd_bf16(const void* dy, const void* x, const void* y, int relu, int64_t M, int C, const float* mean, const float* invstd,
                   const float* gamma, const float* beta, void* dx, void* g_out, float* dgamma, float* dbeta, int accumulate,
                   void* workspace, size_t workspace_bytes, sd_stream_t stream);
int sd_bn_train_stats_bf16(const void* x, int64_t M, int C, float eps, float momentum, float* running_mean, float* running_var,
                           float* mean, float* invstd, void* workspace, size_t workspace_bytes, sd_stream_t stream);
int sd_col_sum_bf16(const void* x, int64_t M, int C, float* out, int accumulate, void* workspace, size_t workspace_bytes, sd_stream_t stream);
int sd_upsample2x_bwd_bf16(const void* dy, const void* add, void* dx, int B, int H, int W, int C, sd_stream_t stream);
int sd_head_fwd_bf16(const void* x_nhwc_bf16, const float* w, const float* bias, float* y_nchw, int B, int HW, int C,
                     int Co, sd_stream_t stream);
/* dX = conv_transpose(dY, W): same kernel with the inverted coordinate map; w_t = weights
 * re-laid as [Cin][R][S][Cout] by sd_conv2d_transpose_weights.  residual (nullable, same layout as
 * dX) is added (skip-connection gradient). */
int sd_conv2d_dgrad(const float* dy_nhwc, const float* w_t, float* dx_nhwc, const sd_conv_desc* d,
                    const float* residual, sd_stream_t stream);
/* The same with a residual that only exists on the even pixels: residual_half is (B, Hi/2, Wi/2, Cin) and is added to
 * dx[b, 2y, 2x, :] -- the data-gradient of a 1x1 / stride 2 "downsample" branch (network: BasicBlock.downsample) is zero on
 * three of four pixels, so it is computed as the stride-1 1x1 data-gradient on the small map and joined here instead of being
 * zero-filled to full size and read back. */
int sd_conv2d_dgrad_half_res(const float* dy, const float* w_t, float* dx, const sd_conv_desc* d, const float* residual_half,
                             sd_stream_t stream);
int sd_conv2d_transpose_weights(const float* w, float* w_t, int Cout, int taps, int Cin, sd_stream_t stream);
/* The same transpose for MANY convs in one launch (a training step re-lays every conv's weights once per backward).  table (device,
 * 8 ints per conv): {offset of w in w_base, offset of w_t in w_t_base (elements), Cout, taps, Cin, first block, ceil(Cin/32),
 * ceil(Cout/32)}; a conv owns ceil(Cin/32) * ceil(Cout/32) * taps consecutive blocks; out_bf16: w_t_base is bf16 (mixed precision). */
int sd_conv2d_transpose_weights_batched(const float* w_base, void* w_t_base, const int* table, int nconv, int total_blocks, int out_bf16,
                                        sd_stream_t stream);
/* dW[co][r][s][ci] (+)= sum_pixels dY * X, split over pixel ranges + deterministic reduce. */
size_t sd_conv2d_wgrad_workspace_bytes(const sd_conv_desc* d);
int sd_conv2d_wgrad(const float* dy_nhwc, const float* x_nhwc, float* dw_krsc, const sd_conv_desc* d,
                    int accumulate, void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* stem weight gradient from the NCHW image: dW [64][7][7][3]. */
size_t sd_conv2d_stem_wgrad_workspace_bytes(const sd_conv_desc* d);
int sd_conv2d_stem_wgrad(const float* dy_nhwc, const float* x_nchw, float* dw_krsc, const sd_conv_desc* d,
                         int accumulate, void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* ... with the product on the bf16 MFMA (mixed-precision training): dy and the image are rounded to bf16 on their way into the operands,
 * the sums and dw stay fp32.  Same workspace. */
int sd_conv2d_stem_wgrad_bf16mm(const float* dy, const float* x_nchw, float* dw, const sd_conv_desc* d, int accumulate, void* workspace,
                                size_t workspace_bytes, sd_stream_t stream);
/* ... from a bf16 dy [M][64] (16-byte aligned): the mixed-precision step's stem (row-ring kernel: a block walks down a 128-pixel column
 * strip and fetches two new image rows per tile; dy by LDS-DMA, transposed by ds_read_b64_tr_b16).  Same workspace, fp32 sums and dw. */
int sd_conv2d_stem_wgrad_bf16(const void* dy_bf16, const float* x_nchw, float* dw, const sd_conv_desc* d, int accumulate, void* workspace,
                              size_t workspace_bytes, sd_stream_t stream);

/* BatchNorm2d over [M][C] (M = B*H*W): training statistics (biased var for normalisation,
 * running stats with momentum and the unbiased var, torch semantics), apply (+residual, +ReLU),
 * eval-mode fold into (scale, shift), backward (dy masked by y > 0 when relu; g_out optionally
 * receives the masked gradient for the skip connection). */
size_t sd_col_reduce_workspace_bytes(int64_t M, int C);
int sd_bn_train_stats(const float* x, int64_t M, int C, float eps, float momentum, float* running_mean,
                      float* running_var, float* mean, float* invstd, void* workspace, size_t workspace_bytes,
                      sd_stream_t stream);
/* relu_mask_out (nullable): M*C/4 bytes, bit j of byte i = element 4i+j of the result is positive -- the ReLU mask the backward of a
 * residual layer needs, at 1/16 of the bytes of y (relu mode 3 of sd_bn_bwd / sd_bn_bwd_apply / sd_conv2d_dgrad_bn_reduce). */
int sd_bn_apply(const float* x, float* y, int64_t M, int C, const float* mean, const float* invstd,
                const float* gamma, const float* beta, const float* residual, int relu, uint8_t* relu_mask_out,
                sd_stream_t stream);
int sd_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
               float eps, int C, float* scale, float* shift, sd_stream_t stream);
/* relu: 0 = none, 1 = ReLU mask from the saved output y, 3 = mask bytes of sd_bn_apply passed as `y`, 2 = mask recomputed from x (layers without a residual
 * input: y is not read at all; needs beta). */
int sd_bn_bwd(const float* dy, const float* x, const float* y, int relu, int64_t M, int C, const float* mean,
              const float* invstd, const float* gamma, const float* beta, float* dx, float* g_out, float* dgamma,
              float* dbeta, int accumulate, void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* out[c] (+)= sum_m x[m][c]  (bias gradients). */
int sd_col_sum(const float* x, int64_t M, int C, float* out, int accumulate, void* workspace,
               size_t workspace_bytes, sd_stream_t stream);

/* MaxPool2d(3, 2, 1) NHWC; idx (uint8 per element) = winning tap for the backward. */
int sd_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int B, int Hi, int Wi, int C, sd_stream_t stream);
int sd_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int B, int Hi, int Wi, int C, sd_stream_t stream);
/* Stem tail (network.py:43-45) in training mode, fused: BatchNorm2d (batch statistics given) + ReLU + MaxPool2d(3, 2, 1) straight from
 * the conv output x (B, Hi, Wi, C); the full-resolution activation is never written.  idx as sd_maxpool3x3s2_fwd. */
int sd_bn_relu_maxpool_fwd(const float* x, int B, int Hi, int Wi, int C, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, float* y_pool, uint8_t* idx, sd_stream_t stream);
/* ... and its backward: dpool (B, Ho, Wo, C) -> dx (B, Hi, Wi, C) through max-pool, ReLU and BatchNorm; dgamma / dbeta (+= when
 * accumulate).  workspace = sd_col_reduce_workspace_bytes(B * Hi * Wi, C). */
int sd_maxpool_bn_relu_bwd(const float* dpool, const uint8_t* idx, const float* x, int B, int Hi, int Wi, int C, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta,
                           int accumulate, void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* backward of nn.Upsample(scale_factor=2): dx (B,H,W,C) = 2x2 block sums of dy (B,2H,2W,C) [+ add]. */
int sd_upsample2x_bwd(const float* dy, const float* add, float* dx, int B, int H, int W, int C, sd_stream_t stream);

/* Head (network.py:22-29): 1x1 conv C -> Co with bias; NHWC in, NCHW out (the layout the decoder
 * and the loss consume).  Co <= 32.  C = 128 (the default FPN depth), Co <= 16, HW % 16 == 0 and 16-byte aligned pointers take the
 * MFMA kernels (forward: wave-private LDS-DMA ring + v_mfma_f32_16x16x4_f32; backward: weight-gradient partials likewise, one partial
 * row per wave, before the data gradient); other shapes the generic kernels.  Same arithmetic either way (fp32 products and sums;
 * the order of the sums differs). */
int sd_head_fwd(const float* x_nhwc, const float* w, const float* bias, float* y_nchw, int B, int HW, int C, int Co,
                sd_stream_t stream);
size_t sd_head_bwd_workspace_bytes(int B, int HW, int C, int Co);
int sd_head_bwd(const float* dy_nchw, const float* x_nhwc, const float* w, float* dx_nhwc, float* dw, float* dbias,
                int B, int HW, int C, int Co, int accumulate, void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* The same with the head's input and its gradient in bf16 (mixed-precision training: no fp32 copies of the FPN output); dy, w, dw, dbias
 * stay fp32.  C in {64, 128}, Co <= 8; same workspace size. */
int sd_head_bwd_bf16(const float* dy, const void* x_bf16, const float* w, void* dx_bf16, float* dw, float* dbias, int B, int HW, int C, int Co,
                     int accumulate, void* workspace, size_t workspace_bytes, sd_stream_t stream);

/* torch.optim.Adam step (defaults: no weight decay / amsgrad) over a flat fp32 buffer;
 * grad is multiplied by grad_scale first (1/world_size after a sum all-reduce). */
int sd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                 float lr, float beta1, float beta2, float eps, float grad_scale, sd_stream_t stream);

/* ---- gradient exchange (the reference's step, trainer.py:113-124, is single-device) -------------------
 * Thin wrapper over RCCL, one communicator per process (= per GPU).  librccl.so.1 is dlopen()ed on first use (the copy a
 * PyTorch-ROCm process has already loaded is reused), so libsdnet_hip.so has no link-time dependency on it.
 * Bootstrap: rank 0 calls sd_allreduce_unique_id and hands the 128 bytes to the other ranks by any host channel
 * (torch.distributed store, MPI, a file); every rank then calls sd_allreduce_init with the device it computes on current.
 * sd_allreduce_run: in-place fp32 sum over all ranks of buf[0..count), asynchronous on `stream`.  Positive return values of
 * these four functions are ncclResult_t codes (text in sd_last_error()). */
#define SD_COMM_ID_BYTES 128
int sd_allreduce_unique_id(void* id_out);
int sd_allreduce_init(const void* id, int rank, int world, void** comm_out);
int sd_allreduce_run(void* comm, float* buf, int64_t count, sd_stream_t stream);
int sd_allreduce_destroy(void* comm);

/* Stand-in for ONE RCCL all-reduce launch on a single GPU (no reference counterpart; sizing tool for the data-parallel step, csrc/sd_commsim.hip):
 * `workgroups` persistent blocks of 256 threads / 8 KB LDS copy `move_bytes` (a multiple of 16; the source of `src_bytes` wraps) from src to dst,
 * throttled on the 100 MHz wall clock to `gbps` GB/s in total -- the launch occupies its CUs for move_bytes / gbps like a link-bound collective.
 * `TrainStep(exchange="sim")` issues it at the five bucket trigger points of the backward on the exchange's side stream; bench.py reports the
 * step time with and without it (`north_star.comm_sim`). */
int sd_comm_sim_copy(const void* src, void* dst, size_t src_bytes, size_t move_bytes, int workgroups, float gbps, sd_stream_t stream);

/* The box's own bf16 MFMA ceiling (no reference counterpart; measurement, csrc/sd_bench.hip): 256 blocks x 512 threads run `iters` K = 32
 * steps of a bare LDS-read + v_mfma_f32_16x16x32_bf16 loop (wave tile 128 x 64, the bf16 conv kernels' shape) on the 64 KB of bf16 operands at
 * `operands64k`; `out` receives 256 x 512 floats (so that nothing is optimised away).  sd_mfma_bf16_stream_flops(iters) = the flop count of one
 * launch; bench.py times warm launches with stream events and reports the bf16 forwards as a fraction of that rate beside the nominal peak. */
double sd_mfma_bf16_stream_flops(int iters);
int    sd_mfma_bf16_stream(const void* operands64k, float* out, int iters, sd_stream_t stream);

/* ---- profiler ranges (no reference counterpart: SURVEY.md section 5 lists tracing as absent from the reference) ----
 * roctx ranges on the calling thread, for `rocprofv3 --marker-trace`.  Active only when the environment holds SDNET_ROCTX=1 at the
 * first call AND a marker library (librocprofiler-sdk-roctx / libroctx64) can be dlopen'ed; otherwise every call is a no-op returning 0.
 * sd_range_enabled: 1 when ranges are recorded.  sd_range_library: the soname that was loaded ("" when disabled).
 * The host mirror (structuredetector_amd/utils/trace.py) brackets target rendering, the forward stages, the loss, the backward stages,
 * every gradient bucket and the Adam launch of `TrainStep` with them. */
int         sd_range_enabled(void);
const char* sd_range_library(void);
int         sd_range_push(const char* name);
int         sd_range_pop(void);
int         sd_range_mark(const char* name);

#ifdef __cplusplus
}
#endif
#endif /* SDNET_HIP_H */
