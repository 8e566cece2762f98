/*
 * cgg_hip.h -- C ABI of libcgg_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the hot path of
 * CGG ("Betrayed by Captions"): MSDeformAttn pixel decoder -> masked-cross-attention query decoder ->
 * mask_embed x mask_feature mask logits -> open-vocabulary post-processing.
 *
 * Conventions (every entry point):
 *   - extern "C", plain pointers and sizes; NO torch / C++ types cross this boundary.
 *   - every data pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator in
 *     this repo); the library never allocates or frees caller-visible memory. Its only state: one 64-byte
 *     overflow flag word per device (created by cgg_init(device), or thread-safely on first use -- call
 *     cgg_init before a hipGraph capture) and the read-only switches it took from the environment at load.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and NOTHING SYNCHRONISES, with
 *     exactly these exceptions, each of which copies a few bytes device->host and waits for them (not legal
 *     inside a graph capture): cgg_msda_read_levels, the mmcv-contract entries that take the DEVICE level
 *     table (cgg_msda_forward, cgg_msda_forward_fused, cgg_msda_backward -- each = cgg_msda_read_levels +
 *     the *_hostlevels entry), and cgg_x3_overflow_check.
 *   - return value: 0 (CGG_OK) on success, a negative CGG_E* for argument errors, or a positive
 *     hipError_t if the launch failed. No C++ exception crosses the ABI.
 *     cgg_last_error_string() gives a thread-local description of the last failure.
 *   - re-entrant and thread-safe (autograd / DDP threads pass their own stream).
 *
 * Each function names the reference interface it replaces (paths relative to the reference repo
 * jianzongwu/betrayed-by-captions; "[3P]" = upstream mmcv 1.7.1 / mmdet 2.28.2 symbol that the
 * reference selects through a `type=` string and whose source is not part of the reference tree).
 */
#ifndef CGG_HIP_H_
#define CGG_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGG_VERSION 100 /* 0.1.0 */

/* error codes (negative); positive return values are hipError_t */
#define CGG_OK 0
#define CGG_EINVAL (-1)       /* bad argument (null pointer, non-positive size, ...)            */
#define CGG_EUNSUPPORTED (-2) /* shape / dtype combination this build has no kernel for        */
#define CGG_EALIGN (-3)       /* pointer not aligned as the kernel requires (16 B)             */
#define CGG_ELIBRARY (-4)     /* a vendor library call (hipBLASLt) failed                      */

/* dtype tags */
#define CGG_F32 0
#define CGG_BF16 1
/* kv_dtype of cgg_masked_xattn_forward_lse / cgg_masked_xattn_backward only: f32 rows in memory, products on bf16 MFMA operands
 * (f32 accumulate) -- the throughput-mode training variant of the two kernels */
#define CGG_F32_BF16MFMA 2
#define CGG_F32_X3 3 /* cgg_masked_xattn_forward_lse (and cgg_masked_xattn_backward_x3): f32 rows, products on the f16 x 3 contraction (csrc/x3.h) */

typedef void* cgg_stream_t; /* hipStream_t */

int cgg_version(void);

/* Creates the library's per-device state (the 64-byte x3a overflow flag word) on `device`. Optional -- the first x3a producer on a
 * device creates it thread-safely -- but REQUIRED before capturing x3a kernels into a hipGraph on a device that has not run one yet
 * (the creation is a hipMalloc). Idempotent; restores the caller's current device. */
int cgg_init(int device);
const char* cgg_last_error_string(void);

/* ------------------------------------------------------------------------------------------------
 * K1/K2  Multi-scale deformable attention (sampling + aggregation).
 *
 * Replaces [3P] mmcv `ext_module.ms_deform_attn_forward(value, spatial_shapes, level_start_index,
 * sampling_locations, attention_weights, im2col_step)` / `ms_deform_attn_backward(...)`, i.e.
 * `MultiScaleDeformableAttnFunction`, selected by `type='MultiScaleDeformableAttention'` in
 * configs/instance/coco_b48n17.py:49-58 and executed 6x per forward by the pixel decoder built at
 * open_set/models/mask2former_head.py:112-117 (called :787).
 *
 *   value            [B, Nv, H, D]        f32 (CGG_F32) or bf16 (CGG_BF16); Nv = sum_l H_l*W_l
 *   spatial_shapes   [L, 2] int64 (H_l, W_l)        (device)
 *   level_start      [L]    int64                   (device)
 *   sampling_loc     [B, Nq, H, L, P, 2] f32, (x, y) in [0,1]
 *   attn_weight      [B, Nq, H, L, P]    f32 (already soft-maxed over L*P)
 *   out              [B, Nq, H*D]        f32
 * out[b,q,h*D+c] = sum_{l,p} w * bilinear_zero_pad(value_l[b,:,h,c]; x*W_l-0.5, y*H_l-0.5)
 * (grid_sample(align_corners=False, padding_mode='zeros') semantics).
 * Requires D % 4 == 0, L <= 8.
 * ---------------------------------------------------------------------------------------------- */
int cgg_msda_forward(const void* value, const int64_t* spatial_shapes, const int64_t* level_start,
                     const float* sampling_loc, const float* attn_weight, float* out, int B, int Nv,
                     int H, int D, int L, int Nq, int P, int value_dtype, cgg_stream_t stream);

/* Fused variant: takes the RAW outputs of the `sampling_offsets` (H*L*P*2) and `attention_weights`
 * (H*L*P) linears and does `loc = ref + off / (W_l, H_l)` and the softmax over L*P in the kernel
 * prologue ([3P] MultiScaleDeformableAttention.forward). ref_points [Nq, 2] (x, y) in [0,1] are
 * shared by all levels and batch items (valid_ratios == 1, as in MSDeformAttnPixelDecoder).
 *   offs_logits  [B, Nq, ld]  f32, row = [H*L*P*2 offsets | H*L*P logits], ld >= H*L*P*3        */
int cgg_msda_forward_fused(const void* value, const int64_t* spatial_shapes,
                           const int64_t* level_start, const float* offs_logits, int ld,
                           const float* ref_points, float* out, int B, int Nv, int H, int D, int L,
                           int Nq, int P, int value_dtype, cgg_stream_t stream);

/* The one synchronising step of the mmcv contract, factored out: spatial_shapes [L,2] / level_start [L] (int64, DEVICE, as mmcv's
 * MultiScaleDeformableAttnFunction receives them) -> HOST int32 level_hw [2L] = (H_l, W_l) and level_start [L], validated against
 * Nv. A binding calls it ONCE per spatial_shapes tensor (once per forward pass: the 6 encoder layers and their backwards share the
 * tensor) and then uses the non-synchronising, graph-capturable *_hostlevels entries (INTEGRATION.md shows the stub). */
int cgg_msda_read_levels(const int64_t* spatial_shapes, const int64_t* level_start, int L, int Nv,
                         int32_t* level_hw_host, int32_t* level_start_host, cgg_stream_t stream);

/* Same kernels with the level table given as HOST int32 arrays (level_hw [L,2] = (H_l, W_l),
 * level_start [L]): no device->host read of the table, so the call is legal inside a hipGraph
 * capture. fused != 0 selects the cgg_msda_forward_fused argument meaning (sampling_loc =
 * offs_logits with row stride ld, attn_weight ignored, ref_points used).                          */
int cgg_msda_forward_hostlevels(const void* value, const int32_t* level_hw,
                                const int32_t* level_start, const float* sampling_loc,
                                const float* attn_weight, const float* ref_points, int ld,
                                float* out, int B, int Nv, int H, int D, int L, int Nq, int P,
                                int value_dtype, int fused, cgg_stream_t stream);

/* cgg_msda_forward_fused (host level table, f32) with `value` as the first H*D COLUMNS of wider rows (row stride vld floats, vld % 32
 * == 0): the x3a encoder stream (round 6) projects value, sampling offsets and attention logits of a layer with ONE GEMM whose rows
 * are [value 256 | offsets 192 | logits 96]; value_rows and offs_logits then point into the same (B, Nq, 544) buffer. H = 8, D = 32,
 * L = 3, P = 4 only (CGG_EUNSUPPORTED otherwise). */
int cgg_msda_forward_fused_vld(const float* value_rows, int vld, const int32_t* level_hw, const int32_t* level_start,
                               const float* offs_logits, int ld, const float* ref_points, float* out, int B, int Nv, int H, int D,
                               int L, int Nq, int P, cgg_stream_t stream);

/* Training: the MSDeformAttn prologue ([3P] MultiScaleDeformableAttention.forward: reference_points + offsets / (W_l, H_l),
 * softmax of the attention logits) and its backward as elementwise kernels on the raw rows [offsets H*L*P*2 | logits H*L*P]
 * (f32, row stride ld): loc (B, Nq, H, L, P, 2) / attn (B, Nq, H, L, P) out; backward: grad_rows (ld == 3 H L P) from the
 * gather's grad_loc / grad_attn, the softmax recomputed from the logits. level_hw host [L, 2] = (H_l, W_l); L * P <= 16. */
int cgg_msda_prologue(const float* rows, int ld, const float* ref_points, const int32_t* level_hw, float* loc, float* attn, int B,
                      int Nq, int H, int L, int P, cgg_stream_t stream);
int cgg_msda_prologue_backward(const float* grad_loc, const float* grad_attn, const float* rows, int ld, const int32_t* level_hw,
                               float* grad_rows, int B, int Nq, int H, int L, int P, cgg_stream_t stream);

/* Throughput-mode encoder stream: value bf16, offs_logits bf16 (the raw output of ONE bf16 GEMM over
 * [sampling_offsets; attention_weights]), out bf16 (input of the output_proj GEMM); host level table.    */
int cgg_msda_forward_fused_bf16(const void* value, const int32_t* level_hw, const int32_t* level_start,
                                const void* offs_logits, int ld, const float* ref_points, void* out, int B,
                                int Nv, int H, int D, int L, int Nq, int P, cgg_stream_t stream);
/* Same with `value` HEAD-MAJOR, (B, H, Nv, D): a 128-byte line then holds two x-neighbouring pixels of ONE head instead of
 * half-lines of two heads (H = 8, D = 32, L = 3, P = 4 only). */
int cgg_msda_forward_fused_bf16_hm(const void* value, const int32_t* level_hw, const int32_t* level_start,
                                const void* offs_logits, int ld, const float* ref_points, void* out, int B,
                                int Nv, int H, int D, int L, int Nq, int P, cgg_stream_t stream);

/* Backward of cgg_msda_forward (f32 value). grad_value / grad_loc / grad_attn must be ZEROED by the
 * caller and are accumulated in place (same contract as mmcv's ms_deform_attn_backward).           */
int cgg_msda_backward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                      const float* sampling_loc, const float* attn_weight, const float* grad_out,
                      float* grad_value, float* grad_loc, float* grad_attn, int B, int Nv, int H,
                      int D, int L, int Nq, int P, cgg_stream_t stream);

/* ... with the level table from the host (level_hw = [h0, w0, h1, w1, ...]): no device->host read-back per call, graph-capturable
 * (what the training-time autograd Function of the encoder calls). Round 5: grad_value comes from a sorted-scatter kernel
 * (csrc/msda_bwd.hip) whenever the queries are the pixels of the value pyramid (D == 32, integer scale between levels).
 * overwrite_loc_attn != 0: grad_loc / grad_attn are WRITTEN instead of accumulated (they need not be zeroed; grad_value still
 * must be) -- only valid where cgg_msda_backward_overwrites(...) returns 1 for the same geometry, CGG_EUNSUPPORTED otherwise. */
int cgg_msda_backward_hostlevels(const float* value, const int32_t* level_hw, const int32_t* level_start,
                                 const float* sampling_loc, const float* attn_weight, const float* grad_out,
                                 float* grad_value, float* grad_loc, float* grad_attn, int B, int Nv, int H, int D, int L,
                                 int Nq, int P, int overwrite_loc_attn, cgg_stream_t stream);
/* ... with a workspace (device, 16-B aligned, >= cgg_msda_backward_workspace_bytes(...) bytes; its contents need not be initialised)
 * for the TWO-PASS sorted scatter of grad_value: the first pass sums the corners inside a 4-pixel halo of 2 x 2-coarse-pixel tiles; a
 * tile with FEW corners outside it scatters them itself, a tile with many (>= 256 of its 1 344) flags itself and counts them per
 * region; a device-side classify step lists the regions that have something left, and persistent second-pass workgroups re-sort
 * exactly those corners on 4 x 4-coarse-pixel tiles with a 12-pixel halo (regions with < 768 corners left: straight to atomics). Offsets of several pixels (trained models) then cost a second sort instead of one 128-byte atomic
 * per corner (round 5: 8.5 ms per call at +-8 px against 2.0 ms at the initialisation's +-0.5 px). ws null or workspace_bytes == 0:
 * identical to cgg_msda_backward_hostlevels_2s. Same results as the single pass up to float summation order. */
/* vld: floats per pixel of the `value` AND `grad_value` rows (0 or H*D = packed (B, Nv, H, D)); a padded stride (e.g. 288 for H*D = 256)
 * keeps neighbouring pixels' lines / atomics off the same L2 channels. */
int cgg_msda_backward_hostlevels_ws(const float* value, int vld, const int32_t* level_hw, const int32_t* level_start,
                                    const float* sampling_loc, const float* attn_weight, const float* grad_out,
                                    float* grad_value, float* grad_loc, float* grad_attn, int B, int Nv, int H, int D, int L,
                                    int Nq, int P, int overwrite_loc_attn, void* ws, long long ws_bytes, cgg_stream_t stream,
                                    cgg_stream_t side_stream);
long long cgg_msda_backward_workspace_bytes(const int32_t* level_hw, const int32_t* level_start, int B, int Nv, int H, int D, int L,
                                            int Nq, int P);

/* ... with the split backward's two kernels on two streams (grad_value on `stream`, grad_loc / grad_attn on `side_stream`, forked
 * from and joined back into `stream` by events inside the call; side_stream null or == stream: the one-stream form). */
int cgg_msda_backward_hostlevels_2s(const float* value, const int32_t* level_hw, const int32_t* level_start,
                                 const float* sampling_loc, const float* attn_weight, const float* grad_out,
                                 float* grad_value, float* grad_loc, float* grad_attn, int B, int Nv, int H, int D, int L,
                                 int Nq, int P, int overwrite_loc_attn, cgg_stream_t stream, cgg_stream_t side_stream);
int cgg_msda_backward_overwrites(const int32_t* level_hw, const int32_t* level_start, int B, int Nv, int H, int D, int L, int Nq,
                                 int P);

/* ------------------------------------------------------------------------------------------------
 * K3/K4/K5  Mask logits: mask_pred[b,q,h,w] = sum_c mask_embed[b,q,c] * mask_feature[b,c,h,w]
 *
 * Replaces `torch.einsum('bqc,bchw->bqhw', mask_embed, mask_feature)` and the attention-mask rule
 * `sigmoid(interpolate(mask_pred, size_l, 'bilinear')) < 0.5` at
 * open_set/models/mask2former_head.py:748-759 (10x per forward), and the all-masked-row fix-up at
 * :825-826.
 *
 * mask_feature is packed ONCE per forward into an MFMA-fragment-major bf16 image
 *   packed[b][t][c/8][p%32][c%8]   (t = p/32, 16 B per (c/8, p) slot; tiles zero padded)
 * so that every `v_mfma_f32_32x32x16_bf16` B operand is one coalesced 1-KiB global_load_dwordx4.
 * `lo` (nullable) receives the bf16 residual f - bf16(f): with it the contraction runs as three
 * bf16 MFMAs (hi*hi + hi*lo + lo*hi), |err| ~1e-5 relative = f32-class accuracy; without it the
 * contraction is plain bf16 (north_star's MFMA bf16 path).
 * `pool` in {1,2,4,8}: pool == 1 packs the feature map itself; pool == s > 1 packs the map that
 * `F.interpolate(.., scale 1/s, bilinear, align_corners=False)` reads, i.e. the mean of the 2x2
 * block at rows/cols {s*i+s/2-1, s*i+s/2}: by linearity, logits of the pooled feature ARE the
 * interpolated logits, so the attention mask of a level needs a GEMM over H*W/s^2 pixels only.
 *   feat   [B, C, H, W] f32;  hi/lo  [B, ceil(H/s*W/s / 32), C/8, 32, 8] bf16
 * Requires C % 16 == 0, H % s == 0, W % s == 0.
 * ---------------------------------------------------------------------------------------------- */
int cgg_pack_mask_feature(const float* feat, void* hi, void* lo, int B, int C, int H, int W,
                          int pool, cgg_stream_t stream);

/*   embed   [B, Q, C] f32          (mask_embed MLP output)
 *   hi, lo  packed feature (lo nullable -> bf16 mode)        npix = number of (pooled) pixels
 *   out     [B, Q, npix] f32, nullable (logits not stored)
 *   bits    [B, Q, ceil(npix/32)] u32, nullable: bit (p%32) of word p/32 = (logit < 0), i.e. the
 *           boolean attention mask (True = blocked) shared by all heads -- never repeated x8.
 * Requires C == 256 (feat_channels of every shipped config), Q <= 256 (bf16); split mode: any Q (row groups of 128, in place).  */
int cgg_mask_logits(const float* embed, const void* hi, const void* lo, float* out, uint32_t* bits,
                    int B, int Q, int C, int npix, cgg_stream_t stream);
/* Exact-f32 variant for parity mode (f32 MFMA, f32 products): feat = the UN-packed f32 map [B, C, npix] (full resolution,
 * or the 2x2-mean map the pooled images are made of); same out / bits contract. Requires C == 256, Q <= 128.            */
int cgg_mask_logits_f32(const float* embed, const float* feat, float* out, uint32_t* bits, int B, int Q, int C, int npix,
                        cgg_stream_t stream);

/* Backward of K3 (the einsum of mask2former_head.py:748 differentiated on the training path :851-921):
 *   grad_feat [B, C, npix] f32 = sum_q embed[b][q][c] * grad_out[b][q][p]     (nullable: skipped)
 *   grad_embed[B, Q, C]   f32 = sum_p grad_out[b][q][p] * feat[b][c][p]       (nullable: skipped)
 * embed [B, Q, C] f32, feat [B, C, npix] f32 (the un-packed mask feature), grad_out [B, Q, npix] f32; outputs are written,
 * not accumulated. bf16 MFMA with f32 accumulation; split != 0 = 3 MFMAs on (hi, lo) bf16 pairs (f32-class accuracy).
 * Any Q: query rows are processed in groups of 128 (split) / 256 launches, a later group adding into grad_feat in place
 * (same lane, same element, fixed order). ws: cgg_mask_logits_backward_workspace_bytes(...) bytes (grad_embed partial
 * planes, summed in a fixed order: no atomics). Requires C == 256, npix % 8 == 0.                                     */
int64_t cgg_mask_logits_backward_workspace_bytes(int B, int Q, int C, int npix);
int cgg_mask_logits_backward(const float* embed, const float* feat, const float* grad_out, float* grad_embed,
                             float* grad_feat, void* ws, int B, int Q, int C, int npix, int split, cgg_stream_t stream);

/* rows of `bits` that block every key are cleared (open_set/models/mask2former_head.py:825-826).
 *   bits [rows, words] u32, npix valid bits per row.                                             */
/* Round 5: the consumer-fused form with the QUERY operand stationary in registers (csrc/mask_logits_astat.hip) -- bf16 MFMA,
 * attn-mask bits (logit < 0, mask2former_head.py:749-759) straight from the accumulators, the logits never stored: the form of the
 * einsum that is not bound by its f32 output (67.2 MB of algorithmic bytes at configs[1] instead of 119.6 MB).
 * embed (B, Q, 256) f32, hi = packed bf16 feature, bits (B, Q, ceil(npix / 32)) u32; Q <= 256. Same bits as cgg_mask_logits. */
int cgg_mask_logits_bits_astat(const float* embed, const void* hi, uint32_t* bits, int B, int Q, int C, int npix,
                               cgg_stream_t stream);
int cgg_attn_mask_fix_full_rows(uint32_t* bits, int rows, int npix, cgg_stream_t stream);

/* generic path for level sizes that are not an even integer divisor of the mask-feature size:
 * bilinear (align_corners=False) resize of stored logits [N, H, W] -> [N, h, w], then (x < 0) bits */
int cgg_attn_mask_from_logits(const float* logits, uint32_t* bits, int N, int H, int W, int h,
                              int w, cgg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K6  Masked multi-head cross-attention core (flash-style, split over keys).
 *
 * Replaces the `baddbmm + masked softmax + bmm` inside `nn.MultiheadAttention` reached through
 * [3P] mmcv MultiheadAttention <- DetrTransformerDecoderLayer, called at
 * open_set/models/mask2former_head.py:829-840 with attn_masks=[attn_mask, None].
 *
 *   q     [B, Q, H*D]  f32   (already projected, NOT yet scaled)
 *   kv    [B, S, 2*H*D] f32 or bf16: row = [K(H*D) | V(H*D)]  (projected keys / values)
 *   bits  [B, Q, ceil(S/32)] u32, nullable; bit set = key blocked; shared by heads. A row with all
 *         S bits set must have been cleared by cgg_attn_mask_fix_full_rows (else output is NaN,
 *         exactly like the reference).
 *   out   [B, Q, H*D]  f32   softmax(q k^T * scale + mask) v, heads concatenated
 *   ws    workspace, cgg_masked_xattn_workspace_bytes(...) bytes (split-K partials)
 * Requires D == 32, H*D <= 256, Q <= 128.
 * ---------------------------------------------------------------------------------------------- */
int64_t cgg_masked_xattn_workspace_bytes(int B, int Q, int H, int D, int S);
int cgg_masked_xattn_forward(const float* q, const void* kv, const uint32_t* bits, float* out,
                             void* ws, int B, int Q, int H, int D, int S, float scale,
                             int kv_dtype, cgg_stream_t stream);
/* Same, kv f32 rows [K | V] at row stride ldkv elements (>= 2 H D) and batch stride kv_bstride: a column slice of the merged
 * projection of several decoder layers (0 = dense). */
int cgg_masked_xattn_forward_strided(const float* q, const float* kv, int ldkv, int64_t kv_bstride, const uint32_t* bits,
                                     float* out, void* ws, int B, int Q, int H, int D, int S, float scale, cgg_stream_t stream);
/* cgg_masked_xattn_forward(_strided) with S^T = K Q^T and O^T = V^T P^T on the f32-class f16 x 3 contraction (csrc/xattn_x3.hip):
 * parity mode's inference path for mask2former_head.py:829-840. ldkv / kv_bstride = 0: contiguous (B, S, 2 E) rows. */
int cgg_masked_xattn_forward_x3(const float* q, const float* kv, int ldkv, int64_t kv_bstride, const uint32_t* bits, float* out,
                                void* ws, int B, int Q, int H, int D, int S, float scale, cgg_stream_t stream);
/* Throughput-mode variant: k [B, S, H*D] bf16 and the value projection TRANSPOSED, vt [B, H*D, S] bf16 (computed
 * as Wv x mem^T by the caller), bf16 MFMA for both contractions, f32 softmax statistics and accumulation.
 * Same mask / output / workspace contract. Requires D == 32, Q <= 128, S % 4 == 0.
 * ldk = row stride of k in elements (0 = H*D), vt_bstride = batch stride of vt in elements (0 = H*D*S): k may be a column
 * slice and vt a row block of the merged projection of all decoder layers that read the same level.
 * auto_unmask != 0: a query row whose mask blocks all S keys attends to all of them (mask2former_head.py:825-826) without
 * a prior cgg_attn_mask_fix_full_rows launch; `bits` is left untouched.                                            */
int cgg_masked_xattn_forward_bf16(const float* q, const void* k, const void* vt, const uint32_t* bits, float* out,
                                  void* ws, int B, int Q, int H, int D, int S, float scale, int ldk, int64_t vt_bstride,
                                  int auto_unmask, cgg_stream_t stream);

/* Training pair of K6 ("cgg_masked_xattn_forward ... + backward"): the forward that also saves the
 * log-sum-exp of every (image, head, query) row, and the backward of the attention core -- what autograd derives for the
 * `baddbmm + masked softmax + bmm` inside nn.MultiheadAttention on the training path
 * open_set/models/mask2former_head.py:829-840 (reached from forward_train, :851-921).
 *
 *   lse       [B, H, Q] f32   max + log(sum exp) of the masked, scaled scores (natural log)
 *   out       [B, Q, H*D] f32 the forward's output (delta = rowsum(grad_out o out) is recomputed from it)
 *   grad_out  [B, Q, H*D] f32
 *   grad_q    [B, Q, H*D] f32   d loss / d q      (written, not accumulated)
 *   grad_kv   [B, S, 2*H*D] f32 d loss / d [K | V] (written, not accumulated; every key row is written exactly once)
 *   ws        cgg_masked_xattn_backward_workspace_bytes(...) bytes (per-(chunk, wavefront) grad_q partial planes,
 *             summed in a fixed order: no floating-point atomics, bit-reproducible)
 * The bit mask is consumed as in the forward (bit set = blocked, shared by the heads); nothing of size Q x S is ever
 * stored. f32 MFMA (exact products). Requires D == 32, Q <= 128, kv_dtype == CGG_F32.                              */
int cgg_masked_xattn_forward_lse(const float* q, const void* kv, const uint32_t* bits, float* out, float* lse,
                                 void* ws, int B, int Q, int H, int D, int S, float scale, int kv_dtype,
                                 cgg_stream_t stream);
int64_t cgg_masked_xattn_backward_workspace_bytes(int B, int Q, int H, int D, int S);
int cgg_masked_xattn_backward(const float* q, const void* kv, const uint32_t* bits, const float* out,
                              const float* lse, const float* grad_out, float* grad_q, void* grad_kv, void* ws,
                              int B, int Q, int H, int D, int S, float scale, int kv_dtype, cgg_stream_t stream);
/* cgg_masked_xattn_backward on the f32-class f16 x 3 contraction of csrc/x3.h (parity-mode training, round 6): the five
 * contractions of the backward (scores, dP, dV, dK, dQ) as three v_mfma_f32_32x32x8_f16 per four f32 k-steps instead of four
 * v_mfma_f32_32x32x2_f32 -- as accurate as the f32 MFMA form (tests/test_kernels_gpu.py), 2.7 x less matrix-pipe time.
 * grad_out_amax = device scalar max |grad_out| (cgg_absmax_f32): the per-tensor pre-scale of the gradient operands; q, K, V
 * must be unit scale (|value| < 4094) like every x3 activation operand. Same tensors, workspace and limits otherwise. */
int cgg_masked_xattn_backward_x3(const float* q, const void* kv, const uint32_t* bits, const float* out, const float* lse,
                                 const float* grad_out, const float* grad_out_amax, float* grad_q, void* grad_kv, void* ws, int B,
                                 int Q, int H, int D, int S, float scale, cgg_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * K16  Caption-grounding pair costs and their backward.
 *
 * Replaces the body of open_set/models/losses/grounding_loss.py:32-58 (`grounding_loss`, evaluated for each of the 10
 * decoder outputs at open_set/models/mask2former_head.py:542-548): for every (caption i, image j) pair
 *     s = caption_i predictions_j^T * inv_temperature                                  (T x Q)
 *     cost[0][i][j] = sum_t mask[i][t] sum_q softmax_q(s)[t][q] * (-s[t][q]) / max(sum_t mask[i][t], 1)
 *     cost[1][i][j] = sum_q sum_t softmax_t(s)[t][q] * (-s[t][q]) / Q         (token softmax unmasked, as :44)
 * The reference's B-fold `repeat`s and its (B*B, T, Q) score / attention tensors are never materialised; the remaining
 * steps of the loss (the +100 fill of captions without nouns and the four log-softmax diagonals over the (Bc, Bp)
 * matrices, :60-77) are a few elementwise ops on B x B values and stay with the caller.
 *
 *   pred      [Bp, Q, d] f32   caption embeddings predicted by the head (all gathered images)
 *   cap       [Bc, T, d] f32   text embeddings of the caption nouns
 *   cap_mask  [Bc, T]   i32    1 = token present
 *   cost      [2, Bc, Bp] f32  (written)
 * backward: grad_cost [2, Bc, Bp] -> dsim [Bp, Bc*T, Q] f32 = d loss / d (caption_i[t] . pred_j[q]); the caller finishes
 * with ONE batched GEMM grad_pred[j] = dsim[j]^T x cap.reshape(Bc*T, d) (captions are constants: frozen text encoder).
 * f32 MFMA (exact products). Requires Q <= 256, T <= 64, d % 8 == 0.
 * ---------------------------------------------------------------------------------------------- */
int cgg_grounding_pair_costs(const float* pred, const float* cap, const int32_t* cap_mask, float* cost, int Bp, int Bc,
                             int Q, int T, int d, float inv_temperature, cgg_stream_t stream);
int cgg_grounding_pair_costs_backward(const float* pred, const float* cap, const int32_t* cap_mask,
                                      const float* grad_cost, float* dsim, int Bp, int Bc, int Q, int T, int d,
                                      float inv_temperature, cgg_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * K18  Row kernels of the caption generator + cross-entropy, so that the (B*34, 30522) logits of
 * open_set/models/mask2former_head.py:551-565 (`caption_generator(...)[1]` -> `loss_caption_generation`) are never
 * stored beyond one row chunk: the caller's library GEMM writes a chunk of logits, these kernels consume it.
 *
 *   logits  [M, ld] f32 or bf16 (dtype), N valid columns per row (N and ld multiples of 4 (f32) / 8 (bf16) elements:
 *           pad the vocabulary with -1e30 bias columns, which contribute exp(.) = 0)
 *   target  [M] i64; rows with target == ignore_index (or outside [0, N)) get loss 0 / gradient 0
 * forward : loss[row] = lse[row] - logits[row][target],  lse[row] = log sum_n exp(logits[row][n])  (both f32, written)
 * backward: logits[row][n] <- grad_rows[row] * (exp(logits[row][n] - lse[row]) - [n == target])   IN PLACE (same dtype)
 * ---------------------------------------------------------------------------------------------- */
int cgg_ce_rows_forward(const void* logits, const int64_t* target, float* loss, float* lse, int M, int N, int64_t ld,
                        int64_t ignore_index, int dtype, cgg_stream_t stream);
int cgg_ce_rows_backward(void* logits, const int64_t* target, const float* lse, const float* grad_rows, int M, int N,
                         int64_t ld, int64_t ignore_index, int dtype, cgg_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Prediction-only halves of the Hungarian matching costs on point-sampled mask logits, one pass.
 *
 * Replaces, per training step, the per-(layer, image) elementwise chain of open_set/models/mask2former_head.py:320-390
 * (`_get_target_single` -> mmdet CrossEntropyLossCost(use_sigmoid=True): pos = softplus(-x), neg = softplus(x),
 * einsum(pos, t) + einsum(neg, 1 - t); DiceCost(pred_act=True): sigmoid(x)). With softplus(-x) - softplus(x) = -x the pair of
 * contractions is sp_sum - x . t, so only row sums and sigmoid(x) are needed from the predictions:
 *   x        (rows, P) f32 logits, contiguous, 16-B aligned, P % 4 == 0
 *   sig      (rows, P) f32 out: sigmoid(x)
 *   sp_sum   (rows)    f32 out: sum_p softplus(x)
 *   sig_sum  (rows)    f32 out: sum_p sigmoid(x)   (square = 0)  |  sum_p sigmoid(x)^2   (square = 1: naive_dice = False)
 * ---------------------------------------------------------------------------------------------- */
int cgg_match_cost_rows(const float* x, float* sig, float* sp_sum, float* sig_sum, int rows, int P, int square,
                        cgg_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * HOST function (no device work, no stream): COCO run-length encoding of bit-packed instance masks.
 *
 * Replaces, for the serving / evaluation path, the per-mask `.cpu().numpy()` of open_set/models/maskformer.py:205-208
 * followed by pycocotools `mask.encode` in the dataset's results2json: the (n, H, W/8) bit planes written by
 * cgg_instance_masks_picks(bitpack = 1) are copied to the host ONCE per batch and encoded here on `threads` host threads.
 *
 *   bits    host pointer, n planes of H rows x row_bytes bytes; pixel x of a row = bit (x & 7) of byte (x >> 3);
 *           plane i starts at bits + i * mask_stride_bytes
 *   out     host buffer of out_cap bytes receiving the n COCO "counts" strings back to back (compressed form of
 *           pycocotools' rleToString: column-major runs, first run = zeros); offsets[i] .. offsets[i+1] = string i
 *   offsets host int64[n + 1] (always written)
 * Returns the total number of bytes of all strings; if that exceeds out_cap nothing is copied (call again with a larger
 * buffer); negative = -CGG_E*.
 * ---------------------------------------------------------------------------------------------- */
int64_t cgg_rle_encode_bitmasks(const uint8_t* bits, int n, int H, int W, int64_t mask_stride_bytes, int row_bytes,
                                int threads, uint8_t* out, int64_t out_cap, int64_t* offsets);

/* Throughput-mode self-attention of the query decoder ([3P] DetrTransformerDecoderLayer self_attn, no mask; S = Q <= 128):
 * q [B*Q, ldq] and kv = [k | v] [B*Q, ldkv] f32 rows (as written by the fused q|k|v projection) -> out [B*Q, H*D] f32 =
 * softmax(scale q k^T) v per head. bf16 MFMA operands, f32 accumulation and softmax; D == 32.                       */
int cgg_self_attn_rows_bf16(const float* q, int ldq, const float* kv, int ldkv, float* out, int B, int Q, int H, int D,
                            float scale, cgg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K7/K11  Skinny linear + fused residual LayerNorm for the QUERY side of the decoder (M = B*Q ~ 200 rows):
 * the q / out / self-attention projections and FFN of DetrTransformerDecoderLayer ([3P], called at
 * open_set/models/mask2former_head.py:829-840) and the cls / v2l / mask_embed MLPs of forward_head (:734-746).
 *   y[M,N] = act(x[M,K] @ w[N,K]^T + bias) (+ res)      x row stride ldx, y row stride ldy, res stride ldr
 *   relu != 0 applies ReLU before the residual add; split == 1 -> 3 bf16 MFMAs on (hi, lo) pairs (~2e-5 of sum |x w|),
 *   split == 2 -> exact f32 (f32 MFMA on the un-rounded operands; what parity mode uses), split == 0 -> plain bf16.
 * Requires K % 16 == 0, ldx % 4 == 0.
 * cgg_add_layernorm: y = LayerNorm(a (+ b)) * gamma + beta over the last dim N (b nullable).
 * ---------------------------------------------------------------------------------------------- */
int cgg_linear_rows(const float* x, int ldx, const float* w, const float* bias, const float* res, int ldr,
                    float* y, int ldy, int M, int N, int K, int relu, int split, cgg_stream_t stream);
int cgg_add_layernorm(const float* a, const float* b, const float* gamma, const float* beta, float* y,
                      int rows, int N, float eps, cgg_stream_t stream);

/* Throughput-mode (bf16) successor of cgg_linear_rows. The weight is packed once by cgg_linear_rows_pack into bf16
 * MFMA-B-fragment order (cgg_linear_rows_packed_bytes(N, K) bytes) and then reused:
 *   y[M,N] = x[M,K] @ W^T + bias; ReLU on columns < relu_cols; + res[M,N];
 *   ln_gamma != NULL (N <= 256): y = LayerNorm(y) * gamma + beta in the epilogue;
 *   yp != NULL: yp = y + pos[row % pos_rows] (pos [pos_rows, N]) written in the same pass;
 *   ksplit > 1: K is split over ksplit workgroups; split z writes its partial sums to the plane y + z * M * ldy
 *   (y must hold ksplit planes; bias / res go into plane 0; no ReLU / LayerNorm / yp in that mode) and the consumer
 *   adds the planes (cgg_layernorm_chain with nsum = ksplit) -- deterministic, no atomics.
 *   x2 != NULL: output columns >= x2_col are computed from x2 [M, K] (row stride ldx2) instead of x, and
 *   y2 != NULL: output columns >= y2_col are stored to y2[m * ldy2 + (n - y2_col)] (x2_col, y2_col multiples of 256):
 *   the self-attention q | k | v projections (q, k from `query + query_pos`, v from `query`) as ONE launch.
 * Replaces, per decoder layer, the q/k/v/out projections, FFN, the three post-norm LayerNorms and the `x + pos`
 * adds of DetrTransformerDecoderLayer ([3P]; open_set/models/mask2former_head.py:829-840) and the cls / v2l /
 * mask_embed MLPs of forward_head (:734-746). Requires K % 16 == 0, ldx % 4 == 0.                               */
int64_t cgg_linear_rows_packed_bytes(int N, int K);
int cgg_linear_rows_pack(const float* w, void* packed, int N, int K, cgg_stream_t stream);
int cgg_linear_rows_bf16(const float* x, int ldx, const void* w_packed, const float* bias, const float* res,
                         int ldr, float* y, int ldy, const float* ln_gamma, const float* ln_beta, float ln_eps,
                         const float* pos, int pos_rows, float* yp, int ldyp, int M, int N, int K, int relu_cols,
                         int ksplit, const float* x2, int ldx2, int x2_col, float* y2, int ldy2, int y2_col,
                         cgg_stream_t stream);

/* y = LN_a(sum_{p < nsum} a[p * plane + ...]) ; yp = y + pos[row % pos_rows] (nullable) ; z = LN_b(y) (nullable): the
 * decoder layer's last norm (fed by the nsum split-K planes of cgg_linear_rows_bf16), the next layer's `query +
 * query_pos`, and the head's post_norm (mask2former_head.py:734) in one pass. N <= 1024.                          */
int cgg_layernorm_chain(const float* a, int lda, const float* gamma_a, const float* beta_a, float eps_a,
                        const float* pos, int pos_rows, const float* gamma_b, const float* beta_b, float eps_b,
                        float* y, float* yp, float* z, int rows, int N, int nsum, int64_t plane,
                        cgg_stream_t stream);

/* Tail of one query-decoder layer in ONE launch (C == 256), on the FFN's split-K planes [nsum][M][ld] (bias and
 * residual already in plane 0):  y = LN_a(sum planes) (the layer's last norm; [3P] DetrTransformerDecoderLayer),
 * yp = y + pos[row % pos_rows] (nullable), z = LN_b(y) (decoder post_norm, mask2former_head.py:734),
 * mask_embed = W3 relu(W2 relu(W1 z + b1) + b2) + b3 (:741-746) and, when wq != NULL, qn = Wq (y + pos) + bq -- the next
 * layer's cross-attention query projection (:829). w1 / w2 / w3 / wq: 256 x 256 weights packed by cgg_linear_rows_pack.
 * Same arithmetic per stage as cgg_layernorm_chain + 4 x cgg_linear_rows_bf16 (bf16 operands, f32 accumulate).     */
int cgg_decoder_tail_bf16(const float* planes, int nsum, int64_t plane_stride, int ld, const float* gamma_a,
                          const float* beta_a, float eps_a, const float* pos, int pos_rows, const float* gamma_b,
                          const float* beta_b, float eps_b, const void* w1, const float* b1, const void* w2,
                          const float* b2, const void* w3, const float* b3, const void* wq, const float* bq, float* y,
                          float* yp, float* mask_embed, float* qn, int M, int C, cgg_stream_t stream);

/* Middle of a query-decoder layer in ONE launch (C == 256): x1 = LayerNorm(core Wo^T + bo + res) (attention output
 * projection, residual, post-norm; [3P] DetrTransformerDecoderLayer) and, when wqkv != NULL, the self-attention's fused
 * projection of the result: q = (x1 + pos) Wq^T + bq -> q [M, 256]; [k | v] = [(x1 + pos) Wk^T + bk | x1 Wv^T + bv] ->
 * kv [M, 512]. wo: 256 x 256, wqkv: 768 x 256 ([Wq; Wk; Wv] = in_proj_weight), packed by cgg_linear_rows_pack.    */
int cgg_decoder_mid_bf16(const float* core, int ldc, const void* wo, const float* bo, const float* res, int ldr,
                         const float* gamma, const float* beta, float eps, const float* pos, int pos_rows,
                         const void* wqkv, const float* bqkv, float* x1, float* q, float* kv, int M, int C,
                         cgg_stream_t stream);

/* The decoder layer's FFN ([3P] mmcv FFN: Linear(256, F) - ReLU - Linear(F, 256), + identity) in ONE launch: the split-K
 * partition of the second projection is the column partition of the first, so each workgroup keeps its 32 x 256 hidden
 * block in LDS. planes [F / 256][M][256] f32: partial sums of x + (relu(x W1^T + b1)) W2^T + b2 (b2 and x in plane 0),
 * to be added in plane order by cgg_decoder_tail_bf16 / cgg_layernorm_chain. w1: F x 256, w2: 256 x F, packed by
 * cgg_linear_rows_pack; F % 256 == 0.                                                                              */
int cgg_decoder_ffn_bf16(const float* x, int ldx, const void* w1, const float* b1, const void* w2, const float* b2,
                         float* planes, int M, int C, int F, cgg_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Parity mode ("x3"): the same query-side kernels on f32-class contractions. The reference runs every linear of
 * open_set/models/mask2former_head.py:711-761 and of the [3P] DetrTransformerDecoderLayer in f32; here each f32 operand is
 * split into two f16 pieces and every product is three v_mfma_f32_32x32x16_f16 into one f32 accumulator (csrc/x3.h:
 * 22-bit operands, pre-scaled by powers of two, f32-GEMM accuracy; activations must satisfy |x| < 4094). The weight is
 * packed once by cgg_x3_pack into an "x3 image" of cgg_x3_packed_bytes(N, K) bytes (hi fragments, lo fragments, per-column
 * un-scaling factors; K % 16 == 0). The *_x3 entry points take the argument lists of their *_bf16 twins with x3 images in
 * place of the bf16 packed weights.
 * ---------------------------------------------------------------------------------------------- */
int64_t cgg_x3_packed_bytes(int N, int K);
int cgg_x3_pack(const float* w, void* packed, int N, int K, cgg_stream_t stream);
int cgg_linear_rows_x3(const float* x, int ldx, const void* w_x3, const float* bias, const float* res,
                       int ldr, float* y, int ldy, const float* ln_gamma, const float* ln_beta, float ln_eps,
                       const float* pos, int pos_rows, float* yp, int ldyp, int M, int N, int K, int relu_cols,
                       int ksplit, const float* x2, int ldx2, int x2_col, float* y2, int ldy2, int y2_col,
                       cgg_stream_t stream);
int cgg_decoder_tail_x3(const float* planes, int nsum, int64_t plane_stride, int ld, const float* gamma_a,
                        const float* beta_a, float eps_a, const float* pos, int pos_rows, const float* gamma_b,
                        const float* beta_b, float eps_b, const void* w1, const float* b1, const void* w2,
                        const float* b2, const void* w3, const float* b3, const void* wq, const float* bq, float* y,
                        float* yp, float* mask_embed, float* qn, int M, int C, cgg_stream_t stream);
int cgg_decoder_mid_x3(const float* core, int ldc, const void* wo, const float* bo, const float* res, int ldr,
                       const float* gamma, const float* beta, float eps, const float* pos, int pos_rows,
                       const void* wqkv, const float* bqkv, float* x1, float* q, float* kv, int M, int C,
                       cgg_stream_t stream);
int cgg_decoder_ffn_x3(const float* x, int ldx, const void* w1, const float* b1, const void* w2, const float* b2,
                       float* planes, int M, int C, int F, cgg_stream_t stream);

/* Parity mode's large contractions (csrc/x3_gemm.hip), f32 in / f32 out, f32-class arithmetic, w_x3 = cgg_x3_pack image:
 *   cgg_gemm_x3:      out[M, N] (row stride ldc) = act(a[M, K] (row stride lda) W^T * colscale + bias (+ res[M, N], stride ldr));
 *                     relu != 0 applies max(., 0) last. K % 32 == 0, lda % 4 == 0, a 16-byte aligned. bias / res nullable.
 *   cgg_conv_x3_nhwc: the same contraction as an implicit GEMM over a channel-last map x [B, H, W, C] f32 (C % 32 == 0):
 *                     out [B, OH, OW, N] = act(conv(x, W, stride, pad) + bias (+ res [B, OH, OW, N])); W's x3 image is packed
 *                     from the filter re-laid-out as [N][KH][KW][C] (k = (ky, kx, c)). No im2col matrix is written.
 * Replace the f32 library GEMMs / convolutions under [3P] MSDeformAttnPixelDecoder, nn.MultiheadAttention's key / value
 * projections (mask2former_head.py:787, :829-840) and the ResNet backbone in parity mode.                          */
/* Channel-last pieces of parity mode's pixel decoder: cgg_group_norm_nhwc with an F32 input map x [B, HW, C] (same outputs
 * and options; y32 may alias x), and the x3 images of the channel-last f32 mask feature for cgg_mask_logits' split mode: up to
 * 4 pools (1 = full resolution, 2 / 4 / 8 = the 2x2-mean images of the decoder levels) from one launch; hi_host / lo_host /
 * pools_host are HOST arrays of n device pointers / ints, each image [B, ceil(npix / 32), C / 8, 32, 8] 16-bit pieces. */
int cgg_group_norm_nhwc_f32(const float* x, const float* gamma, const float* beta, void* ws, int B, int HW, int C, int groups,
                            float eps, int relu, const float* up_src, int up_h, int up_w, int64_t up_bstride, int W,
                            float* y32, int64_t y32_bstride, void* y16, const float* pos, void* yp16, int64_t y16_bstride,
                            cgg_stream_t stream);
int cgg_pack_mask_feature_nhwc_f32_x3(const float* feat, void* const* hi_host, void* const* lo_host, const int* pools_host,
                                      int n, int B, int C, int H, int W, cgg_stream_t stream);
/* Training (parity mode), matching costs: mask_embed . sample(mask_feature) at the P_group random points of each of the
 * P_total / P_group decoder layers (open_set/models/mask2former_head.py:899-921 samples the (B, Q, h, w) mask logits with [3P]
 * mmcv.ops.point_sample; sample(E F) = E sample(F)). feat [B, H, W, C] channel-last f32, pts [B, P_total, 2] in [0, 1] (x, y), the
 * points of layer g at [g P_group, (g + 1) P_group). Writes the SAMPLES as x3 images (the packed B operand of cgg_mask_logits'
 * split mode, as cgg_pack_mask_feature_nhwc_f32_x3 would from an f32 sample tensor that is never materialised): hi / lo
 * [P_total / P_group][B][P_group / 32][C / 8][32][8] 16-bit pieces -- image (g, b) holds layer g's samples of batch image b, so
 * cgg_mask_logits(embed_g, hi + g * B * image, lo + ..., npix = P_group) is that layer's (B, Q, P_group) point logits.
 * grid_sample arithmetic (bilinear, zeros padding, align_corners = False) in ATen's order. P_group % 32 == 0, C % 8 == 0. */
int cgg_point_sample_nhwc_x3(const float* feat, const float* pts, void* hi, void* lo, int B, int H, int W, int C, int P_total,
                             int P_group, cgg_stream_t stream);
/* Parity mode's twin of cgg_encoder_layer_tail_bf16: the whole post-attention half of an encoder layer as ONE launch on the
 * f32-class contraction (csrc/encoder_tail_x3.hip):
 *   x1 = LayerNorm0(x + a Wo^T + bo);   y = LayerNorm1(x1 + W2 relu(W1 x1 + b1) + b2)
 * a32 = attention rows (MSDeformAttn output before output_proj), x32 = layer input rows, both (M, 256) f32; wo / w1 / w2 = x3
 * images (cgg_x3_pack) of the (256 x 256), (F x 256), (256 x F) weights, F % 256 == 0. Outputs y32 = y and (nullable) yp32 =
 * y + pos[row % pos_rows], (M, 256) f32. x1 and the (M x F) hidden activation never reach memory. */
int cgg_encoder_layer_tail_x3(const float* a32, const float* x32, const void* wo_x3, const float* bo, const float* gamma0,
                              const float* beta0, float eps0, const void* w1_x3, const float* b1, const void* w2_x3,
                              const float* b2, const float* gamma1, const float* beta1, float eps1, const float* pos,
                              int pos_rows, float* y32, float* yp32, int M, int C, int F, cgg_stream_t stream);
int cgg_gemm_x3(const float* a, int lda, const void* w_x3, const float* bias, const float* res, int ldr, float* out, int ldc,
                int M, int N, int K, int relu, cgg_stream_t stream);
/* Parity mode's ResNet stem: cgg_stem_conv7x7_nchw on the f32-class contraction -- w_packed = hi | lo f16 A fragments of the
 * per-output-channel pre-scaled, BN-folded filter (2 x cgg_stem_conv7x7_packed_bytes() bytes), wscale[64] un-scales the
 * accumulators; out = RAW convolution (B, Ho, Wo, 64) f32 channel-last -- and the f32 (bias, ReLU, 3x3 / s2 / p1 max-pool) pass
 * over it (x (B, H, W, C) f32 -> y (B, Ho, Wo, C) f32, C % 4 == 0). Replace MIOpen's f32 stem + relu + max_pool + layout copy. */
int cgg_stem_conv7x7_x3_nchw(const float* img, const void* w_packed, const float* wscale, float* out, int B, int H, int W,
                             cgg_stream_t stream);
int cgg_bias_relu_maxpool_nhwc_f32(const float* x, const float* bias, float* y, int B, int H, int W, int C, cgg_stream_t stream);
/* cgg_gemm_x3 with a row-periodic residual (res_mod > 0: row m adds res[m % res_mod], e.g. a per-token table shared by the
 * images of a batch) and a column split (out2 != NULL: columns >= col2, a multiple of 32, are stored to out2[m * ldc2 + n - col2]):
 * the MSDeformAttn layer's value_proj and sampling_offsets | attention_weights projections as ONE launch over the rows x --
 * (x + pos) Wc^T + bc = x Wc^T + (pos Wc^T + bc), the bracket being a per-token table (mmcv MultiScaleDeformableAttention.forward). */
int cgg_gemm_x3_ex(const float* a, int lda, const void* w_x3, const float* bias, const float* res, int ldr, int res_mod, float* out,
                   int ldc, float* out2, int ldc2, int col2, int M, int N, int K, int relu, cgg_stream_t stream);
int cgg_conv_x3_nhwc(const float* x, const void* w_x3, const float* bias, const float* res, float* out, int B, int H, int W,
                     int C, int N, int KH, int KW, int stride, int pad, int relu, cgg_stream_t stream);

/* Round 4: the same contractions on PRE-SPLIT activation rows ("x3a", csrc/x3.h + csrc/x3s_gemm.hip). An x3a tensor has the shape,
 * bytes and addressing of the f32 tensor it stands for (channel-last, channels % 8 == 0), but every group of 8 consecutive
 * channels holds [8 x f16 hi | 8 x f16 lo] of 16 a -- the operand form of the f16 x 3 MFMA contraction -- so the GEMM moves both
 * operands HBM -> LDS by LDS-DMA with no conversion in its loop, and its epilogue can emit the next GEMM's A operand directly.
 *   cgg_gemm_x3s:      out[M, N] = act(a W^T * colscale + bias (+ res[m or m % res_mod])); a = x3a rows (row stride lda elements,
 *                      lda % 8 == 0), res_fmt 0 none / 1 f32 / 2 x3a, out_fmt 1 f32 / 2 x3a; K % 32 == 0, N % 8 == 0.
 *   cgg_conv_x3s_nhwc: implicit GEMM over a channel-last x3a map [B, H, W, C] (C % 32 == 0, KH * KW <= 32).
 *   cgg_x3a_encode / cgg_x3a_decode: f32 <-> x3a over n contiguous elements (n % 8 == 0): API edges and tests.
 *   cgg_x3_overflow_check: a stored x3a value must satisfy |a| < 4094 (f16 range of 16 a). Every x3a producer ORs 1 into a
 *                      per-device flag word when it sees a larger (or non-finite) value; this copies the word to *value_host
 *                      (synchronises the stream) and optionally clears it -- the caller turns it into an error instead of
 *                      letting inf / NaN masks through (open_set/models/mask2former_head.py:763-849 computes in f32, which
 *                      has no such limit).
 *   cgg_gemm_x3s_cfg / cgg_conv_x3s_nhwc_cfg: the same calls with the tile configuration (0 .. 17; -1 = by shape) as an ARGUMENT:
 *                      tests sweep every instantiation, benches compare them. (Round 4's process-global
 *                      cgg_gemm_x3s_force_config is gone: the library keeps no mutable state besides the overflow flag word.)
 * Same reference call sites as cgg_gemm_x3 / cgg_conv_x3_nhwc above.                                                   */
int cgg_gemm_x3s(const void* a_x3a, int lda, const void* w_x3, const float* bias, const void* res, int ldr, int res_fmt,
                 int res_mod, void* out, int ldc, int out_fmt, int M, int N, int K, int relu, cgg_stream_t stream);
/* cgg_gemm_x3s over a stack of images: row m = row (m % rows_per_image) of image m / rows_per_image, image stride a_bstride elements. */
int cgg_gemm_x3s_batched(const void* a_x3a, int lda, int rows_per_image, int64_t a_bstride, const void* w_x3, const float* bias,
                         const void* res, int ldr, int res_fmt, int res_mod, void* out, int ldc, int out_fmt, int M, int N, int K,
                         int relu, cgg_stream_t stream);
int cgg_conv_x3s_nhwc(const void* x_x3a, const void* w_x3, const float* bias, const void* res, int res_fmt, void* out,
                      int out_fmt, int B, int H, int W, int C, int N, int KH, int KW, int stride, int pad, int relu,
                      cgg_stream_t stream);
int cgg_x3a_encode(const float* x, void* out_x3a, int64_t n, cgg_stream_t stream);
int cgg_x3a_decode(const void* x_x3a, float* out, int64_t n, cgg_stream_t stream);
int cgg_x3_overflow_check(int reset, int* value_host, cgg_stream_t stream);
int cgg_gemm_x3s_cfg(const void* a_x3a, int lda, const void* w_x3, const float* bias, const void* res, int ldr, int res_fmt,
                     int res_mod, void* out, int ldc, int out_fmt, int M, int N, int K, int relu, int cfg, cgg_stream_t stream);
int cgg_conv_x3s_nhwc_cfg(const void* x_x3a, const void* w_x3, const float* bias, const void* res, int res_fmt, void* out,
                          int out_fmt, int B, int H, int W, int C, int N, int KH, int KW, int stride, int pad, int relu, int cfg,
                          cgg_stream_t stream);
/* x3a-producing / -consuming twins of the stream's non-GEMM kernels (round 4):
 *   cgg_bias_relu_maxpool_nhwc_f32_x3a: the stem's (bias, ReLU, 3x3 / s2 max-pool) pass with the pooled map written as x3a (C % 8 == 0);
 *   cgg_group_norm_nhwc_f32_x3a:        cgg_group_norm_nhwc_f32 with y (and yp = y + pos[pixel], nullable) written as x3a rows at
 *                                       y + b * y_bstride; up_src_x3a (nullable) = the low-resolution x3a map whose bilinear x2
 *                                       up-sample is added before the ReLU (the FPN's `cur + F.interpolate(out)`);
 *   cgg_encoder_layer_tail_x3a:         cgg_encoder_layer_tail_x3 with the layer input x and both outputs as x3a rows.          */
int cgg_bias_relu_maxpool_nhwc_f32_x3a(const float* x, const float* bias, void* y_x3a, int B, int H, int W, int C,
                                       cgg_stream_t stream);
int cgg_group_norm_nhwc_f32_x3a(const float* x, const float* gamma, const float* beta, void* ws, int B, int HW, int C, int groups,
                                float eps, int relu, const void* up_src_x3a, int up_h, int up_w, int64_t up_bstride, int W,
                                void* y_x3a, int64_t y_bstride, const float* pos, void* yp_x3a, cgg_stream_t stream);

/* Backward of cgg_group_norm_nhwc_f32 (training, parity mode: the pixel decoder's finest FPN level kept channel-last; replaces
 * autograd's native_group_norm_backward + threshold_backward + upsample_bilinear2d_backward on NCHW copies behind the [3P]
 * MSDeformAttnPixelDecoder lateral / output ConvModules): x, dy (and y = the forward's output when relu) (B, HW, C) f32 channel-last,
 * stats = the forward workspace's head (B, groups, 2) = (mean, variance), ws >= cgg_group_norm_nhwc_backward_workspace_bytes.
 * -> dx (B, HW, C); tot (B, groups, 16): per-image pieces of (dgamma | dbeta) (dgamma[8 g + k] = sum_b tot[b][g][k], dbeta[8 g + k] =
 * sum_b tot[b][g][8 + k]); dlo (nullable, only without relu): gradient of the (B, lo_h, lo_w, C) map the forward up-sampled and added. */
int64_t cgg_group_norm_nhwc_backward_workspace_bytes(int B, int HW, int groups);
int cgg_group_norm_nhwc_f32_backward(const float* x, const float* y, const float* dy, const float* stats, const float* gamma, void* ws,
                                     int B, int HW, int C, int groups, float eps, int relu, float* dx, float* tot, float* dlo, int lo_h,
                                     int lo_w, int W, int dx_padded, cgg_stream_t stream);
/* cgg_group_norm_nhwc_f32 with y written into the interior of a (B, H + 2, W + 2, C) channel-last map whose one-pixel border the caller
 * zeroed (W = map width); dx_padded above is the same layout for the backward's dx: the padded maps of the x3 training convolution. */
int cgg_group_norm_nhwc_f32_padout(const float* x, const float* gamma, const float* beta, void* ws, int B, int HW, int C, int groups,
                                   float eps, int relu, const float* up_src, int up_h, int up_w, int64_t up_bstride, int W,
                                   float* y_padded, cgg_stream_t stream);
int cgg_encoder_layer_tail_x3a(const float* a32, const void* x_x3a, const void* wo_x3, const float* bo, const float* gamma0,
                               const float* beta0, float eps0, const void* w1_x3, const float* b1, const void* w2_x3,
                               const float* b2, const float* gamma1, const float* beta1, float eps1, const float* pos,
                               int pos_rows, void* y_x3a, void* yp_x3a, int M, int C, int F, cgg_stream_t stream);

/* Batched transpose of f32 matrices, in (B, R, C) -> out (B, C, R): the NCHW <-> NHWC layout changes around the x3 kernels under
 * autograd (torch `x.permute(0, 2, 3, 1).contiguous()` and back), 64 x 64 tiles through LDS. */
int cgg_transpose_f32(const float* in, float* out, int B, int R, int C, cgg_stream_t stream);
/* in (B, C, H, W) f32 contiguous -> the interior of out (B, H + 2, W + 2, C) (channel-last, one-pixel border NOT written: the caller
 * zeroes it): padded channel-last maps for the x3 training convolution's weight-gradient taps without a padding copy. */
int cgg_nchw_to_nhwc_pad1_f32(const float* in, float* out, int B, int C, int H, int W, cgg_stream_t stream);

/* Training in parity mode: the weight gradient of a linear layer, dW[n][k] = sum_m dy[m][n] x[m][k] -- autograd's
 * `grad_output.t() @ input` behind the F.linear calls of the [3P] MSDeformAttn encoder layers (mask2former_head.py:787) -- on the
 * f32-class f16 x 3 contraction; both row-major operands reach the MFMA through LDS transpose reads (csrc/wgrad_x3.hip). ws
 * (cgg_wgrad_x3_workspace_bytes) receives *splits_out partial (N, K) f32 matrices over contiguous row ranges; dW is their sum.
 * dy (M, N) rows at stride ldy, x (M, K) rows at stride ldx; N, K, ldy, ldx % 4 == 0; |values| < 4094. */
int64_t cgg_wgrad_x3_workspace_bytes(int M, int N, int K);
int cgg_wgrad_x3(const float* dy, int ldy, const float* x, int ldx, float* ws, int* splits_out, int M, int N, int K,
                 cgg_stream_t stream);
/* ... with the bias gradient db[n] = sum_m dy[m][n] (autograd's `grad_output.sum(0)`) from the same pass over dy: ws_bias
 * (>= splits x N floats, splits = workspace bytes / (4 N K)) receives one partial row per split. */
int cgg_wgrad_bias_x3(const float* dy, int ldy, const float* x, int ldx, float* ws, float* ws_bias, int* splits_out, int M, int N,
                      int K, cgg_stream_t stream);

/* Top-k SELECTION per row: idx[r, 0:k] = indices of the k largest x[r, 0:N] (row stride ld), in NO particular order, ties at the
 * threshold broken arbitrarily -- the importance sampling of the mask losses ([3P] mmdet get_uncertain_point_coords_with_randomness,
 * called at mask2former_head.py:605, keeps the 9 408 most uncertain of 37 632 random points per matched query; torch.topk sorts). */
int cgg_topk_select(const float* x, int ld, int rows, int N, int k, int64_t* idx, cgg_stream_t stream);

/* Per-tensor pre-scale of the x3 contractions' grad_output operands (round 5). The fixed 2^4 pre-scale of csrc/x3.h suits O(1)
 * activations; autograd's grad_output behind the F.linear / conv calls of mask2former_head.py:787 is not unit scale (|g| ~ 1e-6
 * keeps ~10 of the pair's 22 bits). cgg_absmax_f32 writes max |x| over an (M, N) f32 matrix (row stride ld; N, ld % 4 == 0) to the
 * device scalar *amax in one streaming pass; the *_scaled entry points pre-scale that operand by 2^(9 - floor(log2 amax)) instead
 * of 2^4 and fold the inverse into their epilogue (exact). amax pointers are nullable (= the fixed 2^4). Otherwise the
 * contracts of cgg_gemm_x3 / cgg_conv_x3_nhwc / cgg_wgrad_x3 / cgg_wgrad_bias_x3 (ws_bias nullable here). */
int cgg_absmax_f32(const float* x, int ld, int M, int N, float* amax, cgg_stream_t stream);
/* ReLU backward + the same maximum in one pass: g[i] = y[i] > 0 ? gy[i] : 0 (autograd's threshold_backward behind a ReLU whose
 * OUTPUT is y: the Bottleneck ReLUs of the trainable ResNet stage, [3P] mmdet ResNet under mask2former_head.py:787's inputs) and
 * *amax = max |g|. Dense f32 tensors of n elements, n % 4 == 0; g may alias gy. */
int cgg_relu_bwd_absmax_f32(const float* gy, const float* y, float* g, long long n, float* amax, cgg_stream_t stream);
/* cgg_gemm_x3_scaled for the backward of a fused training layer: `mask` (M, N; nullable) zeroes the result where mask <= 0 (the
 * ReLU backward of the layer whose output `mask` is), out_amax (device scalar, nullable) receives max |out| from the epilogue. */
int cgg_gemm_x3_bwd(const float* a, int lda, const float* a_amax, const void* w_x3, const float* mask, int ldm, float* out, int ldc,
                    float* out_amax, int M, int N, int K, cgg_stream_t stream);
int cgg_gemm_x3_scaled(const float* a, int lda, const float* a_amax, const void* w_x3, const float* bias, const float* res, int ldr,
                       float* out, int ldc, int M, int N, int K, int relu, cgg_stream_t stream);
int cgg_conv_x3_nhwc_scaled(const float* x, const float* x_amax, const void* w_x3, const float* bias, const float* res, float* out,
                            int B, int H, int W, int C, int N, int KH, int KW, int stride, int pad, int relu, cgg_stream_t stream);
int cgg_wgrad_x3_scaled(const float* dy, int ldy, const float* dy_amax, const float* x, int ldx, float* ws, float* ws_bias,
                        int* splits_out, int M, int N, int K, cgg_stream_t stream);

/* Encoder-stream residual LayerNorm (N == 256): y = LN(a + b) * gamma + beta, a / b f32 or bf16 (b nullable), with up
 * to three outputs written in the same pass: y32 (f32), y16 = bf16(y), yp16 = bf16(y + pos[row % pos_rows]).  */
int cgg_add_layernorm_ex(const void* a, int a_dtype, const void* b, int b_dtype, const float* gamma, const float* beta,
                         const float* pos, int pos_rows, float* y32, void* y16, void* yp16, int rows, int N,
                         float eps, cgg_stream_t stream);

/* Backward of y = LayerNorm(a + b) over N = 256 channels (training; forward = cgg_add_layernorm_ex): one pass over the rows,
 * statistics recomputed, upstream gradient = dy (f32) + dy16a + dy16b (bf16 gradients of the forward's bf16(y) / bf16(y + pos)
 * outputs; each nullable, at least one given), d/da = d/db = dx (f32; dx16 = optional bf16 copy for a bf16 `b`), per-workgroup partial sums of
 * (dgamma | dbeta): partial (cgg_add_layernorm_backward_partials(rows), 512) f32, to be summed over dim 0 by the caller.
 * Replaces autograd's add / layer_norm backward kernels on the encoder stream of [3P] BaseTransformerLayer. */
int64_t cgg_add_layernorm_backward_partials(int rows);
int cgg_add_layernorm_backward(const float* dy, const void* dy16a, const void* dy16b, const float* a, const void* b, int b_dtype,
                               const float* gamma, float eps, float* dx, void* dx16, float* partial, int rows, int N,
                               cgg_stream_t stream);
/* ... and max |dx| as a device scalar (dx_amax, written by the call; b nullable = plain LayerNorm(a)): the per-tensor pre-scale of
 * the f16 x 3 contractions that take dx as grad_output (cgg_gemm_x3_scaled / cgg_wgrad_x3_scaled) without a pass over dx. */
int cgg_add_layernorm_backward_amax(const float* dy, const void* dy16a, const void* dy16b, const float* a, const void* b, int b_dtype,
                                    const float* gamma, float eps, float* dx, void* dx16, float* partial, float* dx_amax, int rows,
                                    int N, cgg_stream_t stream);

/* Input projections of one MSDeformAttn encoder layer as ONE launch ([3P] MultiScaleDeformableAttention.forward:
 * value_proj / sampling_offsets / attention_weights; layers built at open_set/models/mask2former_head.py:112-117):
 *   value (M, 256) bf16 = x16 Wv^T + bv,   offs (M, NC) bf16 = xp16 Wc^T + bc   (Wc = [W_offsets; W_attention_weights],
 *   NC = 3 * heads * levels * points, 256 .. 384 in steps of 32: 288 for the 3-level encoder, 384 for 4 levels),
 * x16 / xp16 (M, 256) bf16 rows; xp16 == NULL: xp = bf16(x16 + pos16[row % pos_rows]) is formed in the kernel from the bf16
 * table pos16 (pos_rows, 256) and the `x + pos` rows are never stored. value_head_major_rows = N > 0: `value` is written
 * head-major, (M / N, 8, N, 32), for cgg_msda_forward_fused_bf16_hm; 0: (M, 256) rows. Biases f32; weights (N x 256 f32, N = 256 / NC) packed once by cgg_encoder_proj_pack
 * (bf16 MFMA-B fragments, cgg_linear_rows_packed_bytes(N, 256) bytes, output columns interleaved for wide stores). */
int cgg_encoder_proj_pack(const float* w, void* packed, int N, int K, cgg_stream_t stream);
int cgg_encoder_proj_bf16(const void* x16, const void* xp16, const void* pos16, int pos_rows, const void* wv_packed,
                          const float* bv, const void* wc_packed, const float* bc, void* value, void* offs, int M, int C, int NV,
                          int NC, int value_head_major_rows, cgg_stream_t stream);

/* K / V projections of the query decoder for one memory level, all decoder layers that read the level stacked (NK = n * 256
 * outputs; open_set/models/mask2former_head.py:795-812 feeding the cross-attention in_proj of [3P] nn.MultiheadAttention):
 *   k  (B * hw, NK) bf16 = mp16 Wk^T + bk;     vt (B, NK, hw) bf16 = Wv m16^T   (value projection transposed, no bias)
 * m16 / mp16 (B * hw, 256) bf16 rows, any hw (hw % 64 != 0 -- the 1050 / 4200 / 16800 keys of an 800 x 1333 input -- cuts the blocks
 * per image and masks the last one's stores: 4-byte vt stores for even hw, 2-byte for odd); wk packed by cgg_decoder_kv_pack_k, wv by cgg_linear_rows_pack (both
 * cgg_linear_rows_packed_bytes(NK, 256) bytes); bk f32. */
int cgg_decoder_kv_pack_k(const float* w, void* packed, int N, int K, cgg_stream_t stream);
int cgg_decoder_kv_proj_bf16(const void* m16, const void* mp16, const void* wk_packed, const float* bk, const void* wv_packed,
                             void* k, void* vt, int B, int hw, int C, int NK, cgg_stream_t stream);

/* ResNet-50 layer1 identity Bottleneck (mmdet ResNet `Bottleneck.forward`, configs/instance/coco_b48n17.py:17-26), BN folded,
 * channel-last bf16, as ONE launch: y = relu(conv1x1(relu(conv3x3(relu(conv1x1(x, W1) + b1), W2) + b2), W3) + b3 + x).
 * x / y (B, H, W, 256) bf16, H % 8 == 0, W % 16 == 0; all weights packed by cgg_linear_rows_pack: W1 (64 x 256), W2 as the
 * (64 x 576) matrix with K = (ky * 3 + kx) * 64 + c_in, W3 (256 x 64) with its ROWS permuted so that packed row 32 T + j is output
 * channel 64 (T / 2) + 4 (j / 2) + 2 (T & 1) + (j & 1) (see ops.pack_bottleneck64); biases f32. */
int cgg_bottleneck64_bf16(const void* x, const void* w1_packed, const float* b1, const void* w2_packed, const float* b2,
                          const void* w3_packed, const float* b3, void* y, int B, int H, int W, int C, int CMID, cgg_stream_t stream);

/* Encoder-stream FFN block of the pixel decoder as ONE launch ([3P] BaseTransformerLayer 'ffn' + 'norm' of the
 * MSDeformAttn encoder layers built at open_set/models/mask2former_head.py:112-117):
 *   y = LayerNorm(x + W2 relu(W1 x + b1) + b2);  x16 (M, 256) bf16 rows, w1 (F x 256) / w2 (256 x F) packed by
 *   cgg_linear_rows_pack, F % 256 == 0. Outputs (each nullable except one of y16 / y32): y16 = bf16(y),
 *   yp16 = bf16(y + pos[row % pos_rows]), y32 = y. The (M x F) hidden activation never reaches HBM. */
int cgg_encoder_ffn_ln_bf16(const void* x16, const void* w1_packed, const float* b1, const void* w2_packed,
                            const float* b2, const float* gamma, const float* beta, float eps, const float* pos,
                            int pos_rows, void* y16, void* yp16, float* y32, int M, int C, int F, cgg_stream_t stream);
/* Same block for the LAST encoder layer: the LayerNorm also emits the query decoder's K / V operands exactly as
 * cgg_add_layernorm_kv does (m16 = bf16(y + shift[s]), mp16 = bf16(y + shift[s] + pos[s]), level-major rows); y32 (nullable)
 * is the f32 memory at its natural rows. M = B * S rows, level_start_host[n_levels] ascending from 0. */
int cgg_encoder_ffn_ln_kv_bf16(const void* x16, const void* w1_packed, const float* b1, const void* w2_packed,
                               const float* b2, const float* gamma, const float* beta, float eps, const float* shift,
                               const float* pos, int S, const int* level_start_host, int n_levels, float* y32, void* m16,
                               void* mp16, int M, int C, int F, cgg_stream_t stream);
/* The whole post-attention half of an encoder layer as ONE launch ('self_attn' output projection + 'norm' + 'ffn' + 'norm' of
 * the [3P] BaseTransformerLayer, open_set/models/mask2former_head.py:112-117):
 *   x1 = LayerNorm0(x + a Wo^T + bo);   y = LayerNorm1(x1 + W2 relu(W1 x1 + b1) + b2)
 * a16 = attention rows (MSDeformAttn output before output_proj), x16 = layer input rows, both (M, 256) bf16; wo / w1 / w2 packed by
 * cgg_linear_rows_pack. x1 and the hidden activation never reach memory. Outputs as cgg_encoder_ffn_ln_bf16 (shift == NULL:
 * y16 / yp16 / y32, pos nullable) or as cgg_encoder_ffn_ln_kv_bf16 (shift != NULL: pos_rows = S, y16 = m16, yp16 = mp16,
 * level-major, level_start_host[n_levels]). */
int cgg_encoder_layer_tail_bf16(const void* a16, const void* x16, const void* wo_packed, const float* bo, const float* gamma0,
                                const float* beta0, float eps0, const void* w1_packed, const float* b1, const void* w2_packed,
                                const float* b2, const float* gamma1, const float* beta1, float eps1, const float* pos,
                                int pos_rows, const float* shift, const int* level_start_host, int n_levels, void* y16,
                                void* yp16, float* y32, int M, int C, int F, cgg_stream_t stream);

/* Last encoder layer of the stream: y = LN(a + b) as above (y32 nullable) plus the two bf16 operands of the query
 * decoder's K / V projections (mask2former_head.py:795-812), written LEVEL-MAJOR -- [level][batch][hw_l][256], level l =
 * rows level_start[l] .. level_start[l+1] of each batch's S rows -- so that every level is one contiguous GEMM operand:
 *   m16 = bf16(y + shift[s]),  mp16 = bf16((y + shift[s]) + pos[s]),  s = row % S;  shift, pos: [S, 256] f32.
 * level_start_host: n_levels ints on the HOST (start[0] == 0, increasing, < S).                                   */
int cgg_add_layernorm_kv(const void* a, int a_dtype, const void* b, int b_dtype, const float* gamma, const float* beta,
                         const float* shift, const float* pos, int S, const int* level_start_host, int n_levels,
                         float* y32, void* m16, void* mp16, int rows, int N, float eps, cgg_stream_t stream);

/* Throughput-mode (bf16, channel-last) variants used by the pixel decoder's inference stream.
 * cgg_group_norm_nhwc: GroupNorm over x [B, HW, C] bf16 with C / groups == 8 ([3P] MSDeformAttnPixelDecoder
 *   input_convs / lateral_convs / output_convs, norm_cfg GN-32), y = (x - mean) * rstd * gamma + beta
 *   (+ bilinear up-sample of up_src [B, up_h, up_w, C] f32 to the (HW / W, W) grid -- the FPN `cur +
 *   F.interpolate(outs[-1])` step) (+ ReLU); outputs (each nullable, at least one): y32 f32 with batch stride
 *   y32_bstride elements (writes straight into the (B, N, C) encoder stream), y16 = bf16(y) and yp16 =
 *   bf16(y + pos), pos [HW, C] f32, both with batch stride y16_bstride elements. ws:
 *   cgg_group_norm_nhwc_workspace_bytes(B, HW, groups) bytes of scratch (per-block partial sums, no atomics).
 * cgg_pack_mask_feature_nhwc: cgg_pack_mask_feature for a [B, H, W, C] bf16 map (hi image only).              */
int64_t cgg_group_norm_nhwc_workspace_bytes(int B, int HW, int groups);
int cgg_group_norm_nhwc(const void* x, const float* gamma, const float* beta, void* ws, int B, int HW, int C,
                        int groups, float eps, int relu, const float* up_src, int up_h, int up_w,
                        int64_t up_bstride, int W, float* y32, int64_t y32_bstride, void* y16, const float* pos,
                        void* yp16, int64_t y16_bstride, cgg_stream_t stream);
int cgg_pack_mask_feature_nhwc(const void* feat, void* hi, int B, int C, int H, int W, int pool,
                               cgg_stream_t stream);
/* Same, up to 4 packed images from one launch (the full-resolution image and the pooled ones of the decoder levels):
 * hi_host[i] receives the pool = pools_host[i] image; both arrays (n <= 4 entries) live on the HOST.              */
int cgg_pack_mask_feature_nhwc_multi(const void* feat, void* const* hi_host, const int* pools_host, int n, int B, int C,
                                     int H, int W, cgg_stream_t stream);

/* f3  Channel-last epilogue of the BN-folded backbone convolutions ([3P] mmdet ResNet Bottleneck tail
 * `relu(bn3(conv3(x)) + identity)`, selected by configs/instance/coco_b48n17.py:17-26), in place:
 *   y[rows, C] bf16 <- act(y + bias[C] + res[rows, C]);  bias, res bf16, nullable;  relu != 0 -> max(., 0).
 * Requires C % 8 == 0 and 16-byte aligned pointers.                                                             */
int cgg_bias_act_nhwc(void* y, const void* bias, const void* res, int64_t rows, int C, int relu,
                      cgg_stream_t stream);

/* Library GEMM with the residual epilogue ATen does not expose, for `relu(conv3(x) + identity)` of a BN-folded
 * [3P] mmdet ResNet Bottleneck (1x1 convolution on channel-last bf16 rows):
 *   y[M, N] = act(x[M, K] w[N, K]^T + bias[N] + res[M, N])   (bf16 in / out, f32 accumulation; bias, res nullable; relu 0/1)
 * = ONE hipBLASLt matmul (beta = 1, HIPBLASLT_EPILOGUE_RELU_BIAS). cgg_blaslt_init(path) dlopens the hipBLASLt copy the
 * process already uses (NULL / "" = "libhipblaslt.so") and must be called once before.                              */
int cgg_blaslt_init(const char* libpath);
/* cgg_blaslt_set_tuning(n > 1): at the first use of a shape outside a stream capture, time the first n heuristic
 * candidates (2 warm + 5 timed launches each) and keep the fastest; cgg_blaslt_last_tuning reports (top-1, chosen) us. */
int cgg_blaslt_set_tuning(int n_candidates);
int cgg_blaslt_last_tuning(float* top1_us, float* chosen_us);
int cgg_gemm_bias_res_act_bf16(const void* x, const void* w, const void* bias, const void* res, void* y, int M, int N,
                               int K, int relu, cgg_stream_t stream);

/* Patch matrix of a 3x3 / padding 1 / stride 1|2 convolution on a channel-last bf16 map x[B, H, W, C]:
 * y[B * Ho * Wo, 9 * C], column (ky * 3 + kx) * C + c = x[b, oy * stride + ky - 1, ox * stride + kx - 1, c] (0 outside),
 * Ho = (H - 1) / stride + 1. With cgg_gemm_bias_res_act_bf16 it replaces the 3x3 convolutions of the deep stages of the
 * BN-folded [3P] mmdet ResNet (conv2 of the Bottlenecks of layer3 / layer4).                                        */
int cgg_im2col3x3_nhwc(const void* x, void* y, int B, int H, int W, int C, int stride, cgg_stream_t stream);

/* HOST function (no GPU work): rectangular linear sum assignment of n_problems f32 cost matrices (row-major, concatenated;
 * problem p is nr[p] x nc[p]) -- the Hungarian matching of open_set/assigners/mask_hungarian_assigner.py:126-131, which the
 * reference solves one matrix at a time with scipy.optimize.linear_sum_assignment. Same algorithm (Crouse 2016), scan order
 * and tie rule as scipy, so the INDICES agree, not only the cost. rows / cols: min(nr[p], nc[p]) pairs per problem,
 * concatenated, rows ascending. CGG_EINVAL for NaN / -inf entries, CGG_EUNSUPPORTED for an infeasible matrix.         */
int cgg_linear_sum_assignment_f32(const float* cost, int n_problems, const int* nr, const int* nc, int64_t* rows,
                                  int64_t* cols);

/* [3P] mmcv.ops.point_sample (= F.grid_sample(2 p - 1, bilinear, zeros, align_corners=False)) on a CHANNEL-LAST f32 map:
 * out[b, p, :] = bilinear sample of feat[b] (H, W, C) at pts[b, p] = (x, y) in [0, 1]; taps outside the map contribute 0.
 * Used for sample(E F) = E sample(F) at the matching points of all decoder layers (mask2former_head.py:357-366). C % 4 == 0. */
int cgg_point_sample_nhwc(const float* feat, const float* pts, float* out, int B, int H, int W, int C, int P,
                          cgg_stream_t stream);
/* Same sampling of single-channel f32 planes [N, H, W] with a plane index per output row: out[j, p] = sample of
 * planes[index[j]] at pts[j, p] (pts [rows, P, 2], out [rows, P]). Used for the matched queries' ground-truth mask targets
 * (mask2former_head.py:609-612): one launch per decoder layer for all images.                                          */
int cgg_point_sample_planes(const float* planes, const int32_t* index, const float* pts, float* out, int N, int H, int W,
                            int rows, int P, cgg_stream_t stream);
/* Backward of cgg_point_sample_planes wrt the planes (points are constants): grad_planes (N, H, W), ZEROED by the caller,
 * accumulates the tap weights x grad_out (rows, P) with f32 atomics; index nullable = row j samples plane j. Replaces
 * F.grid_sample's backward (which also computes the unused grid gradient) at open_set/models/mask2former_head.py:609-620. */
int cgg_point_sample_planes_backward(const float* grad_out, const int32_t* index, const float* pts, float* grad_planes, int N,
                                     int H, int W, int rows, int P, cgg_stream_t stream);
/* ... for rows that each have their OWN plane (the loss' mask planes of the positives): grad_planes (rows, H, W) is overwritten -- no
 * zero-fill, no global atomics (a workgroup accumulates a band of one plane in LDS and writes it once). */
int cgg_point_sample_planes_backward_rows(const float* grad_out, const float* pts, float* grad_planes, int H, int W, int rows, int P,
                                          cgg_stream_t stream);

/* Stem convolution of the BN-folded [3P] mmdet ResNet (conv1: 7x7, stride 2, padding 3, 3 -> 64 channels) straight from
 * the f32 NCHW image: out[B, Ho, Wo, 64] bf16 channel-last = RAW convolution (no bias; bf16 operands, f32 accumulation),
 * Ho = (H - 1) / 2 + 1. w_packed: cgg_stem_conv7x7_packed_bytes() bytes, bf16 MFMA A fragments
 * [2 m-tiles][11 k-steps][64 lanes][8] of W[64, 176] with k = ky * 24 + kx * 3 + c (zero elsewhere).              */
int64_t cgg_stem_conv7x7_packed_bytes(void);
int cgg_stem_conv7x7_nchw(const float* img, const void* w_packed, void* out, int B, int H, int W, cgg_stream_t stream);

/* y[B, Ho, Wo, C] = x[B, ::stride, ::stride, C] (channel-last bf16, Ho = (H - 1) / stride + 1): the input of a stride-2 1x1
 * downsample convolution ([3P] mmdet ResNet, style='pytorch') as a contiguous GEMM operand.                           */
int cgg_subsample_nhwc(const void* x, void* y, int B, int H, int W, int C, int stride, cgg_stream_t stream);

/* Stem tail of the BN-folded [3P] mmdet ResNet (`maxpool(relu(bn1(conv1(x))))`), channel-last bf16, one pass:
 *   y[B, Ho, Wo, C] = relu(maxpool3x3/s2/p1(x[B, H, W, C]) + bias[C]),  Ho = (H - 1) / 2 + 1 (same for W).          */
int cgg_bias_relu_maxpool_nhwc(const void* x, const void* bias, void* y, int B, int H, int W, int C,
                               cgg_stream_t stream);

/* K9  GroupNorm (+ optional ReLU) of the pixel decoder ConvModules ([3P] MSDeformAttnPixelDecoder,
 * norm_cfg=dict(type='GN', num_groups=32), configs/instance/coco_b48n17.py:40).
 *   x, y [B, C, H, W] f32 NCHW; gamma, beta [C]; ws = cgg_group_norm_workspace_bytes(...) bytes.
 * Requires C % groups == 0 (float4 path when H*W % 4 == 0).                                                    */
int64_t cgg_group_norm_workspace_bytes(int B, int C, int H, int W, int groups);
int cgg_group_norm(const float* x, const float* gamma, const float* beta, float* y, void* ws, int B, int C,
                   int H, int W, int groups, float eps, int relu, cgg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K19  Inference tail.
 *
 * cgg_upsample_bilinear: F.interpolate(x, (h, w), 'bilinear', align_corners=False) for
 * [N, H, W] -> [N, h, w] f32 (open_set/models/mask2former_head.py:960-964,
 * open_set/models/maskformer_fusion_head.py:421-425).
 *
 * cgg_instance_masks: fused "upsample -> (>0) -> mask score -> bbox" of
 * MaskFormerFusionHeadOpen.instance_postprocess_emb (maskformer_fusion_head.py:349-363) for the
 * selected queries, reading the LOW-RES logits, so the (Q, H_img, W_img) f32 tensor is never
 * written:
 *   logits [Q, H, W] f32;  sel [n] int32 query index per detection
 *   (up_h, up_w): batch_input_shape the head upsamples to (mask2former_head.py:957-964)
 *   (crop_h, crop_w): img_shape crop (maskformer_fusion_head.py:415-416)
 *   (out_h, out_w): ori_shape when `rescale` (:418-425), else == crop
 *   masks  [n, out_h, out_w] u8 (0/1);  mask_score [n] f32 = sum(sigmoid*[m>0]) / (sum[m>0]+1e-6);
 *   bbox [n, 4] f32 (x0, y0, x1+1, y1+1), zeros for an empty mask ([3P] mmdet mask2bbox)
 *   ws     scratch, n * 32 bytes
 *
 * cgg_panoptic_argmax: per-pixel argmax_k score[k] * sigmoid(resized logit[keep[k]])
 * (maskformer_fusion_head.py:100-120) fused with the resize; first max wins (torch.argmax).
 *   ids [out_h, out_w] int32 in [0, n);  win_half [out_h, out_w] u8 = winner's sigmoid >= 0.5
 *   counts [n, 3] int32: #(id == k), #(sigmoid_k >= 0.5), #(id == k && sigmoid_k >= 0.5)
 *   -- the three areas the segment loop of :122-157 needs (one D2H instead of 3 .item() per query)
 * cgg_panoptic_paint: seg[p] = lut_val[ids[p]] if lut_val >= 0 and (!lut_half[id] || win_half[p])
 *   else void_label  (the decisions of :129-157 applied in one pass).
 * ---------------------------------------------------------------------------------------------- */
int cgg_upsample_bilinear(const float* x, float* y, int N, int H, int W, int h, int w,
                          cgg_stream_t stream);
int cgg_instance_masks(const float* logits, const int32_t* sel, uint8_t* masks, float* mask_score,
                       float* bbox, void* ws, int Q, int H, int W, int up_h, int up_w, int crop_h,
                       int crop_w, int out_h, int out_w, int n, cgg_stream_t stream);
/* Multi-destination variant for the three evaluation types of one image (all / novel / base class sets,
 * maskformer_fusion_head.py:385-400): instance i is QUERY i (no `sel`), its mask is interpolated once and stored to every
 * slot dest_slot[dest_off[i] .. dest_off[i+1]) of masks [n_slots, out_h, out_w]; queries with no slot are skipped.
 * mask_score [Q], bbox [Q, 4] are per query (undefined for skipped queries); ws: Q * 32 bytes.                    */
int cgg_instance_masks_multi(const float* logits, const int32_t* dest_off, const int32_t* dest_slot, uint8_t* masks,
                             float* mask_score, float* bbox, void* ws, int Q, int H, int W, int up_h, int up_w,
                             int crop_h, int crop_w, int out_h, int out_w, cgg_stream_t stream);

/* Open-vocabulary (query, class) picks of MaskFormerFusionHeadOpen.instance_postprocess_emb
 * (open_set/models/maskformer_fusion_head.py:297-347) for all evaluation types and images in one launch.
 * dots [B*Q, ld] f32 = class-embedding predictions x concatenated class tables; type t owns columns
 * col0[t] .. col0[t] + ncols[t], the last one being the background class: prob = softmax over the type's columns,
 * background dropped, then the k best of the Q * (ncols[t] - 1) pairs. Outputs [B, n_types, k]: labels (class),
 * scores (class probability), qidx (query). topk(sorted=False) leaves order / ties open; here: descending score, ties
 * by ascending flat index. col0_host / ncols_host: n_types ints on the HOST. Limits: k <= 1024, n_types <= 8,
 * k <= Q * (ncols-1), 8k + 4Q * max(ncols-1) + 4Q <= 62 KiB of LDS (CGG_EUNSUPPORTED otherwise).                 */
int cgg_class_topk(const float* dots, int ld, int B, int Q, int n_types, const int* col0_host, const int* ncols_host,
                   int k, int64_t* labels, float* scores, int64_t* qidx, cgg_stream_t stream);

/* The rest of :349-363 for ONE image and all its picks (all evaluation types concatenated): every picked query's mask
 * is interpolated once and stored into each detection slot that picked it (as cgg_instance_masks_multi, the slot plan
 * now built on the device), then bboxes[j] = (box of query qidx[j], cls_scores[j] * mask_score[qidx[j]]).
 * masks [n_picks, out_h, out_w] u8, bboxes [n_picks, 5] f32, ws: cgg_instance_masks_picks_workspace_bytes(Q, n_picks).
 * bitpack != 0: masks [n_picks, out_h, out_w / 8] bytes, pixel x = bit (x & 7) of byte x >> 3 -- 8x fewer bytes for a
 * host-side consumer to copy (integer up-scale 2 / 4 / 8 without a second resize, out_w % 16 == 0; else CGG_EUNSUPPORTED). */
int64_t cgg_instance_masks_picks_workspace_bytes(int Q, int n_picks);
int cgg_instance_masks_picks(const float* logits, const int64_t* qidx, const float* cls_scores, int n_picks,
                             uint8_t* masks, float* bboxes, void* ws, int Q, int H, int W, int up_h, int up_w,
                             int crop_h, int crop_w, int out_h, int out_w, int bitpack, cgg_stream_t stream);
int cgg_panoptic_argmax(const float* logits, const int32_t* keep, const float* score, int32_t* ids,
                        uint8_t* win_half, int32_t* counts, int Q, int H, int W, int up_h, int up_w,
                        int crop_h, int crop_w, int out_h, int out_w, int n, cgg_stream_t stream);
int cgg_panoptic_paint(const int32_t* ids, const uint8_t* win_half, const int32_t* lut_val,
                       const int32_t* lut_half, int32_t* seg, int64_t npix, int void_label,
                       cgg_stream_t stream);

/* Row-wise softmax + max/argmax: scores = softmax(x) over the last dim, one wavefront per row
 * (MaskFormerFusionHeadOpen.get_cls_emb_scores, maskformer_fusion_head.py:297-315, and the
 * `.max(-1)` at :99).  x [rows, n] f32 -> prob [rows, n] f32 (nullable), maxv [rows] f32,
 * argmax [rows] int64 (first max wins).                                                          */
int cgg_rowwise_softmax_argmax(const float* x, float* prob, float* maxv, int64_t* argmax, int rows,
                               int n, cgg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CGG_HIP_H_ */
