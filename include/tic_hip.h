/* tic_hip.h -- C ABI of libtic_hip.so: the MI355X (gfx950) hot path of the TIC ViT fine-tune step.
 *
 * The reference (fAKe2004/TouhouImageClassification) has no FFI: its hot path is the chain of
 * torch/cuBLAS/cuDNN kernels reached from `model(x)` / `loss.backward()` / `optimizer.step()`
 * (TIC/ViT/finetune.py:54-67, TIC/ViT/ntrain.py:43-50).  Each entry point below replaces one of
 * those vendor-kernel families; the citation names the reference-side call it stands in for
 * (HF = transformers/models/vit/modeling_vit.py, the library TIC/ViT/model.py:2,27-45 delegates to).
 *
 * Conventions
 *   - plain pointers and sizes only; the CALLER owns every buffer (the library never allocates or
 *     frees device memory).  Process-wide mutable state: the thread-local last-error string, the
 *     tuning knobs of tic_set_option (A/B measurements; set them before, not during, concurrent
 *     use from several threads) and the event table of tic_kernel_timer_* (one user at a time);
 *   - every call enqueues asynchronously on `stream` (a hipStream_t; pass torch's current stream)
 *     and returns 0, or a negative TIC_E* code with tic_last_error_string() set;
 *   - bf16 tensors are raw 16-bit storage (`void*`), fp32 tensors `float*`, all contiguous
 *     row-major, 16-byte aligned; "tokens" matrices are [M, D] with M = B * N (N = 197).
 */
#ifndef TIC_HIP_H
#define TIC_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* tic_stream_t; /* hipStream_t */

#define TIC_OK 0
#define TIC_EINVAL (-1)  /* bad shape / alignment / null pointer */
#define TIC_ELAUNCH (-2) /* HIP launch error */

#define TIC_ABI_VERSION 1
int tic_version(void);
const char* tic_last_error_string(void);
/* Route selection for the parity tests -- SIX knobs, never needed for correct results: each chooses between numerically equivalent
 * forms of a kernel so that a test can force every one of them.  They are plain process-wide ints read when a launch is enqueued:
 * set them while no other host thread is inside a tic_* call (they are NOT per-stream or per-call state; the default of each is what
 * the training step uses).  Unknown names / values return TIC_EINVAL.
 *   "gemm_tile"  0 (auto) | 128 | 256       NT / TN tile family (auto: 256x256 from 128 tiles up)
 *   "gemm_split" -1 (auto) | 0 | 2 | 4      split-K form of the 128x128 NT kernel (needs tic_gemm_nt_scratch; never under stream capture);
 *                                           auto: two parts where both still get a CU of their own (2 x tiles <= 256) and a part keeps >= 16
 *                                           K tiles.  (Launches of <= 256 workgroups of that kernel, split or not, run its 4-stage LDS-DMA
 *                                           ring instead of the 2-stage loop: same products in the same order, bit-identical results.)
 *   "tn_streamk" 1 (256 shares) | 0 | n     stream-K split of the grouped dW launch
 *   "tn_parts"   -1 (auto) | 0 | 2..8       grouped dW whose tile count has no phase-aligned split (ViT-B: 108 tiles): every tile in n equal
 *                                           row parts (auto: 256 / tiles) | 0: flat stream-K
 *   "tn_mfma"    0 (auto) | 16 | 32         MFMA shape of the grouped dW stream-K launch (16x16x32 for long reductions, else 32x32x16)
 *   "stream_nt"  bit mask, default 13       non-temporal cache policy: 1 LayerNorm, 2 AdamW, 4 GEMM epilogue stores, 8 epilogue operand loads
 * Everything that was only ever an A/B measurement (tile-walk shapes, grid caps, stagger, the 256x256 kernel's split-K form, main
 * loops with parts compiled out, in-kernel stage stamps) exists in the measurement library libtic_hip_dbg.so alone (-DTIC_MEASURE,
 * `python -m touhouimageclassification_amd.build dbg`; tools/ab_step.py, tools/gemm_dbg.py, tools/tile_timeline.py). */
int tic_set_option(const char* name, int value);
/* Live timing of the step's dominant kernel (the grouped dW launch of tic_gemm_tn_group_bf16 / tic_vit_backward_layer): while
 * enabled, HIP events are recorded on the launch stream around every such launch (up to 8192); read() waits for them and
 * returns how many launches were timed and their summed duration.  bench.py's roofline.achieved comes from this. */
/* measurement only (tools/hbm_probe.py): streaming read (or copy, dst != NULL) of n floats; mode bit 0 = non-temporal accesses */
int tic_probe_stream(const float* src, float* dst, float* sink, long n, int blocks, int mode, tic_stream_t stream);
int tic_kernel_timer_enable(int on);
int tic_kernel_timer_read(int* launches, float* total_ms);

/* Scratch for the split-K form of the NT kernel (launches with few output tiles and a long reduction: 8-16 images per GPU):
 * caller-owned, at least TIC_NT_SCRATCH_BYTES, 16-byte aligned; its last 4 KiB are flag words the caller zeroes ONCE.  Registered per
 * host thread and used by every tic_gemm_nt_* / tic_vit_* call of that thread until replaced (NULL: never split); calls that share a
 * scratch must be on ONE stream.  tic_vit_* use the scratch inside their workspace (TicVitLayout.nt_scratch) by themselves.
 * A consumer workgroup whose bounded poll for a producer's flag runs out does not pass silently: it stores a code into a host-mapped
 * error word (64 bytes of pinned host memory the library maps on the first registration -- its only allocation) and EVERY later tic_*
 * call returns TIC_ELAUNCH until a scratch is registered again.  Launches enqueued while the stream is being captured into a HIP
 * graph never split (the flag value counts launches and would be replayed). */
#define TIC_NT_SCRATCH_BYTES ((size_t)192 * 32 * 512 * 16 + 4096)
int tic_gemm_nt_scratch(void* scratch, size_t bytes);

/* GEMM epilogues (fused into the MFMA kernel's store) */
#define TIC_EPI_BF16 0  /* out = bf16(acc + bias)                                   Linear            */
#define TIC_EPI_GELU 1  /* out = u = bf16(acc + bias); out2 = bf16(gelu_erf(u))     fc1 + GELU        */
#define TIC_EPI_RESID 2 /* out_f32 = resid + bf16(acc + bias)                       o_proj / fc2 + residual */
#define TIC_EPI_DGELU 3 /* out = bf16(bf16(acc) * gelu'(aux))                       backward through GELU */
#define TIC_EPI_PATCH 4 /* out_f32[(m/P)*(P+1)+1+m%P] = bf16(acc+bias) + rowtab[1+m%P]   patch embed + pos */
#define TIC_EPI_GELU_DG 5 /* u = bf16(acc + bias); out = bf16(gelu'(u)); out2 = bf16(gelu_erf(u))   fc1 + GELU, derivative saved
                             instead of the pre-activation: the erf / exp work is shared and the backward becomes one multiply */
#define TIC_EPI_MULAUX 6  /* out = bf16(bf16(acc) * aux)                               backward through GELU with aux = gelu'(u) */
#define TIC_EPI_GELU_ONLY 8 /* out2 = bf16(gelu_erf(bf16(acc + bias))), nothing else stored     fc1 + GELU in a forward that no backward follows */
#define TIC_EPI_ADDAUX 7  /* out = bf16(bf16(acc) + aux), aux bf16 [M, N]; out == aux allowed (in place)   ResNet: input gradient of a 1x1
                           * convolution added to the gradient of the block's identity branch (TIC/ResNet/model.py:113, out += identity) */

/* C[M,N] = A[M,K] . B[N,K]^T with a fused epilogue.  N % 8 == 0, K % 64 == 0, any M >= 1.
 * Replaces nn.Linear forward (HF:202-205,216-218,235-236,246-247,250-252) and, fed with W^T, the dX half
 * of its backward; EPI_PATCH replaces the patch Conv2d + position add (HF:60,69,146-157). */
int tic_gemm_nt_bf16(const void* A, const void* B, int M, int N, int K, int epilogue, const float* bias,
                     void* out_bf16, void* out2_bf16, float* out_f32, const float* resid,
                     const void* aux_bf16, const float* rowtab, int patches, tic_stream_t stream);

/* same, plus colsum (optional, fp32 [N]) += column sums of the stored output (EPI_BF16 / EPI_DGELU / EPI_MULAUX): the bias gradient
 * of the Linear whose output gradient this GEMM produces, fused into the epilogue */
int tic_gemm_nt_bf16_ex(const void* A, const void* B, int M, int N, int K, int epilogue, const float* bias,
                        void* out_bf16, void* out2_bf16, float* out_f32, const float* resid,
                        const void* aux_bf16, const float* rowtab, int patches, float* colsum, tic_stream_t stream);

/* C[N,K] += A[M,N]^T . B[M,K]  (fp32 accumulate into C; C holds the running gradient).
 * N % 8 == 0, K % 8 == 0.  Replaces the dW half of nn.Linear backward (autograd, finetune.py:62). */
int tic_gemm_tn_bf16(const void* A, const void* B, float* C, int M, int N, int K, tic_stream_t stream);
/* the same for up to 4 problems sharing M (the four Linear layers of a transformer block) in ONE launch of
 * full-reduction 256x256 tiles (no split-K, no atomics) when shapes allow, else one launch per problem */
int tic_gemm_tn_group_bf16(int nprob, const void* const* A, const void* const* B, float* const* C, const int* N,
                           const int* K, int M, tic_stream_t stream);
/* overwrite != 0: C_g = A_g^T . B_g -- what C held is dropped and need not be initialised (the first backward after the gradients were
 * cleared: no read of C, no zeroing pass).  Routes that add partial tiles (stream-K shares, row parts) zero C themselves first. */
int tic_gemm_tn_group_bf16_ex(int nprob, const void* const* A, const void* const* B, float* const* C, const int* N,
                              const int* K, int M, int overwrite, tic_stream_t stream);

/* LayerNorm(eps) over the last dim of fp32 rows -> bf16; saves mean / rstd.  in_stride = elements
 * between consecutive input rows (D for the token stream, N*D to pick the CLS rows).  HF:261-262,274,281,385. */
int tic_layernorm_fwd(const float* x, long in_stride, const float* gamma, const float* beta, void* y_bf16,
                      float* mean, float* rstd, int rows, int D, float eps, tic_stream_t stream);
/* dx = (dres ? dres : 0) + LN'(dy); also writes bf16(dx) to dxb (optional); dgamma/dbeta += . */
int tic_layernorm_bwd(const void* dy_bf16, const float* x, long stride, const float* gamma, const float* mean,
                      const float* rstd, const float* dres, float* dx, void* dxb_bf16, float* dgamma,
                      float* dbeta, int rows, int D, tic_stream_t stream);

/* + colsum (optional, fp32 [D]) += column sums of dx */
int tic_layernorm_bwd_ex(const void* dy_bf16, const float* x, long stride, const float* gamma, const float* mean,
                         const float* rstd, const float* dres, float* dx, void* dxb_bf16, float* dgamma,
                         float* dbeta, float* colsum, int rows, int D, tic_stream_t stream);

/* softmax(q k^T * scale) v per (image, head); qkv packed [M, 3D], head_dim 64, N <= 208.
 * Replaces F.scaled_dot_product_attention (HF:220-233) and its backward. lse: [B*H, N] fp32. */
int tic_attention_fwd(const void* qkv, void* o, float* lse, int B, int H, int N, float scale, tic_stream_t stream);
int tic_attention_bwd(const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv, int B,
                      int H, int N, float scale, tic_stream_t stream);

/* + dbias (optional, fp32 [3D]) += column sums of dqkv (gradient of the fused q/k/v bias) */
int tic_attention_bwd_ex(const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv, float* dbias, int B,
                         int H, int N, float scale, tic_stream_t stream);
/* same with a caller-owned fp32 scratch [B][3*H*64]: each (image, head) stores its 192 column sums there and a second tiny
 * kernel adds the images into dbias -- no global atomics (they cost 25 % of the backward kernel at 5 312 workgroups) */
int tic_attention_bwd_ws(const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv, float* dbias, float* scratch_b3d,
                         int skip_v_bias, int B, int H, int N, float scale, tic_stream_t stream);
/* skip_v_bias != 0 leaves dbias[2D..3D) alone: the gradient of the v bias equals the column sums of d_o (dV = P^T dO and the rows of
 * P sum to 1), which the GEMM producing d_o adds for free through its colsum epilogue */

/* pixel_values [B,C,img,img] fp32 -> patch matrix [B*(img/patch)^2, C*patch*patch] bf16 (HF:60,69) */
int tic_patchify(const float* x, void* P_bf16, int B, int C, int img, int patch, tic_stream_t stream);
/* h[b,0,:] = cls + pos[0,:]  (HF:146-157);  backward: dcls, dpos += sum_b dh */
int tic_embed_cls(const float* cls, const float* pos, float* h, int B, int N, int D, tic_stream_t stream);
int tic_embed_bwd(const float* dh, float* dcls, float* dpos, int B, int N, int D, tic_stream_t stream);
/* out_bf16[b*(N-1)+p, :] = bf16(dh[b, 1+p, :]) : token-stream gradient -> patch rows */
int tic_gather_patch_rows(const float* dh, void* out_bf16, int B, int N, int D, tic_stream_t stream);

/* out[n] += sum_m in[m,n]  (bias gradients) */
int tic_colsum_bf16(const void* in_bf16, float* out, int M, int N, tic_stream_t stream);
/* fp32 -> bf16 (n % 4 == 0) and fp32 [R,C] -> bf16 [C,R] (R, C % 64 == 0): autocast's weight casts */
int tic_cast_bf16(const float* in, void* out_bf16, long n, tic_stream_t stream);
int tic_cast_transpose_bf16(const float* in, void* out_bf16, int R, int C, tic_stream_t stream);

/* torch.optim.AdamW step over a flat fp32 buffer (n % 4 == 0), step is 1-based; optional bf16 shadow.
 * Replaces optimizer.step() with AdamW(lr, weight_decay) on every parameter (ntrain.py:39-41, finetune.py:314). */
int tic_adamw(float* p, const float* g, float* m, float* v, void* w16_or_null, long n, float lr, float beta1,
              float beta2, float eps, float weight_decay, int step, tic_stream_t stream);

/* classifier on the CLS rows (HF:559-561): logits fp32 (bf16-rounded) = z . W^T + b ; and its backward */
int tic_head_fwd(const void* z_bf16, const float* W, const float* bias, float* logits, int B, int C, int D,
                 tic_stream_t stream);
int tic_head_bwd(const float* dlogits, const void* z_bf16, const float* W, void* dz_bf16, float* dW, float* db,
                 int B, int C, int D, tic_stream_t stream);

/* mean cross-entropy; labels (int64 [B]) XOR soft ([B,C] fp32).  *loss_sum += loss (zero it first);
 * dlogits (optional) = gscale * dloss/dlogits.  finetune.py:61,315; ntrain.py:48,55. */
int tic_softmax_xent(const float* logits, const int64_t* labels, const float* soft, float* loss_sum,
                     float* dlogits, int B, int C, float gscale, tic_stream_t stream);

/* On-GPU augmentation (replaces the CPU DataLoader transforms of ntrain.py:93-136, torchvision v2):
 * images [B,Hs,Ws,3] uint8 -> out [B,3,S,S] fp32 normalised; params [B,20] fp32 per image:
 *   0 top 1 left 2 h 3 w (crop box) 4 flip 5..8 ColorJitter op order (0 brightness 1 contrast 2 saturation 3 hue)
 *   9 brightness 10 contrast 11 saturation 12 hue 13 jitter-on 14 grayscale-on 15 erase-on 16..19 erase box (i,j,h,w).
 * mean3/std3 are HOST pointers to 3 floats (ImageNet stats, ntrain.py:111; dataset stats, preprocess.py:61-77). */
#define TIC_AUG_NPARAM 20
int tic_augment(const void* images_u8, int B, int Hs, int Ws, const float* params, float* out, int S,
                const float* mean3, const float* std3, tic_stream_t stream);
/* MixUp (mode 0: out = lam x + (1-lam) roll(x,1,0)) / CutMix (mode 1: box [y1,y2)x[x1,x2) pasted from roll(x,1,0));
 * out-of-place.  ntrain.py:30-33,45-46 (torchvision v2.MixUp / v2.CutMix, alpha = 1). */
int tic_mix(const float* x, float* out, int B, int C, int H, int W, int mode, float lam, int x1, int y1, int x2,
            int y2, tic_stream_t stream);
/* soft labels [B,ncls] = lam onehot(y) + (1-lam) onehot(roll(y,1)) */
int tic_mix_labels(const int64_t* y, float* out, int B, int ncls, float lam, tic_stream_t stream);

/* ---- ResNet conv path (TIC/ResNet/model.py; activations NHWC bf16, weights master fp32 OIHW) --------------
 * A convolution is a GEMM over the kernels above: Y[M,Co] = col(X)[M,Kp] . Wp[Co,Kp]^T with M = B*Ho*Wo and
 * Kp = kh*kw*Ci rounded up to 64 (tap-major k = (ky*kw+kx)*Ci + c); 1x1 stride-1 convs use X itself as col(X). */
/* transposed = 1 writes [Kp, Co] (the B operand of the explicit dgrad GEMM); transposed = 2 writes the implicit-GEMM dgrad
 * filter [Ci][(ky',kx')*Co + o] = w[o][c][kh-1-ky'][kw-1-kx']                                   model.py:8-9,14,148 */
/* transposed = 3: the STEM layout [Co, 256] of the implicit 7x7 / 2 stem (Ci = 3, kh = kw = 7): k = ky*32 + px*4 + c over 7 filter rows x
 * 8 pixels x 4 channels, px = kx + 1 (px 0, c = 3 and row 7 are zeros) -- the filter as it meets an image whose 3 channels are padded to 4 */
int tic_conv_weight_pack(const float* w_oihw, void* w16_ohwi, int Co, int Ci, int kh, int kw, int transposed, tic_stream_t stream);
/* Implicit-GEMM convolution (no im2col buffer) for filters with Cin % 64 == 0 -- the 3x3 convolutions of TIC/ResNet/model.py:6-9
 * (Bottleneck conv2 :87, BasicBlock conv1/conv2 :31-36).  x is NHWC bf16, w_pack = tic_conv_weight_pack(..., transposed = 0).
 *   fwd   : y[B*Ho*Wo, Cout] (bf16) = conv(x, w)            -- also the stride-1 input gradient: call it with dY as x, Cin/Cout
 *           swapped, pad = k-1-pad and w_pack = tic_conv_weight_pack(..., transposed = 2) (flipped, channel-transposed filter)
 *   wgrad : dw[Cout, kh*kw*Cin] (fp32, tap-major, += with atomics) = dY^T . gather(x); fold into OIHW with tic_conv_weight_grad */
int tic_conv_igemm_fwd(const void* x_nhwc, const void* w_pack, void* y, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                       int stride, int pad, tic_stream_t stream);
int tic_conv_igemm_wgrad(const void* dy, const void* x_nhwc, float* dw, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                         int stride, int pad, tic_stream_t stream);
/* Input gradient of a stride-2 convolution (3x3 pad 1 -- the first block of a stage, model.py:87 -- or the 1x1 downsample projection, :193-197)
 * without a column buffer / col2im pass: the input pixels of parity class (py, px) = (row & 1, column & 1) see (1 + py)(1 + px) of the 9 taps
 * (k = 3), or the one tap when py = px = 0 (k = 1: the other classes get no gradient), so every class is a stride-1 implicit GEMM over dY
 * [B, H/2, W/2, Cout] whose rows are stored at (2a + py, 2b + px) of dx [B, H, W, Cin].  One call per class.  w_class =
 * tic_conv_weight_pack(..., transposed = 4 + 2 py + px) for k = 3 ([Cin][(ky', kx') * Cout + o] over the class's taps), transposed = 2 for
 * k = 1.  accumulate != 0: dx rows of the class += the product (as TIC_EPI_ADDAUX); rows of other classes are never touched.  H, W even. */
int tic_conv_igemm_dgrad_s2(const void* dy, const void* w_class, void* dx, int B, int H, int W, int Cin, int Cout, int k, int py, int px,
                            int accumulate, tic_stream_t stream);
/* The 7x7 / 2, pad 3 stem (model.py:148) is an implicit GEMM too: pass the image as NHWC with 4 channels per pixel (tic_nchw_to_nhwc_pad_bf16,
 * Cpad = 4: 8-byte pixels, so the aligned 8-pixel window of a filter row is four 16-byte chunks), Cin = 4, kh = kw = 7, stride 2, pad 3, an even
 * W, and w_pack / dw in layout 3 ([Cout, 256]).  No im2col buffer (1.2 GB at 256 images) is written or read. */
int tic_conv_weight_grad(const float* dw_ohwi, float* grad_oihw, int Co, int Ci, int kh, int kw, int layout, tic_stream_t stream); /* grad +=; layout 0 | 3 */
/* optional scratch for few-tile weight gradients (tic_gemm_tn_bf16 with N, K multiples of 256, M >= 8192, <= 64 tiles: the 1 x 1
 * convolutions of a deep ResNet stage): every tile is cut into row parts that STORE their partial tiles into the scratch and one more
 * launch adds them to C, instead of stream-K shares that each add a whole tile with fp32 atomics.  Caller-owned, per host thread, 64 MiB
 * covers every case; pass NULL to unregister.  Without it nothing changes. */
int tic_gemm_tn_scratch(void* scratch, size_t bytes);
/* tic_conv_weight_pack / tic_conv_weight_grad for a whole network in one launch each: `descs` points to n entries in DEVICE memory
 * (the caller builds the table once; entries are consumed by blockIdx.y).  Same arithmetic per entry as the single calls. */
typedef struct {
    const float* w; /* OIHW fp32 */
    void* out;      /* bf16 operand, layout by `transposed` as in tic_conv_weight_pack */
    int Co, Ci, kh, kw, transposed, pad_;
} TicConvPackDesc;
typedef struct {
    const float* dw; /* [Co, Kp] fp32 (tap-major) */
    float* grad;     /* OIHW fp32, += */
    int Co, Ci, kh, kw, layout, pad_; /* layout: 0 tap-major [Co, Kp] | 3 the stem's [Co, 256] */
} TicConvGradDesc;
int tic_conv_weight_pack_many(const TicConvPackDesc* descs, int n, tic_stream_t stream);
int tic_conv_weight_grad_many(const TicConvGradDesc* descs, int n, tic_stream_t stream);
int tic_nchw_to_nhwc_bf16(const float* x, void* out_bf16, int B, int C, int H, int W, tic_stream_t stream);
int tic_nchw_to_nhwc_pad_bf16(const float* x, void* out_bf16, int B, int C, int Cpad, int H, int W, tic_stream_t stream); /* channels C..Cpad-1 = 0 */
int tic_im2col_bf16(const void* x, void* col, int B, int H, int W, int Ci, int kh, int kw, int stride, int pad, tic_stream_t stream);
int tic_col2im_bf16(const void* dcol, void* dx, int B, int H, int W, int Ci, int kh, int kw, int stride, int pad, int accumulate,
                    tic_stream_t stream);
/* BatchNorm2d over [M, C] bf16 (+ optional residual add, + optional ReLU): y = relu(bn(x) + identity).  train != 0: batch
 * statistics, running stats (unbiased var) and num_batches_tracked updated; else running stats.  scratch: caller-owned,
 * >= tic_batchnorm_scratch_bytes(M, C), contents arbitrary on entry and undefined on return -- the row splits of the column
 * reductions STORE their partial sums there and the next kernel adds them in a fixed order (no atomics: statistics, activations and
 * activation gradients are bit-reproducible run to run; no zero-on-entry contract, one buffer may serve every layer of a stream).
 * model.py:51-52,60-61,99-113,150-151 */
size_t tic_batchnorm_scratch_bytes(long M, int C);
int tic_batchnorm_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                      int64_t* num_batches, float* mean, float* rstd, void* scratch, size_t scratch_bytes, const void* identity, void* y,
                      long M, int C, float eps, float momentum, int train, int relu, tic_stream_t stream);
/* dz = dy * [y > 0] (y_or_null = the block output when a ReLU follows); dx = BN'(dz); dskip (optional) (+)= dz; dgamma/dbeta += */
int tic_batchnorm_bwd(const void* dy, const void* y_or_null, const void* x, const float* mean, const float* rstd, const float* gamma,
                      void* scratch, size_t scratch_bytes, void* dx, void* dskip, int skip_accumulate, float* dgamma, float* dbeta, long M,
                      int C, tic_stream_t stream);
/* backward of y = relu(bn(x)) WITHOUT a residual add (the first one or two BatchNorms of a block): the ReLU mask is recomputed from
 * x with the forward's own fp32 expression and bf16 rounding, so y is not read in either pass (12 instead of 16 B per element) */
int tic_batchnorm_bwd_relu(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           void* scratch, size_t scratch_bytes, void* dx, float* dgamma, float* dbeta, long M, int C, tic_stream_t stream);
int tic_maxpool3x3s2_fwd(const void* x, void* y, int B, int H, int W, int C, tic_stream_t stream);                       /* model.py:152 */
int tic_maxpool3x3s2_bwd(const void* x, const void* y, const void* dy, void* dx, int B, int H, int W, int C, tic_stream_t stream);
/* the same pool, with the window position (ky*3 + kx) of the first maximum saved as one byte per output element; the backward then
 * reads only that and dy (no x, no y, no window re-scan) */
int tic_maxpool3x3s2_fwd_idx(const void* x, void* y, void* idx_u8, int B, int H, int W, int C, tic_stream_t stream);
int tic_maxpool3x3s2_bwd_idx(const void* idx_u8, const void* dy, void* dx, int B, int H, int W, int C, tic_stream_t stream);
/* the stem's tail fused (model.py:150-152, bn1 -> relu -> maxpool): y_pool = maxpool3x3/2(relu(bn(x))) + argmax positions without
 * storing relu(bn(x)); bit-identical to tic_batchnorm_fwd(relu) + tic_maxpool3x3s2_fwd_idx.  Backward: tic_maxpool3x3s2_bwd_idx, then
 * tic_batchnorm_bwd_relu on x */
int tic_bn_relu_maxpool_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, int64_t* num_batches,
                            float* mean, float* rstd, void* scratch, size_t scratch_bytes, void* y_pool, void* idx_u8_or_null, int B, int H, int W,
                            int C, float eps, float momentum, int train, tic_stream_t stream);   /* scratch: as tic_batchnorm_fwd, M = B H W */
int tic_avgpool_fwd(const void* x, void* y, int B, int HW, int C, tic_stream_t stream);                                  /* model.py:164,222 */
int tic_avgpool_bwd(const void* dy, void* dx, int B, int HW, int C, tic_stream_t stream);
int tic_add_bf16(void* a, const void* b, long n, tic_stream_t stream); /* a += b */

/* ---------------------------------------------------------------------------------------------
 * Whole-model step: one call enqueues every kernel of a phase (no per-op Python on the hot path).
 * Parameters, gradients, AdamW state and bf16 shadows live in caller-owned FLAT buffers with the
 * canonical layout returned by tic_vit_layout(); activations live in one caller-owned workspace.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int B;      /* images per call (per GPU) */
    int D, H, F, L, C; /* hidden, heads, mlp, layers, classes */
    int img, patch, chans;
    float eps;
} TicVitDims;

typedef struct {
    /* element offsets into the flat fp32 parameter / gradient / AdamW buffers (and the bf16 shadow) */
    long cls, pos, patch_w, patch_b;
    long layer0, layer_stride;                 /* layer l block starts at layer0 + l*layer_stride */
    long ln1_g, ln1_b, wqkv, bqkv, wo, bo, ln2_g, ln2_b, w1, b1, w2, b2; /* offsets inside a layer block */
    long lnf_g, lnf_b, cls_w, cls_b;
    long n_params;                             /* total elements (padded, multiple of 8) */
    /* element offsets into the bf16 transposed-weight buffer */
    long t_layer_stride, t_wqkv, t_wo, t_w1, t_w2, t_total;
    /* byte offsets into the activation workspace */
    size_t P, hs, hs_stride, layer_ws, layer_ws_stride;
    size_t a1, mean1, rstd1, qkv, lse, o, hmid, a2, mean2, rstd2, u, g; /* inside a layer_ws block; u holds gelu'(fc1 out) */
    size_t zf, meanf, rstdf, logits, dlogits, dzf, dh, dhb, dhb2, du, da, dqkv, dpatch;
    size_t nt_scratch;                         /* TIC_NT_SCRATCH_BYTES; the caller zeroes its last 4 KiB when it allocates the workspace */
    size_t ws_bytes;
} TicVitLayout;

int tic_vit_layout(const TicVitDims* dims, TicVitLayout* out);

typedef struct {
    TicVitDims dims;
    float* params;        /* fp32 master weights          [n_params] */
    float* grads;         /* fp32 gradients               [n_params] */
    void* w16;            /* bf16 shadow of params        [n_params] */
    void* wT16;           /* bf16 transposed GEMM weights [t_total]  */
    void* workspace;      /* ws_bytes */
} TicVitState;

/* refresh the bf16 GEMM operands from params (after load_state_dict or an external optimizer step);
 * transposes_only != 0 when w16 is already current (tic_adamw wrote it) and only wT16 needs rebuilding */
int tic_vit_refresh_weights(const TicVitState* st, int transposes_only, tic_stream_t stream);
/* AdamW over the replica's flat buffers (the arithmetic of tic_adamw, bit for bit) that ALSO writes both bf16 operand copies in the same
 * pass: the Linear matrices leave their 64x64 tiles as w16 and, through an LDS transpose, as wT16; biases / LayerNorm / embeddings / head
 * follow the flat rule.  m, v: flat fp32 moment buffers in the parameter layout.  After it tic_vit_refresh_weights has nothing left to do
 * (saves the per-step 1.2 GB + 0.6 GB cast + transpose pass of ViT-L).  optim.FusedAdamW, full fine-tuning. */
int tic_vit_adamw(const TicVitState* st, float* m, float* v, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  tic_stream_t stream);
/* pixel_values [B,3,224,224] fp32 -> logits [B,C] fp32 (also kept in the workspace) */
int tic_vit_forward(const TicVitState* st, const float* pixel_values, float* logits_out, tic_stream_t stream);
/* the same forward when NO backward will follow (validate_step / serve / full_judge under torch.no_grad): fc1 stores only gelu(u),
 * not the derivative the backward would read (TIC_EPI_GELU_ONLY); identical logits */
int tic_vit_forward_infer(const TicVitState* st, const float* pixel_values, float* logits_out, tic_stream_t stream);
/* backward in three phases so the caller can overlap gradient all-reduce per bucket:
 *   head (classifier + final LN), layer l = L-1 .. 0, embeddings.  grads are ACCUMULATED into st->grads. */
int tic_vit_backward_head(const TicVitState* st, const float* dlogits, tic_stream_t stream);
int tic_vit_backward_layer(const TicVitState* st, int layer, tic_stream_t stream);
/* overwrite_dw != 0: the four weight-matrix gradients of the block are STORED (their ranges of st->grads need not be zero); biases and
 * LayerNorm gradients still accumulate.  Pairs with tic_vit_zero_grads(st, 1, ...). */
int tic_vit_backward_layer_ex(const TicVitState* st, int layer, int overwrite_dw, tic_stream_t stream);
/* st->grads = 0 before a backward (optimizer.zero_grad()).  keep_matrices != 0 leaves the per-layer weight-matrix ranges (99.7 % of
 * ViT-L) untouched, for a backward that stores them: tic_vit_backward_layer_ex(..., 1, ...) */
int tic_vit_zero_grads(const TicVitState* st, int keep_matrices, tic_stream_t stream);
int tic_vit_backward_embed(const TicVitState* st, tic_stream_t stream);

/* ---- mixture of ViT experts: gate / combine / loss arithmetic (BASELINE config 5) ------------------------------------
 * The expert and gate networks are ViT classifiers (tic_vit_*); these are the small ops between them.
 * Expert outputs are EXPERT-major: expert_out[E][B][C] fp32 (each expert, or each peer rank, owns one contiguous slab). */
/* gate (TIC/ResMoE/model.py:33-38 + the scatter at :53-54): z = logits + noise_scale * noise (noise may be NULL: eval mode);
 * top-K by value (ties: lowest expert index); softmax over the K selected values; gate_w[B,E] = those weights scattered, 0
 * elsewhere; topk_idx[B,K] int64 and topk_w[B,K] in descending order of z.  1 <= K <= 8, E <= 64. */
int tic_moe_gate(const float* logits, const float* noise, float noise_scale, float* gate_w, int64_t* topk_idx, float* topk_w,
                 int B, int E, int K, tic_stream_t stream);
/* dlogits[B,E] = gate_w (.) (d_gate_w - <gate_w, d_gate_w>): softmax Jacobian on the selected experts, 0 elsewhere */
int tic_moe_gate_bwd(const float* gate_w, const float* d_gate_w, float* dlogits, int B, int E, tic_stream_t stream);
/* out[B,C] = sum_e gate_w[b,e] expert_out[e,b,:]      (torch.bmm(gate_weights.unsqueeze(1), expert_outputs), model.py:56-57) */
int tic_moe_combine(const float* expert_out, const float* gate_w, float* out, int B, int E, int C, tic_stream_t stream);
/* d_expert_out[e,b,:] = gate_w[b,e] dout[b,:] ; d_gate_w[b,e] = <expert_out[e,b,:], dout[b,:]> */
int tic_moe_combine_bwd(const float* expert_out, const float* gate_w, const float* dout, float* d_expert_out, float* d_gate_w,
                        int B, int E, int C, tic_stream_t stream);
/* total_loss (TIC/ResMoE/train.py:21-36): loss3[0] = a_ce CE(z,t) + b_rce RCE(z,t) + a_balance BAL(w), loss3[1] = the first two
 * terms ("classification loss"), loss3[2] = BAL(w) unweighted.  RCE = -mean_b sum_c softmax(z)_c log_softmax(t)_c (log_softmax
 * of the TARGETS, as the reference writes it); BAL = mean_b <w[b,:], mean_b' w[b',:]>.  targets[B,C] fp32 (one-hot or soft).
 * dlogits[B,C] / d_gate_w[B,E] receive d loss3[0] (either may be NULL); gate_w NULL skips the balance term.  loss3 is overwritten. */
int tic_moe_loss(const float* logits, const float* targets, const float* gate_w, float* loss3, float* dlogits, float* d_gate_w,
                 int B, int C, int E, float a_ce, float b_rce, float a_balance, tic_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TIC_HIP_H */
