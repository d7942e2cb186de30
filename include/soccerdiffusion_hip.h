/*
 * soccerdiffusion_hip.h — C ABI of the MI355X (gfx950) denoiser hot path.
 *
 * Drop-in boundary for the diffusion-policy denoising path of bit-bots/SoccerDiffusion.
 * The reference is pure Python/PyTorch and has no FFI of its own; each entry point below
 * replaces the ATen work behind one reference method (file:line relative to the
 * reference checkout) and is what a ctypes binding on the reference side would call
 * (INTEGRATION.md shows that binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major fp32 unless stated;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); no entry point
 *     allocates, frees or synchronises, so all of them can be captured into a hipGraph;
 *   - scratch memory comes from the caller: ask sd_workspace_floats() and pass a buffer of
 *     at least that many floats;
 *   - return value: 0 = ok, negative = invalid argument (SD_E_*), positive = hipError_t
 *     of the failing launch.  sd_last_error() returns a static description.
 *   - inputs are never modified unless the argument is documented as in/out.
 *
 * Numerics: data, accumulation and every elementwise operation are fp32.  Contractions run either on
 * v_mfma_f32_32x32x2_f32 (exact fp32 fma chain) or - the row GEMMs behind sd_op_linear*, and sd_ddim_sample
 * in mode 2 (sd_sampler_mode) - as three v_mfma_f32_32x32x16_f16 per product on operands split into fp16
 * hi + lo pairs (22 mantissa bits, fp32 accumulate): measured error against fp64 at or below the fp32 fma
 * chain's own.  LayerNorm eps = 1e-5 biased variance, GELU = exact erf form, softmax in
 * fp32.  Parity bar: <= 1e-4 relative L2 vs the fp32 CPU path (BASELINE.json); measured ~4e-7.
 */
#ifndef SOCCERDIFFUSION_HIP_H
#define SOCCERDIFFUSION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD_ABI_VERSION 1

#define SD_E_BADARG (-1)   /* null pointer / non-positive size                       */
#define SD_E_BADDIM (-2)   /* hidden_dim not in {64,128,256,512} or not /heads        */
#define SD_E_TOOBIG (-3)   /* sequence longer than the kernels support               */
#define SD_E_UNSUPPORTED (-4) /* sd_sampler_prepare / sd_sampler_eps: this shape does not take the trajectory kernels */

/* One pre-norm transformer layer (torch nn.TransformerDecoderLayer / EncoderLayer with
 * norm_first=True, activation="gelu", dim_feedforward=d — reference
 * soccer_diffusion/ml/model/decoder.py:25-35, encoder/base.py:29-40).  Pointer names are
 * the checkpoint keys.  Encoder layers leave ca_* and n3_* NULL and
 * use n2_* as the FFN norm. */
typedef struct sd_layer_weights {
    const float *sa_in_w;   /* self_attn.in_proj_weight      (3d, d) */
    const float *sa_in_b;   /* self_attn.in_proj_bias        (3d)    */
    const float *sa_out_w;  /* self_attn.out_proj.weight     (d, d)  */
    const float *sa_out_b;  /* self_attn.out_proj.bias       (d)     */
    const float *ca_in_w;   /* multihead_attn.in_proj_weight (3d, d) */
    const float *ca_in_b;   /* multihead_attn.in_proj_bias   (3d)    */
    const float *ca_out_w;  /* multihead_attn.out_proj.weight (d, d) */
    const float *ca_out_b;  /* multihead_attn.out_proj.bias  (d)     */
    const float *lin1_w;    /* linear1.weight (d, d) */
    const float *lin1_b;    /* linear1.bias   (d)    */
    const float *lin2_w;    /* linear2.weight (d, d) */
    const float *lin2_b;    /* linear2.bias   (d)    */
    const float *n1_w, *n1_b;  /* norm1 */
    const float *n2_w, *n2_b;  /* norm2 */
    const float *n3_w, *n3_b;  /* norm3 (decoder only) */
} sd_layer_weights;

/* DiffusionActionGenerator parameters — reference soccer_diffusion/ml/model/decoder.py:23-36. */
typedef struct sd_denoiser_weights {
    int32_t d;       /* hidden_dim                     */
    int32_t J;       /* num_joints                     */
    int32_t L;       /* num_decoder_layers             */
    int32_t heads;   /* 4 (model.py:115)               */
    const float *emb_w;  /* embedding.weight (d, J)    */
    const float *emb_b;  /* embedding.bias   (d)       */
    const float *out_w;  /* fc_out.weight    (J, d)    */
    const float *out_b;  /* fc_out.bias      (J)       */
    const float *pe;     /* positional table (T_max, d): PositionalEncoding.pe, misc.py:43-56 */
    int32_t T_max;       /* rows of pe (= trajectory_prediction_length) */
    int32_t _pad;
    const sd_layer_weights *layers;  /* HOST array of L entries (device pointers inside) */
} sd_denoiser_weights;

/* BaseEncoder parameters — reference soccer_diffusion/ml/model/encoder/base.py:27-40. */
typedef struct sd_encoder_weights {
    int32_t d;       /* hidden_dim                                  */
    int32_t C;       /* input_dim                                   */
    int32_t p;       /* patch_size (Conv1d kernel = stride)         */
    int32_t L;       /* num_layers                                  */
    int32_t heads;   /* 4 (8 for the image sequence encoder)        */
    int32_t S_max;   /* rows of pe                                  */
    const float *emb_w;  /* embedding.weight (d, C, p)  Conv1d      */
    const float *emb_b;  /* embedding.bias   (d)                    */
    const float *pe;     /* positional table (S_max, d)             */
    const sd_layer_weights *layers;  /* HOST array of L entries     */
} sd_encoder_weights;

int sd_abi_version(void);
const char *sd_last_error(void);

/* Floats of caller-provided scratch for the calls below (upper bound over all of them)
 * for batch B, horizon T (decoder tokens or encoder patches), memory tokens M. */
size_t sd_workspace_floats(int B, int T, int M, int d, int L, int n_steps);

/* Which kernels sd_ddim_sample runs for this shape (reporting only; results agree within fp32 rounding):
 *   0 = fp32 MFMA, per-step cross-attention projections;
 *   1 = fp32 MFMA with the cross-attention Q/out projections folded into the cached memory (heads 4, Mc + 1 <= 16, T >= 64);
 *   2 = mode 1 with every row GEMM and the self-attention as 3 fp16 MFMAs on split (hi + lo) operands, fp32
 *       accumulate (hidden_dim 256; SD_SAMPLER_GEMM=f32 in the environment selects mode 1 instead);
 *   3 = ONE launch per step: a workgroup owns a trajectory (embedding, every layer with its self-attention, fc_out, DDIM
 *       update; q | k | v and the residual stream never leave the CU; soccerdiffusion_amd/csrc/sd_traj.h).  Mode 2's
 *       arithmetic: three fp16 MFMAs per product on hi + lo operands at EVERY site.  hidden_dim 256, 4 heads, T <= 100 (one
 *       instantiation per ceil(T / 16) token tiles), Mc + 1 <= 64 memory rows (up to 16: one key tile in the folded
 *       cross-attention; 17 .. 64: the wide instantiation with 2 .. 4 key tiles - the reference's full-context configs, e.g.
 *       sim_scratch.yaml's 51 rows), any J <= 32 (the embedding's K and fc_out's N are zero-padded in the packed planes; the
 *       reference's database has 22 joints, soccer_diffusion/dataset/models.py:222-247); the layer count (<= 8) is checked at the
 *       call - a deeper model runs mode 2; SD_SAMPLER_TRAJ=0 in the environment selects mode 2.
 *       Every OTHER hidden_dim 128 / 256 / 512 shape - the reference's default.yaml (hidden_dim 128, 312 memory rows) and
 *       larger_model.yaml (hidden_dim 512, 8 layers, 312 memory rows), hidden_dim 256 with more than 64 memory rows - runs the GENERIC
 *       trajectory kernels (soccerdiffusion_amd/csrc/sd_trajg.hip): same ownership and arithmetic, geometry as a template parameter,
 *       the cross-attention un-folded with the memory's projected K / V streamed from HBM (any number of rows), T <= 100 (hidden_dim
 *       512: T <= 48 - a longer panel does not fit the CU's LDS; such a shape runs mode 2's unfused chains), 4 heads, J <= 32, <= 8 layers.
 *   4 = mode 3 with ONE exception: the Q | K | V projection of the self-attention reads a single fp16 plane of LayerNorm
 *       1's output (two MFMAs per product there, 11-bit activation operand), which frees the LDS that lets the four images
 *       of a head live side by side (two barriers per head instead of five; ~ 1.15 x mode 3).  The error this leaves in a
 *       logit grows with the logit: measured noise-prediction error against fp64 ~ 1e-5 x max |logit| (1.6e-5 on freshly
 *       initialised weights, 1e-4 at |logit| ~ 9, 1e-3 at 25; mode 3: 1e-6 throughout - tools/exp/eps_stress.py,
 *       profiles/r04_eps_stress.txt), so the kernel reports SD_STATUS_SHARP_LOGITS (below) when a logit leaves
 *       SD_SHARP_LOGIT_LIMIT and the caller repeats the rollout on mode 3.  Up to 16 memory rows; with more, max_mode = 4
 *       runs mode 3's wide instantiation.
 * Returns the mode an automatic call (max_mode = -1) runs: 3 where the trajectory kernel applies - mode 4 is opt-in. */
int sd_sampler_mode(int d, int heads, int T, int Mc, int J);

/* StepToken.forward — soccer_diffusion/ml/model/misc.py:25-35.
 * steps: B values, int64 if steps_is_i64 else fp32.  freq: d/4 host-built frequencies
 * (built in fp32 exactly as misc.py:32 does).  token: learned (d/2).  Writes row b of the
 * step token to out + b*out_row_stride (d floats), so the caller can aim it at the last
 * row of each sample's memory block (model.py:176). */
int sd_step_token(const void *steps, int steps_is_i64, const float *freq, const float *token,
                  float *out, long out_row_stride, int B, int d, void *stream);

/* DiffusionActionGenerator.forward — soccer_diffusion/ml/model/decoder.py:38-54.
 * x (B,T,J), memory (B,M,d) [context tokens + step token], eps_out (B,T,J). */
int sd_denoiser_forward(const sd_denoiser_weights *w, const float *x, const float *memory,
                        float *eps_out, float *workspace, int B, int T, int M, void *stream);

/* BaseEncoder.forward — soccer_diffusion/ml/model/encoder/base.py:41-53.
 * x (B,S,C) -> out (B,S/p,d). */
int sd_encoder_forward(const sd_encoder_weights *w, const float *x, float *out,
                       float *workspace, int B, int S, void *stream);

/* GameStateEncoder.forward — soccer_diffusion/ml/model/encoder/game_state.py:19-27.
 * idx: B int64 indices; table (n_states,d); writes row b to out + b*out_row_stride. */
int sd_game_state_embed(const int64_t *idx, const float *table, float *out, long out_row_stride,
                        int B, int d, int n_states, void *stream);

/* DDIMScheduler.add_noise — call site soccer_diffusion/ml/training/train.py:218.
 * t: B int64 timesteps; acp: device table alphas_cumprod (n_train). rows = T*J per sample. */
int sd_ddim_add_noise(const float *x0, const float *noise, const int64_t *t, const float *acp,
                      float *out, int B, int per_sample, void *stream);

/* DDIMScheduler.step(...).prev_sample (eta = 0, epsilon prediction, no clipping) — call
 * site soccer_diffusion/ml/inference/plot.py:131.  In place on x is allowed.
 * sqrt_a_t, sqrt_1m_a_t, sqrt_a_prev, sqrt_1m_a_prev are the four fp32 scalars. */
int sd_ddim_step(const float *eps, const float *x, float *x_prev, float sqrt_a_t, float sqrt_1m_a_t,
                 float sqrt_a_prev, float sqrt_1m_a_prev, long n, void *stream);

/* Normalizer.normalize (inverse = 0: (x - mean) / std) and .denormalize (inverse = 1:
 * x * std + mean), per joint — soccer_diffusion/dataset/pytorch.py:410-414.  n = rows * J. */
int sd_normalize(const float *x, const float *mean, const float *stdv, float *out, long n, int J, int inverse,
                 void *stream);

/* The iterated sampler — reference loops soccer_diffusion/ml/inference/plot.py:122-131,
 * ml/training/distill.py:179-189, ml/inference/ros.py:301-310: for every t in timesteps:
 * eps = forward_with_context(ctx, x, t); x = scheduler.step(eps, t, x).prev_sample.
 *   ctx (B,Mc,d) context tokens WITHOUT the step token (may be NULL when Mc = 0);
 *   step_tokens (n_steps,d): row i = StepToken(timesteps[i]) (same for the whole batch);
 *   coef (HOST pointer, n_steps*4 floats): per step sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev);
 *   x (B,T,J) in/out: x_T on entry, the sample on return;
 *   trace (n_steps,B,T,J) or NULL: x after every step (parity tests).
 * Context keys/values are projected once per rollout and the n_steps step-token
 * keys/values once per call (memory is not layer-normed before the K/V projection). */
int sd_ddim_sample(const sd_denoiser_weights *w, const float *ctx, const float *step_tokens,
                   const float *coef, float *x, float *trace, float *workspace,
                   int B, int T, int Mc, int n_steps, void *stream);

/* sd_ddim_sample with a range guard and a kernel-selection cap (same loop, same reference call sites).
 *   status (DEVICE pointer to one int32, or NULL): zeroed at the start of the call; at its end a pass over the sample
 *     ORs SD_STATUS_NONFINITE into it when any value of x is inf / NaN.  The split-fp16 kernels of mode 2 use fixed
 *     power-of-two activation scales (8 for LayerNorm / attention / GELU outputs): a value with |8 v| >= 65520 becomes an
 *     fp16 infinity, its products NaN, and the NaN stays in its trajectory down to x - so a set bit means "operand
 *     range of mode 2 exceeded (e.g. a LayerNorm weight in the thousands) or non-finite input", never silently wrong
 *     finite numbers.  No synchronisation: read the word after the stream has drained.
 *   max_mode: 0 .. 4 = never use a mode above this one; -1 = automatic = sd_sampler_mode (at most 3: valid for any weights).
 *     max_mode = 4 opts in to mode 4 where the shape allows it and REQUIRES `status` (its SD_STATUS_SHARP_LOGITS bit);
 *     max_mode <= 1 also keeps the memory K/V projections on the exact-fp32 MFMA: the rerun path after SD_STATUS_NONFINITE
 *     (soccerdiffusion_amd.ops.ddim_sample_guarded does both reruns). */
#define SD_STATUS_NONFINITE 1
/* Mode 4 only: some self-attention logit q.k / sqrt(hd) exceeded SD_SHARP_LOGIT_LIMIT in magnitude.  The two-product Q | K | V
 * site of mode 4 (see sd_sampler_mode) is validated against the fp64 oracle up to that sharpness (tests/test_gpu_denoiser.py::
 * test_mode3_noise_prediction_*: <= 5e-5, half of north_star's 1e-4); beyond it the caller repeats the rollout with max_mode = 3
 * (three products everywhere), which soccerdiffusion_amd.ops.ddim_sample_guarded does - and remembers for that model.  The
 * result of a flagged call is finite; the bit says "not validated", not "wrong". */
#define SD_STATUS_SHARP_LOGITS 2
#define SD_SHARP_LOGIT_LIMIT 5.0f
int sd_ddim_sample_ex(const sd_denoiser_weights *w, const float *ctx, const float *step_tokens,
                      const float *coef, float *x, float *trace, float *workspace,
                      int B, int T, int Mc, int n_steps, int32_t *status, int max_mode, void *stream);

/* sd_ddim_sample_ex that also hands back the denoiser's noise prediction of every step - the reference's
 * `model.forward_with_context(...)` value inside the loop (soccer_diffusion/ml/inference/plot.py:128, ml/training/distill.py:186),
 * i.e. exactly what the DDIM update of that step consumed, from whichever kernels the call selected.
 *   eps_trace (n_steps,B,T,J) or NULL.  SURVEY 8(d)'s parity gate (i) "single eps-hat" for the sampler kernels: with n_steps = 1
 *   it is one forward of mode 3's trajectory kernel (sd_denoiser_forward runs the row-panel kernels instead). */
int sd_ddim_sample_eps(const sd_denoiser_weights *w, const float *ctx, const float *step_tokens,
                       const float *coef, float *x, float *trace, float *eps_trace, float *workspace,
                       int B, int T, int Mc, int n_steps, int32_t *status, int max_mode, void *stream);

/* The denoiser evaluated ONE step at a time on the trajectory kernels of sampler modes 3 / 4 - the reference's own loop form,
 *   for t in scheduler.timesteps: eps = model.forward_with_context(ctx, x, t); x = scheduler.step(eps, t, x).prev_sample
 * (soccer_diffusion/ml/inference/plot.py:122-131, ml/training/distill.py:179-189, ml/inference/ros.py:301-310), whose
 * forward_with_context (ml/model/model.py:159-179) cannot know that it is inside a loop: what does not depend on x or the step is
 * prepared once into the caller's workspace and reused by every evaluation.
 *   sd_sampler_prepare: `what` = SD_PREPARE_WEIGHTS (abs-max + split fp16 planes of every matrix: once per weight update) |
 *     SD_PREPARE_CONTEXT (K / V of the context rows, folded with Wq / Woc as sd_ddim_sample does: once per context).
 *   sd_sampler_eps: eps (B,T,J) = denoiser(x, [ctx rows | step token]); step_tokens (n_tok,d) with n_tok = 1 (one step for the
 *     whole batch) or n_tok = B (row b = sample b's StepToken); x is only read.  status / max_mode as sd_ddim_sample_ex (3 or -1:
 *     three products everywhere, no status needed; 4: the guarded two-product Q | K | V site).
 * Both calls must see the same (w, workspace, B, T, Mc, n_tok, max_mode); workspace = sd_workspace_floats(B, T, max(Mc, 1), d, L,
 * n_tok) floats, owned by the caller and left alone between the calls.  Return SD_E_UNSUPPORTED (nothing launched) where the shape
 * does not take the trajectory kernels (sd_sampler_mode < 3): the caller then uses sd_denoiser_forward. */
#define SD_PREPARE_WEIGHTS 1
#define SD_PREPARE_CONTEXT 2
int sd_sampler_prepare(const sd_denoiser_weights *w, const float *ctx, float *workspace, int B, int T, int Mc, int n_tok,
                       int what, int max_mode, void *stream);
int sd_sampler_eps(const sd_denoiser_weights *w, const float *step_tokens, const float *x, float *eps, float *workspace,
                   int B, int T, int Mc, int n_tok, int32_t *status, int max_mode, void *stream);

/* ---- image path: ResNet basic-block convolution -------------------------------------------------------
 * y = act(BatchNorm_eval(conv3x3(x, w; stride 1, padding 1)) [+ res]) - torchvision BasicBlock's conv1/bn1/relu and
 * conv2/bn2 (+ identity) / relu as the reference configures them (soccer_diffusion/ml/model/encoder/image.py:55-83), inference mode.
 * Tensors are NHWC fp32: x (N,H,W,Cin), res / y (N,H,W,Cout); Cin, Cout multiples of 64; x, y, res, bn_scale and bn_shift 16-byte aligned
 * (SD_E_BADARG otherwise: the kernels use 16-byte accesses).  bn_scale = gamma / sqrt(var + eps),
 * bn_shift = beta - mean * bn_scale (per output channel).  Implicit GEMM with three fp16 MFMAs per product on hi + lo operands,
 * fp32 accumulate (fp32-grade results; soccerdiffusion_amd/csrc/sd_conv.hip).
 *   sd_conv3x3_pack: w (Cout,Cin,3,3) fp32 -> sd_conv3x3_packed_halfs(Cout,Cin) fp16 values in fragment order + the power-of-two
 *     scale they carry (device float) - once per weight update; amax_word: one uint32 of device scratch.
 *   x_amax / y_amax: device words holding the bits of max|x| / receiving max|y| (atomic max: zero y_amax before the call; NULL =
 *     not needed).  sd_absmax_word computes such a word for a tensor no convolution produced (x 16-byte aligned). */
size_t sd_conv3x3_packed_halfs(int Cout, int Cin);
int sd_conv3x3_pack(const float *w, int Cout, int Cin, void *planes, float *scale, uint32_t *amax_word, void *stream);
int sd_conv3x3_bn_act(const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, const float *bn_scale,
                      const float *bn_shift, const float *res, float *y, uint32_t *y_amax, int N, int H, int W, int Cin, int Cout,
                      int relu, void *stream);
int sd_absmax_word(const float *x, int64_t n, uint32_t *word, void *stream);
/* The same launch for a 1 x 1 stride-1 convolution (torchvision Bottleneck's conv1 / conv3 and the stride-1 shortcut of ResNet-50's
 * layer 1, reference option image_encoder_type "resnet50": soccer_diffusion/ml/model/encoder/image.py:62-66): weights (Cout,Cin,1,1)
 * packed by sd_conv_pack(ksize = 1); same tensors, epilogue (BatchNorm, residual, ReLU) and conventions as sd_conv3x3_bn_act. */
int sd_conv1x1_bn_act(const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, const float *bn_scale,
                      const float *bn_shift, const float *res, float *y, uint32_t *y_amax, int N, int H, int W, int Cin, int Cout,
                      int relu, void *stream);
/* The striding convolutions of a ResNet stage entry (conv1 of layers 2 - 4 and their 1 x 1 shortcut): y = act(BatchNorm_eval(conv(x, w; kernel
 * ksize = 3 with padding 1, or ksize = 1 with padding 0; stride 2))), NHWC fp32, x (N,H,W,Cin) -> y (N,ceil(H/2),ceil(W/2),Cout); Cin a
 * multiple of 64, Cout of 128; weights (Cout,Cin,ksize,ksize) packed by sd_conv_pack (sd_conv_packed_halfs fp16 values).  Same arithmetic,
 * scale and abs-max conventions as sd_conv3x3_bn_act. */
size_t sd_conv_packed_halfs(int Cout, int Cin, int ksize);
int sd_conv_pack(const float *w, int Cout, int Cin, int ksize, void *planes, float *scale, uint32_t *amax_word, void *stream);
int sd_conv_s2_bn_act(const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, const float *bn_scale,
                      const float *bn_shift, float *y, uint32_t *y_amax, int N, int H, int W, int Cin, int Cout, int ksize, int relu,
                      void *stream);
/* The ResNet stem in one launch: y = maxpool3x3/s2/p1(relu(BatchNorm_eval(conv7x7/s2/p3(x, w)))) - torchvision ResNet.conv1 / bn1 / relu /
 * maxpool as the reference instantiates them (soccer_diffusion/ml/model/encoder/image.py:55-83).  x (N,3,H,W) fp32 NCHW frames (the
 * reference's own layout) -> y (N,Hp,Wp,64) fp32 NHWC with Hc = (H-1)/2+1, Hp = (Hc-1)/2+1 (likewise W); w (64,3,7,7) packed by
 * sd_stem_pack (sd_stem_packed_halfs fp16 values).  Same arithmetic, scale and abs-max conventions as sd_conv3x3_bn_act. */
size_t sd_stem_packed_halfs(void);
int sd_stem_pack(const float *w, void *planes, float *scale, uint32_t *amax_word, void *stream);
int sd_stem_conv_bn_relu_pool(const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, const float *bn_scale,
                              const float *bn_shift, float *y, uint32_t *y_amax, int N, int H, int W, void *stream);

/* ---- single-op entry points (unit parity tests and host-side composition) ---------- */

/* out[R,N] = act(LN?(A)[R,d] @ W[N,d]^T + bias) (+ res).  ln_w/ln_b NULL = no LayerNorm;
 * act: 0 none, 1 gelu(erf); res NULL or [R,N] (may alias out).  N % d == 0. */
int sd_op_linear(const float *A, const float *W, const float *bias, const float *ln_w,
                 const float *ln_b, const float *res, float *out, int R, int N, int d, int act,
                 void *stream);

/* Multi-head attention core, unmasked: out[b,i,h*hd:(h+1)*hd] = softmax(q k^T / sqrt(hd)) v.
 * q rows (B*Tq) with row stride ldq; k, v rows (B*S) with row stride ldkv; optional extra
 * key/value row shared by the whole batch (k_extra/v_extra, d floats each, NULL = none).
 * A packed self-attention buffer (k = q + d, v = q + 2d, ldq = ldkv = 3d, Tq = S <= 128, head dim 64,
 * no extra row) runs on the split-fp16 MFMA kernel of the sampler (22-bit operands, fp32 accumulate). */
int sd_op_attention(const float *q, int ldq, const float *k, const float *v, int ldkv,
                    const float *k_extra, const float *v_extra, float *out, int ldo,
                    int B, int Tq, int S, int d, int heads, void *stream);

/* out (B,S/p,d) = Conv1d(k=s=p)(x (B,S,C)) + bias + pe[:S/p]; p = 1 is nn.Linear + PE. */
int sd_op_patch_embed(const float *x, const float *w, const float *b, const float *pe,
                      float *out, int B, int S, int C, int p, int d, void *stream);

/* eps[R,J] = h[R,d] @ W[J,d]^T + b.  If x_io != NULL also applies the DDIM update to it in
 * place using coef[4] (see sd_ddim_step).  eps may be NULL when x_io is given. */
int sd_op_fc_out(const float *h, const float *W, const float *b, float *eps, float *x_io,
                 const float *coef4_host, int R, int d, int J, void *stream);

/* ---- training: backward of the blocks, loss, optimizer -------------------------------
 * One training step of the reference (soccer_diffusion/ml/training/train.py:204-240:
 * add_noise, forward, F.mse_loss, backward, AdamW.step) at dropout p = 0 is composed from
 * these entry points by soccerdiffusion_amd/training.py (autograd nodes per block). */

/* sd_op_linear with a row stride on A (lda >= d, multiple of 4): lets dX = dY[:, slice] W^T-slices
 * accumulate through `res` without copying the slice. */
int sd_op_linear_strided(const float *A, int lda, const float *W, const float *bias, const float *ln_w,
                         const float *ln_b, const float *res, float *out, int R, int N, int d, int act,
                         void *stream);

/* sd_op_attention that also writes lse2[b, head, q] = log2-sum-exp of the scaled scores
 * (log2 domain), which the backward uses to recompute the probabilities. */
int sd_op_attention_lse(const float *q, int ldq, const float *k, const float *v, int ldkv, float *out,
                        int ldo, float *lse2, int B, int Tq, int S, int d, int heads, void *stream);

/* dq, dk, dv of softmax(q k^T / sqrt(hd)) v given dO.  dk/dv rows are fully overwritten. */
int sd_op_attention_bwd(const float *q, int ldq, const float *k, const float *v, int ldkv, const float *o,
                        int ldo, const float *dO, int lddo, const float *lse2, float *dq, int lddq, float *dk,
                        float *dv, int lddkv, int B, int Tq, int S, int d, int heads, void *stream);

/* ---- dropout (training) -----------------------------------------------------------------------------------------
 * The reference trains with torch's default dropout p = 0.1 (soccer_diffusion/ml/model/decoder.py:26-33 and
 * encoder/base.py:29-40 never set it; train.py never calls .eval()): on the attention probabilities
 * (nn.MultiheadAttention), after each attention out-projection (dropout1 / dropout2), after the GELU and after linear2
 * (dropout / dropout3).  ONE counter-based mask for every kernel, nothing stored: element (row, col) of the logical
 * (rows x width) tensor of site `site` is kept iff word (col & 3) of Philox4x32-7(counter = {quad lo, quad hi, site lo,
 * site hi}, key = seed) >= p * 2^32, quad = (row * ceil4(width) + col) >> 2; kept values are scaled by 1 / (1 - p).
 * The backward entry points regenerate the mask from the same (p, seed, site).  p = 0 is the parity path. */

/* out = x o mask (forward of a stand-alone site; backward of a fused one: dy o mask).  out may alias x. */
int sd_op_dropout(const float *x, float *out, long rows, int width, float p, uint64_t seed, uint64_t site, void *stream);
/* mask[rows, width] = 0 or 1 / (1 - p): what the kernels apply (tests hand it to the oracle). */
int sd_op_dropout_mask(float *mask, long rows, int width, float p, uint64_t seed, uint64_t site, void *stream);
/* out = gelu(pre) o mask;  dpre = dy o mask o gelu'(pre)   (FFN: linear2(dropout(gelu(linear1 x)))) */
int sd_op_gelu_dropout_fwd(const float *pre, float *out, long rows, int width, float p, uint64_t seed, uint64_t site, void *stream);
int sd_op_gelu_dropout_bwd(const float *dy, const float *pre, float *dpre, long rows, int width, float p, uint64_t seed,
                           uint64_t site, void *stream);
/* out[R,N] = res + dropout(A[R,d] W[N,d]^T + bias), mask rows = R, width = N (x + dropout1(sa_block(x)) etc.);
 * A may have a row stride lda >= d (0 = d).  res is required and may alias out. */
int sd_op_linear_dropout(const float *A, int lda, const float *W, const float *bias, const float *res, float *out, int R,
                         int N, int d, float p, uint64_t seed, uint64_t site, void *stream);
/* Training GEMMs on pre-split weights.  sd_pack_weight_blocks splits n_blocks d x d fp32 blocks (block b = d rows of d
 * floats at src + src_off_dev[b]; offsets in a DEVICE int64 array) into the fp16 hi | lo fragment planes the kernels read
 * (2 d^2 halfs per block, consecutive in dst; scale 2^8: |w| < 256) in ONE launch - once per optimizer step for every weight
 * and, with transposed != 0 (planes of the block's TRANSPOSE), for the B operands of the dX GEMMs.  sd_op_linear_packed is sd_op_linear_strided / sd_op_linear_dropout with `wpk` =
 * the planes of the N / d consecutive blocks of W instead of W itself (LayerNorm or residual [+ dropout], no activation). */
int sd_pack_weight_blocks(const float *src, const int64_t *src_off_dev, int n_blocks, void *dst, int d, int transposed, void *stream);
int sd_op_linear_packed(const float *A, int lda, const void *wpk, const float *bias, const float *ln_w, const float *ln_b,
                        const float *res, float *out, int R, int N, int d, float p, uint64_t seed, uint64_t site, void *stream);

/* ---- Fused row chains of one transformer layer for TRAINING (reference: the nn.TransformerDecoderLayer / EncoderLayer
 * blocks built by soccer_diffusion/ml/model/decoder.py:26-33 and encoder/base.py:29-40, norm_first = True, as autograd runs
 * them in ml/training/train.py:204-240).  One 64-row panel per workgroup goes through every row-local operation between
 * two attention cores in ONE launch; the forward writes exactly the tensors the backward and the weight-gradient GEMMs read.
 * All weights are split planes (sd_pack_weight_blocks); all row tensors are [R, d] fp32, contiguous, unless noted.
 * d in {64, 128, 256}; p = 0 turns every mask off; mask rows = R, width = d (sites as in sd_op_linear_dropout /
 * sd_op_gelu_dropout_fwd).  Stages with a NULL first pointer are skipped.
 *
 * sd_train_fwd_chain:
 *   [a]    h1 = h_in + dropout_{site_out}(a Wo^T + bo)                 -> h_out            (a == NULL: h1 = h_in)
 *   [w1]   n = LN(h1; ln_w, ln_b) -> n_out;  pre = n W1^T + b1 -> pre;  u = dropout_{site_act}(gelu(pre)) -> u
 *          h2 = h1 + dropout_{site_ffn}(u W2^T + b2)                   -> h2_out           (w1 == NULL: h2 = h1)
 *   [wn]   nn = LN(h2; nln_w, nln_b) -> nn_out;  y = nn Wn^T + bn      -> y_out [R, n_next d]   (n_next = 0: stop)
 * sd_train_bwd_chain (dY [R, passes d] with row stride ldy; wt = planes of the `passes` TRANSPOSED blocks):
 *   g = dropout_{site_in}(dy) -> dym (p > 0, passes == 1, not for the x-without-pre form);   t = g Wt   (dX of a linear layer)
 *   [pre]  t = t o gelu'(pre) o mask_{site_act} -> dpre;  t = t Wt1                    (back through FFN1)
 *   [x]    dx = LayerNorm-backward(t; x, ln_w) + dres -> dx;  dg += sum_rows t o xhat;  db += sum_rows t   (fp32 atomics)
 *          (x == NULL: dx = t) */
#define SD_AMAX_WORDS 64
typedef struct sd_train_fwd_chain_args {
    int64_t R;
    int32_t d, n_next;
    const float *a; const void *wo; const float *bo; const float *h_in; float *h_out;
    const float *ln_w, *ln_b; float *n_out; const void *w1; const float *b1; float *pre; float *u;
    const void *w2; const float *b2; float *h2_out;
    const float *nln_w, *nln_b; float *nn_out; const void *wn; const float *bn; float *y_out;
    float p;
    uint64_t seed, site_out, site_act, site_ffn;
    /* optional: bits of max |x| over a, n_out, u, nn_out are atomically max-ed into these arrays of SD_AMAX_WORDS words
     * (zero them first; a workgroup uses word blockIdx % SD_AMAX_WORDS) - the per-tensor scales of sd_gemm_tn_grouped, a
     * by-product of the row passes */
    uint32_t *amax_a, *amax_n, *amax_u, *amax_nn;
    uint32_t *amax_h2;   /* optional: max |h2_out| (the input of fc_out after the last layer) */
} sd_train_fwd_chain_args;
typedef struct sd_train_bwd_chain_args {
    int64_t R;
    int32_t d, passes, ldy;
    const float *dy; float *dym; const void *wt;
    const float *pre; float *dpre; const void *wt1;
    const float *x; const float *ln_w; const float *dres; float *dg; float *db;
    float *dx;
    float p;
    uint64_t seed, site_in, site_act;
    uint32_t *amax_dy, *amax_dpre;   /* optional, as above: max |.| of the (masked) dy over all passes, and of dpre */
    uint32_t *amax_dx;               /* optional: max |dx| of the LayerNorm-backward form (the embedding's dY under layer 0) */
} sd_train_bwd_chain_args;
int sd_train_fwd_chain(const sd_train_fwd_chain_args *args, void *stream);
int sd_train_bwd_chain(const sd_train_bwd_chain_args *args, void *stream);

/* The forward of ONE decoder layer for training with a workgroup per trajectory (soccerdiffusion_amd/csrc/sd_train_traj.hip): the self-attention
 * core, [h1 = h + drop(a_sa Wo^T + bo); n2 = LN2(h1); q = n2 Wq^T + bq], the cross-attention core over the M projected memory rows,
 * [h2 = h1 + drop(a_ca Woc^T + boc); nf = LN3(h2); pre = nf W1^T + b1; u = drop(gelu(pre)); h3 = h2 + drop(u W2^T + b2)] and - when w_n
 * is given - the next layer's [nn1 = LN1'(h3); qkv2 = nn1 Wn^T + bn] in one launch, i.e. sd_op_attention_lse_dropout x 2 +
 * sd_train_fwd_chain x 2 of the same layer (reference: nn.TransformerDecoderLayer, norm_first, as built by
 * soccer_diffusion/ml/model/decoder.py:26-33; training loop ml/training/train.py:204-240).  It stores the same tensors in the same layouts
 * (rows [B*T, 256]; qkv / qkv2 [B*T, 768]; lse [B, 4, T] as sd_op_attention_lse; kv [B, M, 512] = the memory's K | V projection) and the
 * same abs-max words, with dropout masks at the same (site, row, column) indices, so sd_train_bwd_chain / sd_op_attention_bwd_dropout /
 * sd_gemm_tn_grouped consume them unchanged.  hidden_dim 256, 4 heads, T <= 100, M <= 16 (sd_train_layer_fwd_ok).  Weights: planes from
 * sd_pack_weight_traj (sd_pack_weight_traj_halfs(N, K) fp16 values; K = 256 - or <= 32 for the embedding -, N = 256, or 768 for w_n),
 * repacked after every optimizer step. */
typedef struct sd_train_layer_fwd_args {
    int32_t B, T, M, d, heads;
    const float *h, *qkv;
    float *a_sa, *lse_sa, *h1, *n2, *q;
    const float *kv;
    float *a_ca, *lse_ca, *h2, *nf, *pre, *u, *h3, *nn1, *qkv2;
    const void *w_o, *w_q, *w_oc, *w_1, *w_2, *w_n;
    const float *b_o, *b_q, *b_oc, *b_1, *b_2, *b_n;
    const float *n2_w, *n2_b, *n3_w, *n3_b, *nn_w, *nn_b;
    float p;
    uint64_t seed, site_sa_probs, site_sa_out, site_ca_probs, site_ca_out, site_act, site_ffn;
    uint32_t *amax_a_sa, *amax_n2, *amax_a_ca, *amax_nf, *amax_u, *amax_nn, *amax_out;   /* SD_AMAX_WORDS words each, or NULL */
} sd_train_layer_fwd_args;
int sd_train_layer_fwd_ok(int d, int heads, int T, int M);
int sd_train_layer_fwd(const sd_train_layer_fwd_args *args, void *stream);
/* The entry of the decoder stack in the same geometry: h0 = x Wemb^T + b + pe[:T] (nn.Linear(J -> 256) + PositionalEncoding,
 * soccer_diffusion/ml/model/decoder.py:48-50), n1 = LN1(h0) of layer 0, qkv = n1 Wqkv^T + b: sd_op_patch_embed (p = 1) + the head launch of
 * sd_train_fwd_chain in one.  x (B,T,J), J a multiple of 4 up to 32; w_emb / w_qkv: planes from sd_pack_weight_traj ((256, J) and (768, 256)). */
int sd_train_head_fwd(const float *x, const void *w_emb, const float *b_emb, const float *pe, float *h0, const float *ln_w, const float *ln_b,
                      float *n1, const void *w_qkv, const float *b_qkv, float *qkv, uint32_t *amax_n1, int B, int T, int J, void *stream);
size_t sd_pack_weight_traj_halfs(int N, int K);
int sd_pack_weight_traj(const float *w, int N, int K, void *planes, void *stream);
/* n matrices in one launch: matrix i = rows[i] x 256 floats at base + src_offsets[i] (floats; rows multiples of 16) -> planes +
 * dst_offsets[i] (halfs); the three arrays are DEVICE arrays, max_rows = the largest rows[i]. */
int sd_pack_weight_traj_multi(const float *base, const int64_t *src_offsets, const int32_t *rows, const int64_t *dst_offsets, int n,
                              int max_rows, void *planes, void *stream);

/* Several weight gradients dW += dY^T X (and db += column sums of dY, db may be NULL) in one launch, on the fp16 matrix
 * pipe with ONE power-of-two scale per operand tensor: amax_dy / amax_x point at SD_AMAX_WORDS device words whose maximum is
 * the bits of (an upper bound of) max |dY| / max |X| - what sd_train_*_chain leave behind.  dY [R, N] and X [R, K] with row strides ldy / ldx
 * (multiples of 4, 16-byte aligned), N and K multiples of 4 (tiles of 128 x 128, ragged ones masked); dW [N, K] with row
 * stride ldw.  Accumulates with fp32 atomics.  sd_op_absmax fills the words of an operand no chain produced (zero them first). */
typedef struct sd_gemm_tn_problem {
    const float *dY; const float *X; float *dW; float *db;
    const uint32_t *amax_dy; const uint32_t *amax_x;
    int64_t R;
    int32_t N, K, ldy, ldx, ldw;
} sd_gemm_tn_problem;
int sd_gemm_tn_grouped(const sd_gemm_tn_problem *problems, int n_problems, void *stream);
/* Host-only: the slabs-per-workgroup sd_gemm_tn_grouped would choose for cnt (<= 64) problems of R[i] rows and N[i] x K[i]
 * outputs, and the workgroups that makes (*total_wgs, may exceed one chip-full for very large groups).  No launch; -1 on bad
 * arguments.  (CPU unit test of the partitioning arithmetic.) */
long sd_gemm_tn_grouped_plan(const long *R, const long *N, const long *K, int cnt, long *total_wgs);
int sd_op_absmax(const float *x, int64_t rows, int width, int ld, uint32_t *amax, void *stream);

/* sd_op_attention_lse / sd_op_attention_bwd with dropout on the probabilities: O = (softmax(S) o mask) V, the softmax
 * normaliser and lse2 are those of the un-dropped probabilities; mask rows = (b * heads + h) * Tq + q, width = S. */
int sd_op_attention_lse_dropout(const float *q, int ldq, const float *k, const float *v, int ldkv, float *out,
                                int ldo, float *lse2, int B, int Tq, int S, int d, int heads, float p, uint64_t seed,
                                uint64_t site, void *stream);
int sd_op_attention_bwd_dropout(const float *q, int ldq, const float *k, const float *v, int ldkv, const float *o,
                                int ldo, const float *dO, int lddo, const float *lse2, float *dq, int lddq, float *dk,
                                float *dv, int lddkv, int B, int Tq, int S, int d, int heads, float p, uint64_t seed,
                                uint64_t site, void *stream);

/* dW[N,K] += dY[R,N]^T X[R,K];  db[N] += column sums of dY (db may be NULL).  Accumulates
 * with fp32 atomics: zero dW/db first (summation order is not fixed). */
int sd_op_gemm_tn(const float *dY, int ldy, const float *X, int ldx, float *dW, int ldw, float *db, long R,
                  int N, int K, void *stream);

/* y = LayerNorm(x) * g + b (eps 1e-5, biased variance); mean/rstd (R) may be NULL. */
int sd_op_layernorm_fwd(const float *x, const float *g, const float *b, float *y, float *mean, float *rstd,
                        long R, int d, void *stream);
/* dx = LN backward of dy (+ dres if not NULL; dx may alias dres); dg/db accumulate (atomics). */
int sd_op_layernorm_bwd(const float *dy, const float *x, const float *mean, const float *rstd, const float *g,
                        const float *dres, float *dx, float *dg, float *db, long R, int d, void *stream);

int sd_op_gelu_fwd(const float *pre, float *out, long n, void *stream);
int sd_op_gelu_bwd(const float *dy, const float *pre, float *dpre, long n, void *stream);

/* out[c] += sum over rows of src[r*row_stride + c], c < width. */
int sd_op_colsum(const float *src, long row_stride, long rows, int width, float *out, void *stream);
/* out[R,N] = A[R,K] B[K,N] for K <= 64 (input gradient of fc_out). */
int sd_op_small_k_matmul(const float *A, const float *Bm, float *out, long R, int K, int N, void *stream);

/* F.mse_loss(pred, target) -> loss[0]; grad (may be NULL) = 2 (pred - target) / n.
 * scratch256d: 256 doubles of device scratch.  Deterministic. (train.py:229) */
int sd_mse_loss(const float *pred, const float *target, float *loss, float *grad, void *scratch256d, long n,
                void *stream);

/* torch.optim.AdamW update on flat buffers (train.py:162,239), `step` = 1-based step count. */
int sd_adamw_step(float *p, const float *g, float *m, float *v, long n, double lr, double beta1, double beta2,
                  double eps, double weight_decay, long step, void *stream);

/* The same update with its seven scalars in DEVICE memory (hyper7[0..6] = 1 - lr*wd, 1 - beta1, beta2, 1 - beta2,
 * lr / (1 - beta1^step), sqrt(1 - beta2^step), eps; sd_adamw_hyper fills a HOST array with them): kernel arguments are
 * frozen in a captured hipGraph, the learning rate (OneCycleLR), beta1 and the bias corrections change every step. */
int sd_adamw_step_dev(float *p, const float *g, float *m, float *v, long n, const float *hyper7_dev, void *stream);
int sd_adamw_hyper(double lr, double beta1, double beta2, double eps, double weight_decay, long step, float *hyper7_host);

/* Per-step part of the dropout mask key from DEVICE memory: when set (non-NULL), every dropout kernel adds *device_word to
 * the high half of its Philox key at run time, so a hipGraph replay of a training step draws fresh masks once the host has
 * changed the word.  Process-wide; NULL switches it off. */
int sd_set_dropout_epoch(const uint32_t *device_word);

/* ---- image path, TRAINING: torchvision BasicBlock / Bottleneck under autograd with BatchNorm2d in training mode, as the
 * reference trains its backbone with every step (soccer_diffusion/ml/training/train.py:226-240 -> ml/model/encoder/image.py:38-83).  NHWC fp32
 * tensors of npix = N * H * W pixels x C channels (C in {64, 128, 256, 512, 1024, 2048}), 16-byte aligned (soccerdiffusion_amd/csrc/sd_conv_train.hip).
 *   sd_bn_train_fwd: batch statistics of y (the raw convolution output) -> mean, rstd = 1 / sqrt(biased var + eps) (C floats each);
 *     z = relu?((y - mean) rstd gamma + beta (+ res)); running_mean / running_var updated as torch.nn.BatchNorm2d does (momentum, unbiased
 *     variance) unless NULL; z_amax (one uint32, zeroed by the caller) receives the bits of max |z| for the next convolution's fp16 scale.
 *     acc: 2 C doubles (the channel sums, an output); scratch: sd_bn_scratch_floats(npix, C) floats - the workgroups' partial sums (plain
 *     stores, added up in double by a second launch: no atomics, deterministic).
 *   sd_bn_train_bwd: g = dz behind the ReLU mask (z > 0; z NULL with relu = 1 and NO residual operand: the mask is recomputed from y, gamma, beta -
 *     one tensor less to read; beta may be NULL otherwise); dgamma = sum g x_hat, dbeta = sum g,
 *     dy = gamma rstd (g - mean(g) - x_hat mean(g x_hat)); dres (or NULL) receives g, the gradient of the residual operand; dy_amax as above.
 *   sd_conv_wgrad: dw (Cout, Cin, k, k) (torch layout, ZEROED by the caller) += sum over output pixels of dy[pixel][co] x[pixel * stride + tap - k/2][ci]
 *     for the k x k (1 or 3), stride 1 or 2, padding k / 2 convolution of x (N,H,W,Cin) with output dy (N,Ho,Wo,Cout); Cin, Cout multiples
 *     of 64.  Split-fp16 MFMAs, fp32 atomics.  dy_amax / x_amax: the tensors' abs-max words (as sd_bn_train_bwd / sd_bn_train_fwd leave them)
 *     -> one power-of-two scale per operand for the launch; either NULL -> block floating point per 32 pixels (abs-max taken inside).
 * The data gradient is the forward convolution (sd_conv3x3_bn_act / sd_conv1x1_bn_act, identity epilogue) of dy - dilated with zeros for a
 * stride-2 convolution - with the flipped, transposed weights. */
size_t sd_bn_scratch_floats(int64_t npix, int C);
int sd_bn_train_fwd(const float *y, const float *gamma, const float *beta, const float *res, float *z, float *mean, float *rstd,
                    float *running_mean, float *running_var, double *acc, float *scratch, uint32_t *z_amax, int64_t npix, int C, float eps,
                    float momentum, int relu, void *stream);
int sd_bn_train_bwd(const float *dz, const float *z, const float *y, const float *mean, const float *rstd, const float *gamma, const float *beta,
                    float *dy,
                    float *dres, float *dgamma, float *dbeta, double *acc, float *scratch, uint32_t *dy_amax, int64_t npix, int C, int relu,
                    void *stream);
/* Data gradient of a 3 x 3 / stride-2 / padding-1 convolution (conv1 of ResNet layers 2 - 4): dx (N, H, W, Cout) = the transposed convolution of
 * dy (N, (H + 1) / 2, (W + 1) / 2, Cin) with w_planes = sd_conv3x3_pack of the FLIPPED, TRANSPOSED weights (Cout = the forward's input channels) -
 * what sd_conv3x3_bn_act computes on dy dilated with zeros, without the zeros: one launch per parity class of the output pixels, each with
 * its 1 / 2 / 2 / 4 live taps (csrc/sd_conv.hip: convt3x3_s2_kernel).  Epilogue as sd_conv3x3_bn_act without ReLU: dx = acc * bn_scale + bn_shift (+ res). */
int sd_convt3x3_s2(const float *dy, const void *w_planes, const float *w_scale, const uint32_t *dy_amax, const float *bn_scale, const float *bn_shift,
                   const float *res, float *dx, uint32_t *dx_amax, int N, int H, int W, int Cin, int Cout, void *stream);
/* ... and of the 1 x 1 / stride-2 shortcut (w_planes = sd_conv_pack(ksize 1) of the transposed weights): dx[2 i][2 j] = dy[i][j] . w; the other
 * three quarters of dx are NOT written - the caller zeroes dx first. */
int sd_convt1x1_s2(const float *dy, const void *w_planes, const float *w_scale, const uint32_t *dy_amax, const float *bn_scale, const float *bn_shift,
                   float *dx, int N, int H, int W, int Cin, int Cout, void *stream);
/* The stem's BatchNorm (batch statistics) + ReLU + max-pool 3 x 3 / stride 2 / padding 1 (torchvision ResNet.forward: maxpool(relu(bn1(conv1 x)))) fused:
 * y (N, Hc, Wc, C) -> p (N, Hp, Wp, C), Hp = (Hc - 1) / 2 + 1, and idx (N, Hp, Wp, C / 4 words: one byte per element = the winner's position 0 .. 8
 * in its window, first maximum in scan order as ATen's kernel, 9 = no positive value); z = relu(BN(y)) is never materialised.  The backward gathers
 * the pool's gradient from dp and idx (no dz tensor) inside the two BatchNorm-backward launches: dy, dgamma, dbeta as sd_bn_train_bwd.
 * acc / scratch / the abs-max words as sd_bn_train_fwd (scratch: sd_bn_scratch_floats(N * Hc * Wc, C)). */
int sd_bn_relu_pool_fwd(const float *y, const float *gamma, const float *beta, float *p, uint32_t *idx, float *mean, float *rstd, float *running_mean,
                        float *running_var, double *acc, float *scratch, uint32_t *p_amax, int N, int Hc, int Wc, int C, float eps, float momentum,
                        void *stream);
int sd_bn_relu_pool_bwd(const float *dp, const uint32_t *idx, const float *y, const float *mean, const float *rstd, const float *gamma, float *dy,
                        float *dgamma, float *dbeta, double *acc, float *scratch, uint32_t *dy_amax, int N, int Hc, int Wc, int C, void *stream);
/* The stem in training: sd_stem_conv_raw = the bare 7 x 7 / stride-2 / padding-3 convolution of sd_stem_conv_bn_relu_pool (same packed planes,
 * same kernel) -> y_raw (N, Hc, Wc, 64) NHWC, Hc = (H - 1) / 2 + 1; sd_stem_wgrad: its weight gradient dw (64, 3, 7, 7) from dy (N, Hc, Wc, 64)
 * and the frames x (N, 3, H, W), both with their abs-max words; scratch: sd_stem_wgrad_scratch_floats(N, H, W) floats.  (The frames need no gradient.) */
int sd_stem_conv_raw(const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, float *y_raw, int N, int H, int W,
                     void *stream);
size_t sd_stem_wgrad_scratch_floats(int N, int H, int W);
int sd_stem_wgrad(const float *dy, const float *x, const uint32_t *dy_amax, const uint32_t *x_amax, float *dw, float *scratch, int N, int H, int W,
                  void *stream);
/* scratch (sd_conv_wgrad_scratch_floats floats, or NULL): the 3 x 3 kernel (LDS-staged column strips, all taps of a staging; stride 2: one launch
 * per row / column parity class of x) stores one partial tile per workgroup there and a second launch adds them up (no atomics: hundreds of
 * workgroups adding to the same Cout x Cin x 9 floats are bound by the L2's atomic rate; deterministic).  Without scratch (or abs-max words), and for
 * 1 x 1 convolutions, the per-wave kernel with atomics runs. */
size_t sd_conv_wgrad_scratch_floats(int N, int H, int W, int Cin, int Cout, int ksize, int stride);
int sd_conv_wgrad(const float *dy, const float *x, const uint32_t *dy_amax, const uint32_t *x_amax, float *dw, float *scratch, int N, int H, int W,
                  int Cin, int Cout, int ksize, int stride, void *stream);

/* ---- measurement hooks (bench.py roofline leg; not part of the reference's surface) ----
 * While enabled, every kernel launch made by this library is bracketed by a hipEvent pair
 * on the launch stream.  sd_profile_collect waits for them, returns the summed device
 * milliseconds and launch counts per kernel class, and clears the records.  Do not
 * enable during hipGraph capture. */
#define SD_KCLASS_PANEL_GEMM 0   /* panel_gemm_kernel<...>  (all LN/act/res variants)  */
#define SD_KCLASS_ATTENTION 1    /* attention_kernel<HD>                                */
#define SD_KCLASS_PATCH_EMBED 2  /* patch_embed_kernel                                  */
#define SD_KCLASS_FC_OUT 3       /* fc_out_kernel (+ fused DDIM update)                 */
#define SD_KCLASS_LAYER_CHAIN 4  /* decoder_layer_kernel / chain_a_kernel / chain_b_kernel */
#define SD_KCLASS_HEAD 5         /* decoder_head_kernel (embed + LN1 + QKV of layer 0)   */
#define SD_KCLASS_TRAJ_STEP 6    /* traj_step_kernel: one whole denoiser step per launch (sampler mode 3) */
#define SD_KCLASS_COUNT 7
int sd_profile_enable(int on);
int sd_profile_collect(double *ms_by_class, long *launches_by_class, int n_classes);

#ifdef __cplusplus
}
#endif
#endif /* SOCCERDIFFUSION_HIP_H */
