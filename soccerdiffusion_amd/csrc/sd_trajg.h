// Host interface of the generic trajectory step kernels (sd_trajg.hip) towards the sampler's driver (sd_kernels.hip).  Not part of the
// public ABI.
#ifndef SD_TRAJG_H
#define SD_TRAJG_H
#include <hip/hip_runtime.h>
#include <stddef.h>
struct sd_denoiser_weights;

// shapes the generic kernels take: hidden_dim 128 / 256 (horizon <= 100) or 512 (horizon <= 48), 4 heads, <= 8 layers, <= 32 joints, any
// number of memory rows.  SD_SAMPLER_TRAJ=0 / SD_SAMPLER_GEMM=f32 in the environment switch them off with the other trajectory kernels.
bool trajg_ok(int d, int heads, int T, int Mk, int J, int L);
// floats of workspace behind the driver's own carve-up: split weight planes, the memory's K / V^T planes, step rows, scales
size_t trajg_workspace_floats(int B, int Mc, int d, int L, int n_tok);
// the three preparation stages of the trajectory path (see traj_prepare_* in sd_kernels.hip); kvtmp / kvstep: fp32 scratch of
// L * B * Mc * 2 d and L * n_tok * 2 d floats for the projected rows
int trajg_prepare_weights(const sd_denoiser_weights *w, float *gws, int B, int Mc, int n_tok, hipStream_t st);
int trajg_prepare_ctx(const sd_denoiser_weights *w, float *gws, const float *ctx, float *kvtmp, int B, int Mc, int n_tok, hipStream_t st);
// map (or NULL): per-trajectory step tokens, trajectory b uses the rows of token map[b] (step_map_kernel: duplicates of token 0 are prepared once)
int trajg_prepare_steps(const sd_denoiser_weights *w, float *gws, const float *tokens, float *kvstep, int B, int Mc, int n_tok, hipStream_t st,
                        const int *map = nullptr);
// one denoiser step (+ DDIM update when coef != NULL); step index i of the n_tok prepared step rows, or row b for trajectory b (per_traj)
int trajg_step(const sd_denoiser_weights *w, float *gws, float *x, float *eps, int B, int T, int Mc, int i, int n_tok, const float *coef,
               bool per_traj, hipStream_t st, const int *map = nullptr);
#endif
