"""Per-step error of the native 50-step rollout against the fp64 oracle loop (and the fp32 oracle's own error),
for the sampler GEMM modes.  usage (GPU box): python tools/sampler_error.py ; SD_SAMPLER_GEMM=f32 python tools/sampler_error.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import ddim_ref  # noqa: E402
from oracle import denoiser_ref as ref  # noqa: E402
from soccerdiffusion_amd import ops  # noqa: E402


def rel(a, b):
    return float((a.double().cpu() - b.double()).norm() / b.double().norm())


def main():
    d, L, T, Mc, B, n_steps, J = 256, 4, 100, 10, 4, 50, 20
    sd = ref.synthetic_state_dict(d, J, L, seed=9)
    g = torch.Generator().manual_seed(1234)
    x_T = torch.randn(B, T, J, generator=g)
    ctx = torch.randn(B, Mc, d, generator=torch.Generator().manual_seed(1235))
    acp = ddim_ref.alphas_cumprod()
    ts = ddim_ref.timesteps(n_steps).tolist()

    def denoise(dtype):
        def f(x, t):
            return ref.forward_with_context(sd, [ctx], x, torch.full((B,), t, dtype=torch.int64), dtype=dtype)
        return f

    want64 = ddim_ref.sample(denoise(torch.float64), x_T.double(), n_steps, acp)
    want32 = ddim_ref.sample(denoise(torch.float32), x_T, n_steps, acp)
    packed = ops.pack_denoiser(sd, "cuda", max_len=T)
    freq = ops.step_frequencies(d).cuda()
    toks = ops.step_token(torch.tensor(ts).cuda(), freq, sd["step_encoding.token"].cuda()).reshape(n_steps, d)
    coef = ops.ddim_coefficients(ts, acp, n_steps)
    _, trace = ops.ddim_sample(packed, ctx.cuda(), toks, coef, x_T.cuda(), trace=True)
    e_native = [rel(trace[i], want64[i]) for i in range(n_steps)]
    e_cpu32 = [rel(want32[i], want64[i]) for i in range(n_steps)]
    print("mode", os.environ.get("SD_SAMPLER_GEMM", "default"))
    print("native vs fp64 oracle: max over steps %.3e, final %.3e" % (max(e_native), e_native[-1]))
    print("fp32 CPU oracle vs fp64 oracle: max over steps %.3e, final %.3e" % (max(e_cpu32), e_cpu32[-1]))


if __name__ == "__main__":
    main()
