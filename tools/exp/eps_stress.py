"""Single-step noise prediction of sampler mode 4 (two fp16 products at the Q | K | V site), mode 3 (the trajectory kernel with three
products everywhere) and mode 2 (panel kernels, three everywhere) against
the fp64 oracle, on weights stressed towards sharp self-attention: LayerNorm-1 gains
x a, in_proj (q, k rows) x b, |x| ~ 30 as at t = 980.  Prints the largest self-attention logit of the fp64 oracle beside the
errors, which is what SD_SHARP_LOGIT_LIMIT is chosen from.   usage (GPU box): python tools/exp/eps_stress.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ddim_ref, denoiser_ref as ref  # noqa: E402
from soccerdiffusion_amd import ops  # noqa: E402

d, L, T, Mc, B, J = 256, 4, 100, 10, 4, 20
acp = ddim_ref.alphas_cumprod()
ts = ddim_ref.timesteps(50).tolist()


def stressed(seed, ln_gain, qk_gain, v_gain=1.0):
    sd = {k: v.clone() for k, v in ref.synthetic_state_dict(d, J, L, seed=seed).items()}
    for l in range(L):
        pre = f"diffusion_action_generator.transformer_decoder.layers.{l}."
        sd[pre + "norm1.weight"] *= ln_gain
        sd[pre + "self_attn.in_proj_weight"][: 2 * d] *= qk_gain
        sd[pre + "self_attn.in_proj_weight"][2 * d:] *= v_gain
    return sd


def max_logit(sd, ctx, x, t):
    rec = []
    orig = ref.attention

    def spy(q, k, v, heads, masks=None, kind=ref.SITE_SA_PROBS):
        if kind == ref.SITE_SA_PROBS:
            Bq, Tq, dd = q.shape
            hd = dd // heads
            s = (q.view(Bq, Tq, heads, hd).transpose(1, 2) @ k.view(Bq, -1, heads, hd).transpose(1, 2).transpose(-1, -2)) / math.sqrt(hd)
            rec.append(float(s.abs().max()))
        return orig(q, k, v, heads, masks, kind)

    ref.attention = spy
    try:
        ref.forward_with_context(sd, [ctx], x, torch.full((x.shape[0],), t, dtype=torch.int64), dtype=torch.float64)
    finally:
        ref.attention = orig
    return max(rec)


def rel(a, b):
    return float((a.double().cpu() - b).norm() / b.norm())


print(f"{'case':44s} {'max|logit|':>10s} {'mode4':>10s} {'mode3':>10s} {'mode2':>10s} {'cpu fp32':>10s}  status4")
cases = [("base", 1, 1, 1), ("LN1 x4", 4, 1, 1), ("in_proj(q,k) x3", 1, 3, 1), ("LN1 x4, in_proj x3", 4, 3, 1), ("LN1 x2, in_proj x2", 2, 2, 1),
         ("LN1 x4, in_proj x3, |x|~30", 4, 3, 30), ("|x|~30", 1, 1, 30), ("LN1 x6, in_proj x4", 6, 4, 1), ("LN1 x8, in_proj x6", 8, 6, 1),
         ("LN1 x3, in_proj x2", 3, 2, 1), ("LN1 x1.5, in_proj x1.5", 1.5, 1.5, 1), ("in_proj x2", 1, 2, 1)]
for name, lg, qg, xs in cases:
    for seed in (21, 22):
        sd = stressed(seed, lg, qg)
        g = torch.Generator().manual_seed(77 + seed)
        x = torch.randn(B, T, J, generator=g) * xs
        ctx = torch.randn(B, Mc, d, generator=g)
        t = ts[0]
        want = ref.forward_with_context(sd, [ctx], x, torch.full((B,), t, dtype=torch.int64), dtype=torch.float64)
        cpu32 = ref.forward_with_context(sd, [ctx], x, torch.full((B,), t, dtype=torch.int64))
        ml = max_logit(sd, ctx, x, t)
        packed = ops.pack_denoiser(sd, "cuda", max_len=T)
        toks = ops.step_token(torch.tensor(ts[:1]).cuda(), ops.step_frequencies(d).cuda(), sd["step_encoding.token"].cuda()).reshape(1, d)
        coef = ops.ddim_coefficients(ts, acp, 50)[:1]
        errs = {}
        st3 = None
        for mode in (4, 3, 2):
            status = torch.zeros(1, dtype=torch.int32, device="cuda")
            _, et = ops.ddim_sample(packed, ctx.cuda(), toks, coef, x.cuda(), eps_trace=True, max_mode=mode, status=status)
            errs[mode] = rel(et[0], want)
            if mode == 4:
                st3 = int(status.item())
        print(f"{name + ' seed ' + str(seed):44s} {ml:10.2f} {errs[4]:10.2e} {errs[3]:10.2e} {errs[2]:10.2e} {rel(cpu32, want):10.2e}  {st3}")
