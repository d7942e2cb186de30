import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ddim_ref, denoiser_ref as ref
from soccerdiffusion_amd import ops
d, L, T, Mc, B, n, J = 256, int(os.environ.get("L", 4)), 100, 10, 4, int(os.environ.get("N", 50)), 20
sd = ref.synthetic_state_dict(d, J, L, seed=9)
g = torch.Generator().manual_seed(1234)
x_T = torch.randn(B, T, J, generator=g); ctx = torch.randn(B, Mc, d, generator=torch.Generator().manual_seed(1235))
acp = ddim_ref.alphas_cumprod(); ts = ddim_ref.timesteps(n).tolist()
want = ddim_ref.sample(lambda x, t: ref.forward_with_context(sd, [ctx], x, torch.full((B,), t, dtype=torch.int64)), x_T, n, acp)
packed = ops.pack_denoiser(sd, "cuda", max_len=T)
toks = ops.step_token(torch.tensor(ts).cuda(), ops.step_frequencies(d).cuda(), sd["step_encoding.token"].cuda()).reshape(n, d)
for mode in (3, 2):
    _, trace = ops.ddim_sample(packed, ctx.cuda(), toks, ops.ddim_coefficients(ts, acp, n), x_T.cuda(), trace=True, max_mode=mode)
    for i in range(n):
        tr = trace[i].cpu()
        bad = (~torch.isfinite(tr)).sum().item()
        err = float((torch.nan_to_num(tr) - want[i]).norm() / want[i].norm())
        if bad or err > 1e-5 or i < 2 or i == n - 1:
            nz = (~torch.isfinite(tr)).nonzero()[:3].tolist() if bad else []
            print(f"mode {mode} step {i}: nonfinite {bad} first {nz} err {err:.3e} |x| {float(want[i].abs().max()):.3f}")
