"""Is sd_ddim_sample deterministic run to run, and eager vs hipGraph?  (diagnostic)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ddim_ref, denoiser_ref as ref
from soccerdiffusion_amd import ops

d, L, T, Mc, J = 256, 2, 100, 10, 20
for B, n in ((37, 8), (64, 8), (16, 8), (37, 1)):
    sd = ref.synthetic_state_dict(d, J, L, seed=11)
    acp = ddim_ref.alphas_cumprod(); ts = ddim_ref.timesteps(n).tolist()
    toks = ops.step_token(torch.tensor(ts).cuda(), ops.step_frequencies(d).cuda(), sd["step_encoding.token"].cuda()).reshape(n, d)
    coef = ops.ddim_coefficients(ts, acp, n)
    packed = ops.pack_denoiser(sd, "cuda", max_len=T)
    g = torch.Generator().manual_seed(1)
    x_T = torch.randn(B, T, J, generator=g).cuda(); ctx = torch.randn(B, Mc, d, generator=g).cuda()
    runs = []
    for r in range(3):
        x, tr = ops.ddim_sample(packed, ctx, toks, coef, x_T, trace=True)
        runs.append(tr.clone())
        ops._ws_cache.clear()   # fresh workspace next time
        junk = torch.full((1 << 26,), float(r + 1), device="cuda"); del junk
    gs = ops.GraphedSampler(packed, B, T, Mc, toks, coef)
    ga = gs(ctx, x_T); gb = gs(ctx, x_T)
    torch.cuda.synchronize()
    for r in (1, 2):
        diff = [(i, int((runs[0][i] != runs[r][i]).sum())) for i in range(n)]
        print(f"B={B} eager run0 vs run{r}: differing elements per step {diff}")
    print(f"B={B} graph vs graph equal {torch.equal(ga, gb)}; graph vs eager0 differing {int((ga != runs[0][-1]).sum())}", flush=True)
    bad = (runs[0][0] != runs[1][0]).nonzero()
    if len(bad):
        print("  first step, differing trajectories:", sorted(set(bad[:, 0].tolist())), "rows:", sorted(set(bad[:, 1].tolist()))[:20])
