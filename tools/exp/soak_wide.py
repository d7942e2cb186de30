"""Randomised soak of the trajectory kernels against the unfused / row-panel kernels (max_mode 2): random horizons 1 .. 100, memories 0 .. 63
rows, batches 1 .. 300, joint counts, layer counts; 6-step rollouts, every shape three times with fresh data.  Prints the worst relative
difference per key-tile count.  usage (GPU box): python tools/exp/soak_wide.py [n_shapes=60]"""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import denoiser_ref as ref  # noqa: E402  (synthetic weights only)
from soccerdiffusion_amd import ops  # noqa: E402

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
random.seed(0)
worst = {}
for s in range(n_shapes):
    T, Mc, B = random.randint(1, 100), random.choice([0, 5, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, random.randint(0, 63)]), random.choice([1, 2, 7, 33, 300])
    J, L, n = random.choice([4, 8, 20, 28, 32]), random.randint(1, 4), 6
    sd = ref.synthetic_state_dict(256, J, L, seed=100 + s)
    packed = ops.pack_denoiser(sd, "cuda", max_len=T)
    ts = ops.ddim_timesteps(n)
    coef = ops.ddim_coefficients(ts, ops.alphas_cumprod(), n)
    toks = ops.step_token(torch.tensor(ts).cuda(), ops.step_frequencies(256).cuda(), sd["step_encoding.token"].cuda()).reshape(n, 256)
    for rep in range(3):
        g = torch.Generator(device="cuda").manual_seed(1000 * s + rep)
        x = torch.randn(B, T, J, device="cuda", generator=g)
        ctx = torch.randn(B, Mc, 256, device="cuda", generator=g) if Mc else None
        a = ops.ddim_sample(packed, ctx, toks, coef, x, max_mode=3)
        b = ops.ddim_sample(packed, ctx, toks, coef, x, max_mode=2)
        assert torch.isfinite(a).all()
        e = float((a - b).norm() / b.norm())
        kt = (Mc + 1 + 15) // 16
        worst[kt] = max(worst.get(kt, 0.0), e)
        assert e < 2e-5, (T, Mc, B, J, L, e)
print("shapes", n_shapes, "worst relative difference per key-tile count:", {k: f"{v:.2e}" for k, v in sorted(worst.items())})
