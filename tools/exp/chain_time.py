"""Times the fused training chains (forward feed-forward chain, backward feed-forward chain, LayerNorm + projection backward) at
d = 256 for several row counts: how far from linear in the rows the launches are (400 panels on 256 CUs at B = 256)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import ops
d = 256
g = torch.Generator(device="cuda").manual_seed(0)
def rnd(*s, scale=1.0): return torch.randn(*s, device="cuda", generator=g) * scale
def planes(W):
    nb = W.shape[0] // d
    Wt = torch.cat([W[b * d:(b + 1) * d].t().contiguous().reshape(-1) for b in range(nb)])
    off = torch.arange(nb, dtype=torch.int64, device="cuda") * d * d
    f = torch.empty(2 * d * d * nb, dtype=torch.float16, device="cuda"); t = torch.empty_like(f)
    ops.pack_weight_blocks(W.reshape(-1).contiguous(), off, nb, d, f); ops.pack_weight_blocks(W.reshape(-1).contiguous(), off, nb, d, t, transposed=True)
    return f, t
s = 1 / math.sqrt(d)
Wo, W1, W2, Wn = (planes(rnd(n, d, scale=s)) for n in (d, d, d, 3 * d))
vec = lambda n: rnd(n, scale=0.1)
bo, b1, b2, bn, g3, be3, g1, be1 = vec(d), vec(d), vec(d), vec(3 * d), 1 + vec(d), vec(d), 1 + vec(d), vec(d)
p = float(os.environ.get("P", "0.1"))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for R in (6400, 12800, 25600, 51200, 102400):
    a, h, dy, pre, x = rnd(R, d), rnd(R, d), rnd(R, d, scale=1e-3), rnd(R, d), rnd(R, d)
    dY3 = rnd(R, 3 * d, scale=1e-3)
    o = {k: torch.empty(R, d, device="cuda") for k in ("h_out", "n_out", "pre", "u", "h2_out", "nn_out")}
    y = torch.empty(R, 3 * d, device="cuda")
    dym, dpre, dx = (torch.empty(R, d, device="cuda") for _ in range(3)); dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    fwd = lambda: ops.train_fwd_chain(R, d, h, a=a, wo=Wo[0].data_ptr(), bo=bo, ln=(g3, be3), w1=W1[0].data_ptr(), b1=b1, w2=W2[0].data_ptr(), b2=b2,
                                      nln=(g1, be1), wn=Wn[0].data_ptr(), bn=bn, n_next=3, y_out=y, p=p, seed=1, sites=(1, 2, 3), **o)
    bffn = lambda: ops.train_bwd_chain(R, d, dy, W2[1].data_ptr(), dx, dym=dym if p > 0 else None, pre=pre, dpre=dpre, wt1=W1[1].data_ptr(), x=x, ln_w=g3,
                                       dres=dy, dg=dg, db=db, p=p, seed=1, sites=(4, 5))
    bproj3 = lambda: ops.train_bwd_chain(R, d, dY3, Wn[1].data_ptr(), dx, passes=3, x=x, ln_w=g1, dres=dy, dg=dg, db=db)
    bout = lambda: ops.train_bwd_chain(R, d, dy, Wo[1].data_ptr(), dx, dym=dym if p > 0 else None, p=p, seed=1, sites=(6, 0))
    print("R", R, "panels", R // 64, " fwd-ffn %.1f  bwd-ffn %.1f  bwd-proj3 %.1f  bwd-out %.1f us" % (timeit(fwd), timeit(bffn), timeit(bproj3), timeit(bout)), flush=True)
