"""sd_train_layer_fwd against the launches it replaces (attention cores + row chains), tensor by tensor, on random data.
usage (GPU box): python tools/exp/dbg_train_layer.py [T=100] [M=11] [B=3] [p=0.0]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
M = int(sys.argv[2]) if len(sys.argv) > 2 else 11
B = int(sys.argv[3]) if len(sys.argv) > 3 else 3
p = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
d, heads = 256, 4
R = B * T
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def rn(*s, scale=1.0):
    return torch.randn(*s, device=dev, generator=g) * scale


h, qkv, kv = rn(B, T, d), rn(B, T, 3 * d), rn(B, M, 2 * d)
W = {k: rn(d, d, scale=d ** -0.5) for k in ("o", "q", "oc", "1", "2")}
Wn = rn(3 * d, d, scale=d ** -0.5)
b = {k: rn(d, scale=0.1) for k in ("o", "q", "oc", "1", "2")}
bn = rn(3 * d, scale=0.1)
ln = {k: (1 + rn(d, scale=0.1), rn(d, scale=0.1)) for k in ("2", "3", "n")}
seed, sites = 1234, (11, 12, 13, 14, 15, 16)

# ---- reference: the existing launches (weights split per row panel inside linear(); dropout through the chain kernels)
new = lambda *s: torch.empty(*s, device=dev)  # noqa: E731
a_sa, lse_sa = ops.attention_lse(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], heads, (p, seed, sites[0]) if p else None)


def lin(A, Wm, bias, res=None, drop=None, lnp=None):
    if drop is not None and p:
        return ops.linear_dropout(A.contiguous(), Wm, bias, res, (p, seed, drop))
    return ops.linear(A, Wm, bias, ln=lnp, res=res)


h1 = lin(a_sa.view(R, d), W["o"], b["o"], res=h.view(R, d), drop=sites[1])
n2 = torch.nn.functional.layer_norm(h1, (d,), *ln["2"])
q = ops.linear(n2, W["q"], b["q"])
a_ca, lse_ca = ops.attention_lse(q.view(B, T, d), kv[..., :d], kv[..., d:], heads, (p, seed, sites[2]) if p else None)
h2 = lin(a_ca.view(R, d), W["oc"], b["oc"], res=h1, drop=sites[3])
nf = torch.nn.functional.layer_norm(h2, (d,), *ln["3"])
pre = ops.linear(nf, W["1"], b["1"])
u = torch.nn.functional.gelu(pre)
if p:
    u = u * ops.dropout_mask(R, d, (p, seed, sites[4]), dev)
h3 = lin(u, W["2"], b["2"], res=h2, drop=sites[5])
nn1 = torch.nn.functional.layer_norm(h3, (d,), *ln["n"])
qkv2 = ops.linear(nn1, Wn, bn)
want = dict(a_sa=a_sa, lse_sa=lse_sa, h1=h1, n2=n2, q=q, a_ca=a_ca, lse_ca=lse_ca, h2=h2, nf=nf, pre=pre, u=u, h3=h3, nn1=nn1, qkv2=qkv2)

# ---- the layer kernel
out = dict(a_sa=new(B, T, d), lse_sa=new(B, heads, T), h1=new(R, d), n2=new(R, d), q=new(R, d), a_ca=new(B, T, d), lse_ca=new(B, heads, T), h2=new(R, d),
           nf=new(R, d), pre=new(R, d), u=new(R, d), h3=new(R, d), nn1=new(R, d), qkv2=new(R, 3 * d))
for v in out.values():
    v.fill_(float("nan"))
planes = {k: ops.pack_weight_traj(v) for k, v in W.items()}
pn = ops.pack_weight_traj(Wn)
amax = torch.zeros(7, 64, dtype=torch.int32, device=dev)
ops.train_layer_fwd(B, T, M, heads,
                    tensors=dict(h=h, qkv=qkv, kv=kv, b_o=b["o"], b_q=b["q"], b_oc=b["oc"], b_1=b["1"], b_2=b["2"], b_n=bn, n2_w=ln["2"][0], n2_b=ln["2"][1],
                                 n3_w=ln["3"][0], n3_b=ln["3"][1], nn_w=ln["n"][0], nn_b=ln["n"][1], **out),
                    weights=dict(w_o=planes["o"].data_ptr(), w_q=planes["q"].data_ptr(), w_oc=planes["oc"].data_ptr(), w_1=planes["1"].data_ptr(),
                                 w_2=planes["2"].data_ptr(), w_n=pn.data_ptr()),
                    p=p, seed=seed, sites=sites, amax=tuple(amax[i].data_ptr() for i in range(7)))
torch.cuda.synchronize()
for k in want:
    a, w_ = out[k].reshape(-1).double(), want[k].reshape(-1).double()
    bad = int((~torch.isfinite(a)).sum())
    err = float((a - w_).norm() / w_.norm()) if bad == 0 else float("nan")
    print(f"{k:8s} rel err {err:10.3e}  non-finite {bad}")
names = ("a_sa", "n2", "a_ca", "nf", "u", "nn1", "h3")
for i, k in enumerate(names):
    got = float(amax[i].view(torch.float32).max())
    print(f"amax {k:5s} {got:.6f} want {float(want[k].abs().max()):.6f}")

# ---- timing of the launch alone (B as given; 256 = one workgroup per CU)
import time  # noqa: E402


def launch():
    ops.train_layer_fwd(B, T, M, heads,
                        tensors=dict(h=h, qkv=qkv, kv=kv, b_o=b["o"], b_q=b["q"], b_oc=b["oc"], b_1=b["1"], b_2=b["2"], b_n=bn, n2_w=ln["2"][0], n2_b=ln["2"][1],
                                     n3_w=ln["3"][0], n3_b=ln["3"][1], nn_w=ln["n"][0], nn_b=ln["n"][1], **out),
                        weights=dict(w_o=planes["o"].data_ptr(), w_q=planes["q"].data_ptr(), w_oc=planes["oc"].data_ptr(), w_1=planes["1"].data_ptr(),
                                     w_2=planes["2"].data_ptr(), w_n=pn.data_ptr()),
                        p=p, seed=seed, sites=sites, amax=tuple(amax[i].data_ptr() for i in range(7)))


for _ in range(3):
    launch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    launch()
torch.cuda.synchronize()
print(f"train_layer_fwd: B={B} T={T} M={M} p={p}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per launch")
