"""Times sd_op_attention_bwd (self: packed q|k|v, S = T = 100; cross: S = 11) at B = 256.  Env: SD_ATT_BWD=f32 selects the fp32-MFMA kernel."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import ops
B, T, d, H = 256, 100, 256, 4
g = torch.Generator(device="cuda").manual_seed(0)
for name, S in (("self", 100), ("cross", 11)):
    if S == T:
        qkv = torch.randn(B, T, 3 * d, device="cuda", generator=g)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        dqkv = torch.empty_like(qkv); dq, dk, dv = dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:]
    else:
        q = torch.randn(B, T, d, device="cuda", generator=g); kv = torch.randn(B, S, 2 * d, device="cuda", generator=g)
        k, v = kv[..., :d], kv[..., d:]
        dq = torch.empty_like(q); dkv = torch.empty_like(kv); dk, dv = dkv[..., :d], dkv[..., d:]
    out, lse = ops.attention_lse(q, k, v, H)
    dO = torch.randn(B, T, d, device="cuda", generator=g)
    for _ in range(3):
        ops.attention_bwd(q, k, v, out, dO, lse, dq, dk, dv, H)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.attention_bwd(q, k, v, out, dO, lse, dq, dk, dv, H)
    b.record(); torch.cuda.synchronize()
    print(name, "us per call", round(a.elapsed_time(b) / 20 * 1e3, 1), os.environ.get("SD_ATT_BWD"), flush=True)
