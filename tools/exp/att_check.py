import os, sys, subprocess, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
if len(sys.argv) > 1:
    from soccerdiffusion_amd import ops
    torch.manual_seed(0)
    B, T, d, heads = 8, 100, 256, 4
    qkv = torch.randn(B, T, 3 * d, device="cuda")
    out, lse = ops.attention_lse(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], heads)
    q, k, v = (t.reshape(B, T, heads, 64).transpose(1, 2).double() for t in (qkv[..., :d], qkv[..., d:2*d], qkv[..., 2*d:]))
    s = q @ k.transpose(-1, -2) / 8.0
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, d)
    lref = torch.logsumexp(s, -1) / 0.6931471805599453
    print(sys.argv[1], "out err", float((out.double() - ref).abs().max()), "lse err", float((lse.double() - lref).abs().max()),
          "checksum", float(out.double().sum()))
else:
    for mode in ("f16", "f32"):
        env = dict(os.environ)
        if mode == "f32": env["SD_ATT_OP"] = "f32"
        subprocess.run([sys.executable, __file__, mode], env=env, check=True)
