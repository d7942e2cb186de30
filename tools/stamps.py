"""Phase profile of decoder_layer_kernel from in-kernel shader-clock stamps (diagnostic build).

  tools/ab_build.sh stamps "-DSD_STAMPS"     # here
  gpurun -- 'SD_HIP_LIB=$PWD/soccerdiffusion_amd/lib/variants/lib_stamps.so python tools/stamps.py'

Runs 2 DDIM steps of the bench workload and prints, per layer launch, the median share of a workgroup's
lifetime spent between consecutive stamps (wave 0 of each of the workgroups)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from soccerdiffusion_amd import _lib, ops  # noqa: E402
from soccerdiffusion_amd.synthetic import synthetic_state_dict  # noqa: E402

NAMES = {1: "load a,h", 2: "K out-proj", 3: "bias+LDS+LN2", 4: "K scores", 5: "softmax+P", 6: "K PV'", 7: "bias+LDS+LN3",
         8: "K ffn1", 9: "gelu->LDS", 10: "K ffn2", 11: "bias+h store", 12: "LDS+LN1' | fc_out", 13: "K q", 14: "store q",
         15: "K k", 16: "store k", 17: "K v", 18: "store v"}


# last layer of a step with the next step's head merged in (stamps 13.. = the head's 1.. + 12)
TAIL_NAMES = {13: "x in LDS + emb weights", 14: "embed MFMA", 15: "+ bias + pe", 16: "h store + LDS + LN1", 17: "K q", 18: "store q",
              19: "K k", 20: "store k", 21: "K v", 22: "store v"}
HEAD_NAMES = {1: "x staging + emb weights", 2: "embed MFMA", 3: "+ bias + pe", 4: "h store + LDS + LN1", 5: "K q", 6: "store q",
              7: "K k", 8: "store k", 9: "K v", 10: "store v"}
# attention_f16_head_kernel (one workgroup per (sample, head); the default)
ATT_NAMES = {1: "loads issued", 2: "K/V landed, split, in LDS", 3: "Q split", 4: "barrier", 5: "S MFMAs", 6: "softmax", 7: "P split + PV",
             8: "normalise + stores issued"}
ATT_LV_NAMES = {1: "loads issued", 2: "K landed, split, in LDS", 3: "Q split", 4: "barrier", 5: "S MFMAs", 6: "barrier + V^T split, in LDS",
                7: "softmax", 8: "barrier + P split + PV", 9: "normalise + stores issued"}
if os.environ.get("SD_ATT16") != "stage2":
    ATT_NAMES = ATT_LV_NAMES
ATT_STREAM_NAMES = {1: "barrier", 2: "stage K/V", 3: "fetch issue + Q split", 4: "barrier", 5: "S MFMAs", 6: "softmax", 7: "P V", 8: "next unit"}


def main():
    B = int(os.environ.get("B", 4096))
    lib = _lib.load()
    fn = lib.sd_debug_set_stamps
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_long]
    D, L, T, J, MC = bench.D, bench.L, bench.T, bench.J, bench.MC
    sd = synthetic_state_dict(D, J, L, seed=0)
    packed = ops.pack_denoiser(sd, "cuda", max_len=T)
    n_steps = 2
    ts = ops.ddim_timesteps(50)[:n_steps]
    acp = ops.alphas_cumprod()
    coef = ops.ddim_coefficients(ops.ddim_timesteps(50), acp, 50)[: n_steps]
    freq = ops.step_frequencies(D).cuda()
    toks = ops.step_token(torch.tensor(ts).cuda(), freq, sd["step_encoding.token"].cuda()).reshape(n_steps, D)
    x = torch.randn(B, T, J, device="cuda")
    ctx = torch.randn(B, MC, D, device="cuda")
    wgs = (B * T + 63) // 64
    buf = torch.zeros(L + 3, max(wgs, B), 4, 32, dtype=torch.int64, device="cuda")   # slot L: head kernel, L+1: attention
    ops.ddim_sample(packed, ctx, toks, coef, x.clone())   # warm
    assert fn(buf.data_ptr(), max(wgs, B)) == 0
    ops.ddim_sample(packed, ctx, toks, coef, x.clone())
    torch.cuda.synchronize()
    fn(None, 0)
    st = buf.cpu().double()
    for l in (L - 2, L - 1, L, L + 1):
        s = st[l][: (B if l == L + 1 else wgs)]     # [workgroups][4][32]
        used = [i for i in range(30) if (s[:, :, i] > 0).all()]
        if not used:
            continue
        w0 = s[:, 0]
        life = s[:, :, used[-1]].max(1).values - s[:, :, used[0]].min(1).values
        if not used:
            continue
        print(f"{'head kernel' if l == L else 'attention_f16_head_kernel (last launch)' if l == L + 1 else 'layer %d' % l}: workgroup lifetime median {life.median():.0f} cycles")
        print(f"   {'phase (ends at stamp)':26s} {'min':>8s} {'p10':>8s} {'wave0 med':>10s} {'p90':>9s} {'share':>7s}   {'slowest-wave med':>16s}  {'skew at end med':>16s}")
        prev = used[0]
        for i in used[1:]:
            seg = w0[:, i] - w0[:, prev]
            seg_all = (s[:, :, i] - s[:, :, prev]).max(1).values
            skew = s[:, :, i].max(1).values - s[:, :, i].min(1).values
            names = HEAD_NAMES if l == L else ATT_NAMES if l == L + 1 else {**NAMES, **TAIL_NAMES} if l == L - 1 else NAMES
            print(f"   {names.get(i, str(i)):26s} {seg.min():8.0f} {seg.quantile(0.1):8.0f} {seg.median():10.0f} {seg.quantile(0.9):9.0f} "
                  f"{100 * seg.median() / life.median():6.1f}%   {seg_all.median():16.0f}  {skew.median():16.0f}")
            prev = i
        # MFMA-pipe occupancy per CU: union of the K-loop intervals of the workgroups that ran there
        kl = [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (12, 13), (14, 15), (16, 17)]
        kl = [(a, b) for a, b in kl if a in used and b in used]
        hw = s[:, 0, 30].long()
        xcc = s[:, 0, 31].long() & 0xF
        cu_key = (xcc << 16) | ((hw >> 8) & 0xFF) | (((hw >> 13) & 0x7) << 8)
        any_f, both_f, n_cu = [], [], 0
        for key in cu_key.unique():
            idx = (cu_key == key).nonzero().flatten()
            if len(idx) < 8:
                continue
            n_cu += 1
            ev = []
            for w in idx.tolist():
                for a, b in kl:
                    ev.append((float(w0[w, a]), 1)); ev.append((float(w0[w, b]), -1))
            ev.sort()
            t0 = sorted(float(w0[w, used[0]]) for w in idx.tolist())[2]     # skip the ramp: from the 3rd start ...
            t1 = sorted(float(w0[w, used[-1]]) for w in idx.tolist())[-3]   # ... to the 3rd-last end
            depth, last, t_any, t_both = 0, ev[0][0], 0.0, 0.0
            for t, dlt in ev:
                lo, hi = max(last, t0), min(t, t1)
                if hi > lo:
                    if depth >= 1: t_any += hi - lo
                    if depth >= 2: t_both += hi - lo
                depth += dlt
                last = t
            any_f.append(t_any / (t1 - t0)); both_f.append(t_both / (t1 - t0))
        import statistics
        print(f"   CUs seen {n_cu} ({len(cu_key.unique())} keys); time with >=1 workgroup in a K loop: {100 * statistics.median(any_f):.1f} % "
              f"(>=2: {100 * statistics.median(both_f):.1f} %)")


if __name__ == "__main__":
    main()
