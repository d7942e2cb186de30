"""B = 256 rollout (north_star's shape) as k concurrent sub-batches on k streams inside one hipGraph: does the attention kernel
(HBM-bound) of one sub-batch overlap the layer kernel (issue-bound) of another when neither fills the chip?"""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from soccerdiffusion_amd import _lib, ops
from soccerdiffusion_amd.synthetic import synthetic_state_dict
D, L, T, J, MC, N = bench.D, bench.L, bench.T, bench.J, bench.MC, 50
dev = torch.device("cuda", 0)
sd = synthetic_state_dict(D, J, L, seed=7)
packed = ops.pack_denoiser(sd, dev, max_len=T)
ts = ops.ddim_timesteps(N); coef = ops.ddim_coefficients(ts, ops.alphas_cumprod(), N)
toks = ops.step_token(torch.tensor(ts, device=dev), ops.step_frequencies(D).to(dev), sd["step_encoding.token"].to(dev)).reshape(N, D).contiguous()
lib = _lib.load()
REPS = int(os.environ.get("SPLIT_REPS", "10"))
for B in tuple(int(b) for b in os.environ.get("SPLIT_B", "256,512").split(",")):
    x_T = torch.randn(B, T, J, device=dev); ctx = torch.randn(B, MC, D, device=dev)
    ref = ops.ddim_sample(packed, ctx, toks, coef, x_T)
    for k in tuple(int(v) for v in os.environ.get("SPLIT_K", "1,2,4").split(",")):
        bs = B // k
        xs = [torch.zeros(bs, T, J, device=dev) for _ in range(k)]
        cs = [ctx[i * bs:(i + 1) * bs].contiguous() for i in range(k)]
        wss = [torch.empty(lib.sd_workspace_floats(bs, T, MC, D, L, N), dtype=torch.float32, device=dev) for _ in range(k)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
        def run():
            cur = torch.cuda.current_stream()
            for i in range(k):
                st = cur if i == 0 else streams[i]
                if i: st.wait_stream(cur)
                with torch.cuda.stream(st):
                    xs[i].copy_(x_T[i * bs:(i + 1) * bs])
                    _lib.check(lib.sd_ddim_sample(C.byref(packed.struct), cs[i].data_ptr(), toks.data_ptr(), coef.ctypes.data_as(_lib.c_float_p),
                                                  xs[i].data_ptr(), None, wss[i].data_ptr(), bs, T, MC, N, st.cuda_stream), "sample")
            for i in range(1, k):
                cur.wait_stream(streams[i])
        side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        g.replay(); torch.cuda.synchronize()
        same = torch.equal(torch.cat(xs), ref)
        t0 = time.perf_counter()
        for _ in range(REPS):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / REPS
        print(f"B={B} streams={k}: {dt * 1e3:.2f} ms per rollout -> {B / dt:.0f} traj/s   (close to unsplit: {float((torch.cat(xs) - ref).abs().max()):.2e}, identical {same})", flush=True)
