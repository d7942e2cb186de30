"""Phase stamps (shader clock cycles) of workgroup 0 / wave 0 of the forward feed-forward chain, alone on the chip (R = 64) and in
a full launch (R = 25 600).  Needs the stamps build: AB_TU=sd_train_chain tools/ab_build.sh stamps -DSD_TC_STAMPS, then
SD_HIP_LIB=soccerdiffusion_amd/lib/variants/lib_stamps.so python tools/exp/chain_stamps.py"""
import ctypes, os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soccerdiffusion_amd import ops, _lib
lib = _lib.load()
d = 256
g = torch.Generator(device="cuda").manual_seed(0)
def rnd(*s, scale=1.0): return torch.randn(*s, device="cuda", generator=g) * scale
def planes(W):
    nb = W.shape[0] // d
    off = torch.arange(nb, dtype=torch.int64, device="cuda") * d * d
    f = torch.empty(2 * d * d * nb, dtype=torch.float16, device="cuda")
    ops.pack_weight_blocks(W.reshape(-1).contiguous(), off, nb, d, f)
    return f
s = 1 / math.sqrt(d)
Wo, W1, W2, Wn = (planes(rnd(n, d, scale=s)) for n in (d, d, d, 3 * d))
vec = lambda n: rnd(n, scale=0.1)
bo, b1, b2, bn, g3, be3, g1, be1 = vec(d), vec(d), vec(d), vec(3 * d), 1 + vec(d), vec(d), 1 + vec(d), vec(d)
p = float(os.environ.get("P", "0.1"))
names = ["start", "a loaded", "a planes", "gemm Wo", "epilogue h1 (+store)", "LN3 planes (+store n)", "gemm W1", "epilogue pre,gelu,u (+2 stores)",
         "u planes", "gemm W2", "epilogue h2 (+store)", "LN1' planes (+store)", "gemm qkv0", "store qkv0", "gemm qkv1", "store qkv1", "gemm qkv2", "store qkv2"]
for R in (64, 25600):
    a, h = rnd(R, d), rnd(R, d)
    o = {k: torch.empty(R, d, device="cuda") for k in ("h_out", "n_out", "pre", "u", "h2_out", "nn_out")}
    y = torch.empty(R, 3 * d, device="cuda")
    for _ in range(3):
        ops.train_fwd_chain(R, d, h, a=a, wo=Wo.data_ptr(), bo=bo, ln=(g3, be3), w1=W1.data_ptr(), b1=b1, w2=W2.data_ptr(), b2=b2,
                            nln=(g1, be1), wn=Wn.data_ptr(), bn=bn, n_next=3, y_out=y, p=p, seed=1, sites=(1, 2, 3), **o)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    lib.sd_tc_stamps.restype = ctypes.c_int
    assert lib.sd_tc_stamps(buf) == 0
    t = list(buf)[:18]
    print("R", R, "total cycles", t[17] - t[0])
    for i in range(1, 18):
        print("   %-36s %7d" % (names[i], t[i] - t[i - 1]))
