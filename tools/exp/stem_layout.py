"""A/B: torch's 7x7 stem convolution (3 -> 64, stride 2) forward + weight gradient at 160 frames of 480 x 640, NCHW vs channels_last."""
import os, time, torch
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
torch.manual_seed(0)
conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
x = torch.rand(160, 3, 480, 640, device="cuda")
def run(cl):
    c = conv.to(memory_format=torch.channels_last) if cl else conv
    xi = x.contiguous(memory_format=torch.channels_last) if cl else x
    def step():
        c.weight.grad = None
        y = c(xi)
        if cl:
            yn = y.permute(0, 2, 3, 1)          # NHWC view, no copy
            assert yn.is_contiguous()
            g = torch.ones_like(yn).permute(0, 3, 1, 2)
        else:
            yn = y.permute(0, 2, 3, 1).contiguous()
            g = torch.ones_like(yn).permute(0, 3, 1, 2)
        y.backward(g)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3
print("NCHW ms", run(False), flush=True)
print("channels_last ms", run(True), flush=True)
