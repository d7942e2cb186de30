"""What this box's HBM delivers to plain elementwise kernels (a reference point for the convolution kernels' memory phases):
torch copy / add of 768-MB fp32 tensors (the size of a layer-1 activation at 160 frames of 480 x 640)."""
import time, torch
n = 160 * 120 * 160 * 64
a, b, c = (torch.rand(n, device="cuda") for _ in range(3))
def timed(fn, k=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
t = timed(lambda: b.copy_(a)); print(f"copy  (1 read + 1 write): {t * 1e3:.3f} ms  {2 * 4 * n / t / 1e12:.2f} TB/s")
t = timed(lambda: torch.add(a, b, out=c)); print(f"add   (2 reads + 1 write): {t * 1e3:.3f} ms  {3 * 4 * n / t / 1e12:.2f} TB/s")
t = timed(lambda: torch.relu_(c)); print(f"relu_ (1 read + 1 write, in place): {t * 1e3:.3f} ms  {2 * 4 * n / t / 1e12:.2f} TB/s")
