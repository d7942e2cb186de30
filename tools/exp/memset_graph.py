"""Isolates the round-2 finding "hipMemsetAsync nodes replay garbage fill bytes from a captured graph" (DESIGN.md 5.7).

Captures three kinds of zero-fill of a device buffer into a hipGraph (torch.cuda.graphs), dirties the buffer and the caching
allocator's pool between replays, replays, and reports which form left anything but zeros:
  (a) torch.Tensor.zero_()                    - what optimizer.zero_grad / torch.zeros issue (a fill KERNEL on ROCm torch builds)
  (b) hipMemsetAsync through ctypes           - a genuine memset NODE in the graph (what the library used before round 2)
  (c) the library's zero_fill_kernel           - via sd_ddim_sample's own prologue is not reachable alone; (a) covers kernel fills
Run ONCE on the GPU box:  python tools/exp/memset_graph.py        (do not loop it)
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemsetAsync.restype = C.c_int
hip.hipMemsetD32Async.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemsetD32Async.restype = C.c_int


def trial(name, fill, n_words, replays=6):
    dev = torch.device("cuda", 0)
    buf = torch.full((n_words,), 0x58585858, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fill(buf, side)                      # warm-up outside capture
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            fill(buf, side)
    bad = []
    for r in range(replays):
        buf.fill_(0x7F7F7F7F if r % 2 else -0x7F7F7F80)   # dirty the target
        junk = [torch.full((1 << 18,), 0x5A5A5A5A + r, dtype=torch.int32, device=dev) for _ in range(8)]   # churn the pool
        del junk
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        nz = int((buf != 0).sum().item())
        if nz:
            bad.append((r, nz, hex(int(buf[buf != 0][0].item()) & 0xFFFFFFFF)))
    print(f"{name:58s} words={n_words:8d}  {'OK: zeros after every replay' if not bad else 'GARBAGE: ' + str(bad)}")
    return not bad


def main():
    assert torch.cuda.is_available()
    print("torch", torch.__version__, "hip", torch.version.hip)
    ok = True
    for n in (1, 40, 1 << 16):
        ok &= trial("(a) Tensor.zero_() captured", lambda b, s: b.zero_(), n)
        ok &= trial("(b) hipMemsetAsync(ptr, 0, bytes, stream) captured",
                    lambda b, s: hip.hipMemsetAsync(b.data_ptr(), 0, b.numel() * 4, s.cuda_stream), n)
        ok &= trial("(b') hipMemsetD32Async(ptr, 0, words, stream) captured",
                    lambda b, s: hip.hipMemsetD32Async(b.data_ptr(), 0, b.numel(), s.cuda_stream), n)
    print("verdict:", "no form misbehaved in this run" if ok else "see GARBAGE lines above")


if __name__ == "__main__":
    main()
