"""Can a captured hipGraph signal an outside stream in the middle of a replay?  (The data-parallel training step wants to start
layer l's gradient all-reduce while the graph still runs the backward of layers l-1 .. 0.)  Captures  A -> record(ev) -> B (long)
with an EXTERNAL event (an event-record node), replays, and lets a side stream wait for ev and run C: C must see A's result and
finish before B does.  Run once:  python tools/exp/graph_event.py"""
import inspect
import time

import torch

dev = torch.device("cuda", 0)
print("torch", torch.__version__, "Event signature:", inspect.signature(torch.cuda.Event.__new__) if hasattr(torch.cuda.Event, "__new__") else "?")
try:
    ev = torch.cuda.Event(external=True)
except TypeError as e:
    print("torch.cuda.Event(external=True) not supported:", e)
    raise SystemExit(0)
a = torch.zeros(1 << 20, device=dev)
big = torch.randn(4096, 4096, device=dev)
out = torch.zeros(4096, 4096, device=dev)
c = torch.zeros(1 << 20, device=dev)
side, cap = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    a.add_(1.0)
    for _ in range(20):
        out.copy_(big @ big)
    cap.synchronize()
    a.zero_()
    cap.synchronize()
    with torch.cuda.graph(g, stream=cap):
        a.add_(1.0)                  # A
        ev.record()                  # event-record node
        for _ in range(40):          # B: ~40 fp32 GEMMs
            out.copy_(big @ big)
torch.cuda.synchronize()
for trial in range(3):
    a.zero_()
    c.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        c.copy_(a)                   # C
        done_c = torch.cuda.Event()
        done_c.record()
    done_c.synchronize()
    t_c = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"trial {trial}: C saw a = {float(c[0].item())} (want 1.0), C done after {t_c * 1e3:.2f} ms, graph done after {t_all * 1e3:.2f} ms "
          f"-> {'OVERLAPPED' if t_c < 0.5 * t_all and float(c[0].item()) == 1.0 else 'no overlap / wrong'}")
