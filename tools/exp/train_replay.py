"""The graphed training step: full call (stage inputs, upload hyper-parameters, replay, bookkeeping) against the bare graph replay."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
leg = bench.TrainLeg(torch.device("cuda", 0), 0, 1, 256, 400, dropout=0.1, graph=True)
for _ in range(6): leg.step()
torch.cuda.synchronize()
def t(f, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("full step %.3f ms" % t(leg.step))
g = leg.graphed
print("bare graph replay %.3f ms" % t(g.graph.replay))
print("stage inputs only %.3f ms" % t(lambda: g._stage(leg.x0, leg.ctx, None)))
print("full step again %.3f ms" % t(leg.step))
