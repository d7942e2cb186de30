import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(10, device=dev); dist.all_reduce(t); dist.barrier()
tt = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
print("rccl ok", float(t.sum()), float(tt)); dist.destroy_process_group()
