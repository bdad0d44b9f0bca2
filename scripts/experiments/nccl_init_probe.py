import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t=torch.ones(4, device="cuda"); dist.all_reduce(t); dist.barrier()
objs=[None]; dist.all_gather_object(objs, "x")
print("ok", t.tolist(), objs, dist.get_backend(), dist.get_world_size())
dist.destroy_process_group()
