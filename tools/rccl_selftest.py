import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
dev=torch.device("cuda",0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
t=torch.ones(1<<20, device=dev); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
print("rccl ok", float(t[0]))
dist.destroy_process_group()
