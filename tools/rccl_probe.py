"""One-rank RCCL check: process-group init on cuda:0, an all-reduce of the gradient vector size, a barrier."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29555")
os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
t=torch.ones(166676, device="cuda")
dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
print("nccl world=1 all_reduce ok", float(t.sum()))
dist.destroy_process_group()
