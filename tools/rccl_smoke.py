"""RCCL through torch.distributed with the calls bench.py makes at N > 1 (run under torch.distributed.run, any world size
the box has GPUs for): communicator bound to the rank's GPU, barrier, MAX and SUM all-reduce."""
import os, torch, torch.distributed as dist
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([float(dist.get_rank() + 1)], dtype=torch.float64).cuda()
dist.all_reduce(t, op=dist.ReduceOp.MAX)
u = torch.tensor([1.0, 2.0], dtype=torch.float64).cuda()
dist.all_reduce(u, op=dist.ReduceOp.SUM)
print("rank", dist.get_rank(), "of", dist.get_world_size(), "max", t.item(), "sum", u.cpu().tolist(), flush=True)
dist.barrier()
dist.destroy_process_group()
