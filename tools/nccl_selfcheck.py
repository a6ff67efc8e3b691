"""tools/nccl_selfcheck.py -- the collectives bench.py issues at N > 1 (barrier, all_reduce MAX of one f64, all_reduce SUM of
five int64 mapstats) on a one-rank RCCL group: checks the calls, dtypes and the RCCL install on a 1-GPU box."""
import os, torch, torch.distributed as dist, numpy as np
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dist.barrier()
tt = torch.tensor([1.25], dtype=torch.float64, device="cuda")
st = torch.from_numpy(np.array([1,2,3,4,5], dtype=np.int64)).cuda()
dist.all_reduce(tt, op=dist.ReduceOp.MAX); dist.all_reduce(st, op=dist.ReduceOp.SUM)
print("nccl ok", float(tt.item()), st.cpu().numpy())
dist.destroy_process_group()
