import os
os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT", NCCL_DEBUG_FILE="/tmp/rr_%d.log" % os.getpid(), MASTER_ADDR="127.0.0.1", MASTER_PORT="29777", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(4, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize()
p = "/tmp/rr_%d.log" % os.getpid()
print("exists", os.path.exists(p), os.path.getsize(p) if os.path.exists(p) else -1)
import glob; print(glob.glob("/tmp/rr_*"))
dist.destroy_process_group()
print("after destroy", os.path.exists(p), os.path.getsize(p) if os.path.exists(p) else -1)
