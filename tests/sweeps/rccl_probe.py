"""One process, one GPU: bring RCCL up for a world of one through the bench's own init_groups and run the collectives the
N > 1 path uses on it (all_gather of counts inside gather_records is skipped at world 1, so they are called directly)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29577"), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
import numpy as np
import tomahawk_amd as T
from tomahawk_amd.dist import init_groups, gather_records
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
t0 = time.time()
group, xdev, desc = init_groups("nccl", dev, timeout_s=120)
print("init_groups ->", desc, xdev, f"{time.time() - t0:.1f}s", flush=True)
me = torch.tensor([0], dtype=torch.int64, device=xdev)
seen = [torch.zeros_like(me)]
dist.all_gather(seen, me, group=group)
cnt = torch.tensor([5], dtype=torch.int64, device=xdev)
g = [torch.zeros_like(cnt)]
dist.all_gather(g, cnt, group=group)
dist.barrier(group=group)
print("collectives on the RCCL group ok:", int(seen[0].item()), int(g[0].item()), flush=True)
dist.barrier(); dist.destroy_process_group()
