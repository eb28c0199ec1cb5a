"""RCCL smoke test for one rank per GPU: `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1
tools/rccl_selftest.py` (or bare, for a single rank). The same check runs in the preamble of every bench*.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metatrain_amd import distributed as D

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", os.environ["RANK"])
rank, local_rank, world = D.env_rank()
dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
torch.cuda.set_device(dev)
D.init("nccl", dev)
D.selftest(world, dev)
print(f"rccl ok: rank {rank} of {world} on {dev}")
torch.distributed.destroy_process_group()
