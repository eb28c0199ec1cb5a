import sys, os
sys.path.insert(0, os.getcwd())
import torch
from metatrain_amd import runtime as rt
from metatrain_amd.synthetic import random_box
dev = torch.device("cuda:0")
pos, z, cell = random_box(1000, seed=0)
posd = pos.to(dev)
for _ in range(30):
    p, _v = rt.neighbor_list(posd, cell, [True] * 3, 4.5)
torch.cuda.synchronize()
