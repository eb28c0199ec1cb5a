"""The second-order training pass a few times on B x 10k-atom boxes (for rocprofv3 kernel traces)."""
import sys
import torch
sys.path.insert(0, ".")
exec(open("tools/gpu_train_bench.py").read().split("def timeit")[0])
u = torch.randn(boxes * natoms, 3, device=dev) * 1e-3
for _ in range(4):
    model.zero_grad()
    fw.forward()
    fw.backward_train2(seeds, seeds, u)
torch.cuda.synchronize()
