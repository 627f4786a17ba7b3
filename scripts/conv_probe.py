"""The value net's two kernels on fixed inputs, N launches of B states (for rocprofv3 passes of the product and of variant
builds: TETRIS_MCTS_LIB=build_variants/X.so): python scripts/conv_probe.py [B] [N]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_mcts_amd.model import Model_VV  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1867
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
torch.manual_seed(0)
states = (torch.randint(0, 3, (B, 200), device="cuda") - 1).to(torch.int8)
m = Model_VV(backend="hip", seed=0)
for _ in range(N):
    v, var = m.inference_device(states)
torch.cuda.synchronize()
print("B", B, "launches", N, "checksum", float(v.double().sum()), float(var.double().sum()))
