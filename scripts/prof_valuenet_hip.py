"""Run only the HIP value net at B=4096 (for rocprofv3 --kernel-trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_mcts_amd.model import Model_VV
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
states = (torch.randint(0, 3, (B, 200), device="cuda") - 1).to(torch.int8)
m = Model_VV(backend="hip", seed=0)
for _ in range(30):
    m.inference_device(states)
torch.cuda.synchronize()
