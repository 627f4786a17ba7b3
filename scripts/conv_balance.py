"""How the convolution kernel's time depends on the number of states per SIMD (dense batches of n states through
tm_valuenet_forward, rocprofv3-free: HIP events around 50 launches; conv + fc1 together, fc1 is the same for all n <= 4096)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_mcts_amd.model import Model_VV  # noqa: E402
m = Model_VV(backend="hip", seed=0)
for n in (1024, 2048, 3072, 3234, 3584, 4096):
    states = (torch.randint(0, 3, (n, 200), device="cuda") - 1).to(torch.int8)
    for _ in range(3):
        m.inference_device(states)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        m.inference_device(states)
    e1.record()
    torch.cuda.synchronize()
    print("n=%d dense states: %.1f us per forward (conv + fc1)" % (n, 1e3 * e0.elapsed_time(e1) / 50), flush=True)
