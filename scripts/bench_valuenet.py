"""Microbenchmark of the value-net back ends (per-launch time, TFLOP/s vs the 157.3 TF fp32 matrix peak)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_mcts_amd.model import Model_VV  # noqa: E402

FLOP = 3803136
for B in (4096, 28672):
    states = (torch.randint(0, 3, (B, 200), device="cuda") - 1).to(torch.int8)
    for backend in ("hip", "torch", "hip_plain"):
        if backend == "hip_plain" and B > 4096:
            continue
        m = Model_VV(backend=backend, seed=0)
        for _ in range(3):
            m.inference_device(states)
        torch.cuda.synchronize()
        n = 20 if backend != "hip_plain" else 3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            m.inference_device(states)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("B=%d %-9s %.3f ms/launch  %.1f TFLOP/s (%.1f%% of 157.3)" % (B, backend, ms, FLOP * B / ms / 1e9, FLOP * B / ms / 1e9 / 157.3 * 100), flush=True)
