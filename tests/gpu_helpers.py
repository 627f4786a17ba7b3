import numpy as np
import torch


def _s64(x):
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(x, n):
    return (x >> n) & ((1 << (64 - n)) - 1)


def _splitmix64(z):
    z = z + _s64(0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def hash_eval_torch(states):
    """oracle/agent_oracle.c orc_hash_eval on the device: states int8 [B,200] -> (v, var) float32."""
    w = states.contiguous().view(torch.int64)  # [B,25] little-endian words
    h = torch.full((w.shape[0],), _s64(0x9E3779B97F4A7C15), dtype=torch.int64, device=states.device)
    for j in range(25):
        h = _splitmix64(h ^ w[:, j])
    v = (h & 0xFFFF).to(torch.float32) * np.float32(40.0 / 65536.0)
    var = np.float32(0.1) + (_lsr(h, 16) & 0xFFFF).to(torch.float32) * np.float32(200.0 / 65536.0)
    return v, var
