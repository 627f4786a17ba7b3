"""Distributional value head of the reference (model/model_distributional.py:18-57 `Net`, 60-107 `Model_Dist`;
BASELINE configs[4]): conv4x4(1->32) -> LeakyReLU -> conv4x4(32->32) -> LeakyReLU -> flatten -> FC 128 -> LeakyReLU ->
FC `atoms` -> softmax, on the reference's 22 x 10 input (the two rows that are hidden today included: the reference's
network was never moved to the 20-row engine, `convOutShape((22, 10), ...)` is hard-coded at
model_distributional.py:27).  Same module names, so `state_dict`s are interchangeable.  PyTorch-ROCm ops (MIOpen /
rocBLAS): this head is not on the benchmarked path - the reference's only agent that names it
(agents/DistValueSimOnline.py) does not import - so it gets the interface and the numerics, not a hand-written kernel.
The distribution arithmetic around it is in csrc/core_api.hip (tm_dist_transform, tm_dist_mean_variance)."""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv_out(shape, k, stride):
    return ((shape[0] - k) // stride + 1, (shape[1] - k) // stride + 1)


class Net(nn.Module):
    def __init__(self, input_shape=(22, 10), atoms=50):
        super().__init__()
        k, stride, filters, n_fc1 = 4, 1, 32, 128
        shape = _conv_out(_conv_out((22, 10), k, stride), k, stride)
        act = nn.LeakyReLU(inplace=True)
        self.seq = nn.Sequential(OrderedDict([
            ("conv1", nn.Conv2d(1, filters, k, stride)), ("act1", act),
            ("conv2", nn.Conv2d(filters, filters, k, stride)), ("act2", act),
            ("flatten", nn.Flatten()),
            ("fc1", nn.Linear(shape[0] * shape[1] * filters, n_fc1)), ("act3", act),
            ("fc_v", nn.Linear(n_fc1, atoms)),
        ]))

    def forward(self, x):
        return F.softmax(self.seq(x), 1)

    def log_prob(self, x):
        return F.log_softmax(self.seq(x), 1)


class Model_Dist:
    """inference(batch [B,1,22,10]) -> [dist [B, atoms]] (model_distributional.py:100-107); loss = the cross entropy
    against a target distribution, -value * (log p - log value) summed over atoms (model_distributional.py:86-98)."""

    def __init__(self, atoms=50, device=None, seed=None):
        if seed is not None:
            torch.manual_seed(seed)
        self.device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
        self.model = Net(atoms=atoms).to(self.device).eval()

    @torch.no_grad()
    def inference(self, batch):
        b = torch.as_tensor(batch, dtype=torch.float32, device=self.device)
        return [self.model(b).cpu().numpy()]

    def loss(self, state, value, weight=None):
        lp = self.model.log_prob(state)
        per = -value * (lp - value.log())
        if weight is not None:
            per = weight.squeeze() * per          # model_distributional.py:93-94
        std, mean = torch.std_mean(per.sum(dim=1))
        return mean, std
