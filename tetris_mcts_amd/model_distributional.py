"""Distributional value head of the reference (model/model_distributional.py:18-57 `Net`, 60-107 `Model_Dist`;
BASELINE configs[4]): conv4x4(1->32) -> LeakyReLU -> conv4x4(32->32) -> LeakyReLU -> flatten -> FC 128 -> LeakyReLU ->
FC `atoms` -> softmax, on the reference's 22 x 10 input (the two rows that are hidden today included: the reference's
network was never moved to the 20-row engine, `convOutShape((22, 10), ...)` is hard-coded at
model_distributional.py:27).  Same module names, so `state_dict`s are interchangeable.  Two back ends, as model.Model_VV:
  * "hip"   - csrc/distnet.hip: hand-written gfx950 kernels on the fp32 matrix cores (wave-per-state convolutions, a batched
              FC + softmax kernel), bit-identical to oracle/distnet_oracle.c's fma chains, driven by the native launch loop
              (csrc/search.hip) when it is the leaf evaluator of DistValueSim;
  * "torch" - PyTorch-ROCm ops (MIOpen / rocBLAS): training (`loss`) and a cross-check.
The distribution arithmetic around it is in csrc/tree.hip (wave_dist_front / wave_dist_back)."""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

PARAM_ORDER = ["seq.conv1.weight", "seq.conv1.bias", "seq.conv2.weight", "seq.conv2.bias", "seq.fc1.weight", "seq.fc1.bias",
               "seq.fc_v.weight", "seq.fc_v.bias"]
PREPARED = 278528       # TM_DISTNET_PREPARED
SCRATCH = 2048          # TM_DISTNET_SCRATCH
ROW = 64                # TM_DIST_ROW


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

    def __init__(self, atoms=50, device=None, seed=None, backend=None):
        if seed is not None:
            torch.manual_seed(seed)
        self.device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
        self.atoms = int(atoms)
        self.model = Net(atoms=atoms).to(self.device).eval()
        self.backend = backend or ("hip" if self.device.type == "cuda" else "torch")
        if self.backend == "hip" and not (0 < self.atoms <= ROW):
            raise ValueError("the HIP head holds 1..64 atoms")
        self._flat = self._prepared = self._scratch = None

    def training(self, mode=True):
        self.model.train(mode)

    def weights_changed(self):
        """call after the parameters were written (an optimiser step, load_state_dict): the HIP operand streams are rebuilt"""
        self._flat = self._prepared = None

    def flat_params(self):
        if self._flat is None:
            sd = self.model.state_dict()
            self._flat = torch.cat([sd[k].detach().reshape(-1).float() for k in PARAM_ORDER]).contiguous()
        return self._flat

    def set_flat_params(self, flat):
        """flat: TM_DISTNET_PARAMS(atoms) floats in PARAM_ORDER"""
        sd = self.model.state_dict()
        off = 0
        flat = torch.as_tensor(flat, dtype=torch.float32)
        for k in PARAM_ORDER:
            n = sd[k].numel()
            sd[k].copy_(flat[off:off + n].reshape(sd[k].shape))
            off += n
        assert off == flat.numel()
        self.weights_changed()

    @torch.no_grad()
    def hip_buffers(self, n_states):
        """(params, prepared operand streams, scratch for n_states) as ctypes pointers for the C ABI (search.hip)."""
        from . import _lib
        from .store import _p, _stream
        if self._scratch is None or self._scratch.shape[0] < n_states:
            self._scratch = torch.empty(n_states, SCRATCH, dtype=torch.float32, device=self.device)
        P = self.flat_params()
        if self._prepared is None:
            self._prepared = torch.empty(PREPARED, dtype=torch.float32, device=self.device)
            _lib.check(_lib.lib().tm_distnet_prepare(_p(P), _p(self._prepared), _stream()), "tm_distnet_prepare")
        return _p(P), _p(self._prepared), _p(self._scratch)

    @torch.no_grad()
    def inference_device(self, states, out=None):
        """states: int8 [B,200] (or [B,20,10]) on the device, the 20 visible rows -> float32 [B, 64] (atoms first, zero padded
        rows are NOT guaranteed: only [:, :atoms] is written)."""
        from . import _lib
        from .store import _p, _stream
        B = states.shape[0]
        if out is None:
            out = torch.zeros(B, ROW, dtype=torch.float32, device=self.device)
        if self.backend == "hip":
            P, prep, scr = self.hip_buffers(B)
            st = states.reshape(B, 200).to(torch.int8).contiguous()
            _lib.check(_lib.lib().tm_distnet_forward(P, prep, _p(st), B, self.atoms, _p(out), out.stride(0), scr, _stream()),
                       "tm_distnet_forward")
        else:
            x = torch.zeros(B, 1, 22, 10, dtype=torch.float32, device=self.device)
            x[:, 0, 2:, :] = states.reshape(B, 20, 10).float()   # the reference's net sees 22 rows (model_distributional.py:27)
            out[:, :self.atoms].copy_(self.model(x))
        return out

    @torch.no_grad()
    def inference_requests(self, store):
        """Evaluate a TreeStore's pending leaf requests into its eval_dist (fused render + forward, HIP back end only)."""
        import ctypes as C
        from . import _lib
        from .store import _stream
        P, prep, scr = self.hip_buffers(store.n_games)
        _lib.check(_lib.lib().tm_distnet_forward_requests(P, prep, C.byref(store.s), scr, _stream()), "tm_distnet_forward_requests")

    @torch.no_grad()
    def inference(self, batch):
        """Reference signature (model_distributional.py:100-107): float array [B,1,22,10] -> [dist [B, atoms]] numpy."""
        b = torch.as_tensor(batch, dtype=torch.float32, device=self.device)
        if self.backend == "hip" and bool((b[:, 0, :2] == 0).all().item()):
            out = self.inference_device(b[:, 0, 2:, :].reshape(b.shape[0], 200).to(torch.int8))
            return [out[:, :self.atoms].cpu().numpy()]
        return [self.model(b).cpu().numpy()]     # something in the two hidden rows: only the torch ops take 22 rows

    def loss(self, state, value, weight=None):
        """model_distributional.py:84-98: -value * (log p - log value) summed over the atoms = KL(value || p), optionally
        weighted.  The reference's expression is NaN wherever a target atom is exactly 0 (0 * -inf; its targets would be the
        search's shifted distributions, whose lowest bins are empty): value * log value is taken as 0 there (torch.xlogy), the
        same number everywhere else."""
        lp = self.model.log_prob(state)
        per = torch.xlogy(value, value) - value * lp
        if weight is not None:
            per = weight.reshape(-1, 1) * per     # model_distributional.py:93-94 (weight.squeeze() broadcasts over the LAST axis
                                                  # there, which only lines up for batch == atoms; per-sample weights are what is meant)
        std, mean = torch.std_mean(per.sum(dim=1))
        return mean, std

    def _optimizer(self):
        if getattr(self, "optimizer", None) is None:
            # Model_Dist.__init__ (model_distributional.py:66): what the class ends up with (its _init_model's Yogi is overwritten)
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=1e-4, eps=1e-5, amsgrad=True)
        return self.optimizer

    def train_data(self, data, **kwargs):
        """data: [states [n,1,22,10], distributions [n,atoms], weights [n,1]]; Model.train_data (model/model.py:176-249) with
        this class's loss; data-parallel over the ranks as train.train_data describes."""
        from . import train as T
        data = [torch.as_tensor(d, dtype=torch.float32, device=self.device) for d in data]
        best = {}

        def save():
            best["model"] = {k: v.detach().clone() for k, v in self.model.state_dict().items()}

        def load():
            self.model.load_state_dict(best["model"])

        def loss_fn(net, batch, weighted):
            return self.loss(batch[0], batch[1], batch[2] if weighted else None)
        res = T.train_data(self.model, self._optimizer(), data, save=save, load=load, loss_fn=loss_fn, **kwargs)
        self.model.eval()
        self.weights_changed()
        return res
