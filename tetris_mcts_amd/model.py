"""Value network of the reference (model/model_vv.py:13-52, `Net`; `Model_VV.inference` 210-217) on the GPU.

Same architecture, same state_dict keys (`head.conv1.weight` ... `head.fc_out.bias`, `out_ubound`,
`out_lbound`) and the same checkpoint dict format (`model_state_dict`, model/model.py:152-160), so the
reference's checkpoints load unchanged.  fp32 end to end (outputs must stay within 1e-4 of the
reference's CPU result).  Two back ends:
  * "hip"   — tm_valuenet_forward: hand-written gfx950 kernels (fp32 MFMA), bit-identical to
              oracle/valuenet_oracle.c's fma chains;
  * "torch" — PyTorch-ROCm ops (MIOpen / rocBLAS), used for training and as a cross-check.
"""
import ctypes as C
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from .store import _p, _stream

SCRATCH_MFMA = 2064     # TM_VALUENET_SCRATCH_MFMA (include/tetris_mcts_hip.h): floats of scratch per state (no initial contents required)
PARAM_ORDER = ["head.conv1.weight", "head.conv1.bias", "head.conv2.weight", "head.conv2.bias", "head.conv3.weight",
               "head.conv3.bias", "head.fc1.weight", "head.fc1.bias", "head.fc_out.weight", "head.fc_out.bias",
               "out_ubound", "out_lbound"]
EXP_PATH = "./pytorch_model/"
variance_bound = 1e-1
_epoch = [0]


def next_weights_epoch():
    """A number no other set of evaluator weights in this process has had (>= 1): the tree engine files the evaluator's output
    per observation under it (tm_store::obs_eval) and uses only what was filed under the current one."""
    _epoch[0] += 1
    return _epoch[0]


class Net(nn.Module):
    def __init__(self, eps=variance_bound):
        super().__init__()
        act = nn.ReLU(inplace=True)
        self.head = nn.Sequential(OrderedDict([
            ("conv1", nn.Conv2d(1, 32, 3, 1, bias=True)), ("act1", act),
            ("conv2", nn.Conv2d(32, 32, 3, 1, bias=True)), ("act2", act),
            ("conv3", nn.Conv2d(32, 32, 3, 1, bias=True)), ("act3", act),
            ("flatten", nn.Flatten()),
            ("fc1", nn.Linear(32 * 14 * 4, 256)), ("fc_act1", act),
            ("fc_out", nn.Linear(256, 2)), ("act_out", nn.Sigmoid()),
        ]))
        self.out_ubound = nn.Parameter(torch.tensor([1e2, 1e3]), requires_grad=False)
        self.out_lbound = nn.Parameter(torch.tensor([0, eps]), requires_grad=False)

    def forward(self, x):
        return self.head(x) * self.out_ubound + self.out_lbound


class Model_VV:
    """Inference-side mirror of the reference's Model_VV (load / inference / training(False))."""

    def __init__(self, backend="hip", device="cuda", seed=None, **kwargs):
        if seed is not None:
            torch.manual_seed(seed)
        self.device = torch.device(device)
        self.model = Net().to(self.device).eval()
        self.backend = backend
        self._flat = None
        self._prepared = None
        self.weights_epoch = next_weights_epoch()
        self._scratch = None         # "hip": rows of SCRATCH_MFMA floats (k_vn_fc1's per-tile counters live in their padding; every evaluation clears them)
        self._scratch_plain = None   # "hip_plain": its own buffer - never handed to the matrix-core kernels

    def training(self, mode=True):
        self.model.train(mode)

    # ---- training side (model/model.py:97-249, model_vv.py:125-134,227-231) ----
    def _optimizer(self):
        if getattr(self, "optimizer", None) is None:
            from .train import Yogi
            # all 12 parameters in one group, out_ubound / out_lbound included, as model_vv.py:132 does
            # (Yogi(self.model.parameters(), ...)): a checkpoint's optimizer_state_dict then loads on either side
            self.optimizer = Yogi(self.model.parameters(), lr=1e-3, eps=1e-3, weight_decay=1e-3)
        return self.optimizer

    def train_data(self, data, **kwargs):
        """data: [states [n,1,20,10], values [n,1], variances [n,1], weights [n,1]] (numpy or tensors)."""
        from . import train as T
        dev = self.device
        data = [torch.as_tensor(d, dtype=torch.float32, device=dev) for d in data]
        with torch.no_grad():
            self.model.out_ubound.copy_(torch.stack([data[1].max(), data[2].max()]))   # model_vv.py:228-229
        best = {}

        def save():
            best["model"] = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
            from . import dist as tdist
            if tdist.rank() == 0:      # replicas are identical: one checkpoint file, written by rank 0
                self.save(verbose=False)

        def load():
            self.model.load_state_dict(best["model"])
        res = T.train_data(self.model, self._optimizer(), data, save=save, load=load, **kwargs)
        self.model.eval()
        self._flat = None
        self._prepared = None
        self.weights_epoch = next_weights_epoch()
        return res

    def load(self, filename=EXP_PATH + "model_checkpoint", verbose=True):
        if os.path.isfile(filename):
            if verbose:
                print("Loading model...", flush=True)
            ck = torch.load(filename, map_location=self.device)
            self.model.load_state_dict(ck["model_state_dict"])
            if ck.get("optimizer_state_dict"):
                try:
                    self._optimizer().load_state_dict(ck["optimizer_state_dict"])
                except (ValueError, KeyError) as e:
                    from sys import stderr
                    print("WARNING: optimizer state of %s not loaded (%s): the optimiser restarts with fresh moments"
                          % (filename, e), file=stderr, flush=True)
        elif verbose:
            print("Checkpoint not found, using default model", flush=True)
        self._flat = None
        self._prepared = None
        self.weights_epoch = next_weights_epoch()

    def save(self, filename=EXP_PATH + "model_checkpoint", verbose=True):
        if verbose:
            print("Saving model...", flush=True)
        os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
        # the reference's Model.load indexes optimizer_state_dict['param_groups'] (model.py:166-170): always write a real one
        torch.save({"model_state_dict": self.model.state_dict(), "optimizer_state_dict": self._optimizer().state_dict()},
                   filename)

    def set_flat_params(self, flat):
        """flat: 478342 floats in PARAM_ORDER (tests/golden/ref_valuenet.npz 'params')."""
        sd = self.model.state_dict()
        off = 0
        flat = torch.as_tensor(flat, dtype=torch.float32)
        for k in PARAM_ORDER:
            n = sd[k].numel()
            sd[k].copy_(flat[off:off + n].reshape(sd[k].shape))
            off += n
        self._flat = None
        self._prepared = None
        self.weights_epoch = next_weights_epoch()

    def flat_params(self):
        if self._flat is None:
            sd = self.model.state_dict()
            self._flat = torch.cat([sd[k].detach().reshape(-1).float() for k in PARAM_ORDER]).contiguous()
        return self._flat

    @torch.no_grad()
    def inference_device(self, states, v_out=None, var_out=None):
        """states: int8 [B,200] (or [B,20,10]) on the device -> (v[B], var[B]) float32 device tensors."""
        B = states.shape[0]
        if v_out is None:
            v_out = torch.empty(B, dtype=torch.float32, device=self.device)
            var_out = torch.empty(B, dtype=torch.float32, device=self.device)
        if self.backend == "hip":
            if self._scratch is None or self._scratch.shape[0] < B or self._scratch.shape[1] < SCRATCH_MFMA:
                self._scratch = torch.zeros(B, SCRATCH_MFMA, dtype=torch.float32, device=self.device)
            P = self.flat_params()
            if self._prepared is None:
                self._prepared = torch.empty(477184, dtype=torch.float32, device=self.device)
                _lib.check(_lib.lib().tm_valuenet_prepare(_p(P), _p(self._prepared), _stream()), "tm_valuenet_prepare")
            _lib.check(_lib.lib().tm_valuenet_forward(_p(P), _p(self._prepared), _p(states), B, _p(v_out), _p(var_out),
                                                      _p(self._scratch), _stream()), "tm_valuenet_forward")
        elif self.backend == "hip_plain":
            if self._scratch_plain is None or self._scratch_plain.shape[0] < B:
                self._scratch_plain = torch.empty(B, 9728, dtype=torch.float32, device=self.device)
            _lib.check(_lib.lib().tm_valuenet_forward_plain(_p(self.flat_params()), _p(states), B, _p(v_out), _p(var_out),
                                                            _p(self._scratch_plain), _stream()), "tm_valuenet_forward_plain")
        else:
            out = self.model(states.reshape(B, 1, 20, 10).float())
            v_out.copy_(out[:, 0])
            var_out.copy_(out[:, 1])
        return v_out, var_out

    @torch.no_grad()
    def hip_buffers(self, n_states):
        """(params, prepared operand streams, scratch for n_states) as ctypes pointers for the C ABI (search.hip)."""
        if self._scratch is None or self._scratch.shape[0] < n_states or self._scratch.shape[1] < SCRATCH_MFMA:
            self._scratch = torch.zeros(n_states, SCRATCH_MFMA, dtype=torch.float32, device=self.device)
        P = self.flat_params()
        if self._prepared is None:
            self._prepared = torch.empty(477184, dtype=torch.float32, device=self.device)
            _lib.check(_lib.lib().tm_valuenet_prepare(_p(P), _p(self._prepared), _stream()), "tm_valuenet_prepare")
        return _p(P), _p(self._prepared), _p(self._scratch)

    @torch.no_grad()
    def inference_requests(self, store):
        """Evaluate a TreeStore's pending leaf requests in place (fused render + forward, HIP back end only)."""
        import ctypes as C
        B = store.n_games * store.eval_slots
        if self._scratch is None or self._scratch.shape[0] < B or self._scratch.shape[1] < SCRATCH_MFMA:
            self._scratch = torch.zeros(B, SCRATCH_MFMA, dtype=torch.float32, device=self.device)
        P = self.flat_params()
        if self._prepared is None:
            self._prepared = torch.empty(477184, dtype=torch.float32, device=self.device)
            _lib.check(_lib.lib().tm_valuenet_prepare(_p(P), _p(self._prepared), _stream()), "tm_valuenet_prepare")
        _lib.check(_lib.lib().tm_valuenet_forward_requests(_p(P), _p(self._prepared), C.byref(store.s),
                                                           _p(self._scratch), _stream()), "tm_valuenet_forward_requests")

    def inference(self, batch):
        """Reference signature (model_vv.py:210-217): float array [B,1,20,10] -> [v[B,1], var[B,1]] numpy."""
        b = torch.as_tensor(batch).to(self.device)
        B = b.shape[0]
        v, var = self.inference_device(b.reshape(B, 200).to(torch.int8).contiguous())
        return [v.cpu().numpy().reshape(B, 1), var.cpu().numpy().reshape(B, 1)]
