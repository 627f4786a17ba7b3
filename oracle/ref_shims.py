"""ORACLE — test infrastructure only.

Import the reference's OWN Python agents (agents/Vanilla.py, ValueSim.py, ValueSimLP.py) straight
from /root/reference, unmodified, on top of:
  * `pyTetris`                -> oracle/_build/pyTetris*.so (ENGINE_SPEC.md engine; the real one is absent)
  * `agents.cppmodule.core`   -> oracle/_ref/core*.so  (the reference's core.cpp compiled in place)
  * `agents.cppmodule.agent`  -> oracle/_ref/agent*.so (the reference's agent.cpp compiled in place)
  * `numba`, `cppimport`      -> inert stand-ins (absent packages; only decorators / a build trigger)
Only usable where /root/reference exists (this container).  Used by tests/golden/make_golden.py.
"""
import ctypes
import os
import sys
import types

import numpy as np

from . import binding

REF = os.environ.get("TETRIS_MCTS_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "agents", "agent.py"))


def install():
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    binding.build(ref=True)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")

        def _decorator(*args, **kwargs):
            if len(args) == 1 and callable(args[0]) and not kwargs:
                return args[0]
            return lambda f: f
        for name in ("jit", "njit", "vectorize", "guvectorize"):
            setattr(nb, name, _decorator)
        nb.float32 = nb.float64 = nb.int32 = nb.int64 = nb.boolean = None
        sys.modules["numba"] = nb
    if "cppimport" not in sys.modules:
        ci = types.ModuleType("cppimport")
        ci.imp = lambda name: None
        sys.modules["cppimport"] = ci
    # The reference's own modules say `from pyTetris import Tetris`: in a process that runs them, that name is the oracle's
    # engine - explicitly, whatever was imported before (the product has a module of the same name).
    sys.modules["pyTetris"] = binding.oracle_pytetris()
    import agents  # namespace package under /root/reference
    import agents.cppmodule
    for name in ("core", "agent"):
        mod = binding.load_ref_native(name)
        if mod is None:
            raise RuntimeError("oracle/_ref/%s not built" % name)
        sys.modules["agents.cppmodule." + name] = mod
        setattr(agents.cppmodule, name, mod)


_libc = ctypes.CDLL("libc.so.6")


def srand(seed=1):
    """The reference's check_low draws from the process-global libc rand() (core.h:62,76)."""
    _libc.srand(seed)


def make_agent(name, sims, max_nodes=None, evaluator=None, **kwargs):
    """Construct reference agent `name` ('ValueSim' | 'ValueSimLP' | 'Vanilla') the way play.py:83-91 does.

    max_nodes: shrink the node pool after construction (ValueSim hard-codes 100000, ValueSim.py:16) by
               re-running the reference's own init_array() with a smaller self.max_nodes.
    evaluator: callable(states int8[k,20,10]) -> (v float32[k], var float32[k]); replaces the torch model's
               inference() so both sides of a parity test see bit-identical leaf values.  Scalars handed to
               ValueSim are Python floats so `score + v` is float64 as under numpy 1.17 (SURVEY 8c trap).
    """
    install()
    from importlib import import_module
    Tetris = binding.oracle_pytetris().Tetris
    mod = import_module("agents." + name)
    cls = getattr(mod, name)
    env_args = kwargs.pop("env_args", ((20, 10), 1, 0, 0))
    agent = cls(sims=sims, env=Tetris, env_args=env_args, benchmark=kwargs.pop("benchmark", False),
                online=kwargs.pop("online", False), min_visit=kwargs.pop("min_visit", 40), **kwargs)
    if max_nodes is not None and max_nodes != agent.max_nodes:
        agent.max_nodes = max_nodes
        agent.init_array()
    if evaluator is not None and hasattr(agent, "model"):
        def inference(batch):
            b = np.asarray(batch)
            k = b.shape[0]
            v, var = evaluator(b.reshape(k, 20, 10).astype(np.int8))
            return [np.asarray(v, np.float32).reshape(k, 1), np.asarray(var, np.float32).reshape(k, 1)]
        agent.model.inference = inference

        def evaluate_state(state):
            v, var = inference(state[None, None, :, :])
            return float(v[0][0]), float(var[0][0])
        agent.evaluate_state = evaluate_state
    return agent
