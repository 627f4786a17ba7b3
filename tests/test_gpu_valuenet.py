"""HIP value net vs the oracle (bit-exact) and vs the reference's own Net outputs (1e-4), plus the torch backend."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("pk,ok", [("params", "out"), ("params2", "out2")])
def test_hip_valuenet(oracle, golden_dir, pk, ok):
    import torch
    from tetris_mcts_amd.model import Model_VV
    z = np.load(os.path.join(golden_dir, "ref_valuenet.npz"))
    m = Model_VV(backend="hip")
    m.set_flat_params(z[pk])
    states = torch.from_numpy(z["states"].reshape(-1, 200)).cuda()
    # ragged batch sizes, including 1 and a non multiple of any tile
    for B in (64, 1, 7, 33):
        v, var = m.inference_device(states[:B].contiguous())
        v, var = v.cpu().numpy(), var.cpu().numpy()
        ov, ovar = np.zeros(B, np.float32), np.zeros(B, np.float32)
        p = np.ascontiguousarray(z[pk])
        s = np.ascontiguousarray(z["states"].reshape(-1, 200)[:B])
        oracle.lib().orc_valuenet_forward(oracle.ptr(p), oracle.ptr(s), B, oracle.ptr(ov), oracle.ptr(ovar))
        assert v.tobytes() == ov.tobytes() and var.tobytes() == ovar.tobytes(), (B, np.abs(v - ov).max(), np.abs(var - ovar).max())
        ref = z[ok][:B]
        scale = max(1.0, float(z[pk][478339]) / 1000.0)
        assert np.abs(v - ref[:, 0]).max() <= TOL * max(1.0, float(z[pk][478338]) / 100.0)
        assert np.abs(var - ref[:, 1]).max() <= TOL * scale
    # the scratch needs no initial contents: garbage in it (the arrival counters of k_vn_fc1 live in the rows' padding) - e.g.
    # what a launch aborted between two arrivals would leave - does not reach the outputs
    m._scratch.view(torch.int32).random_(-2**31, 2**31 - 1)
    v, var = m.inference_device(states[:64].contiguous())
    ov, ovar = np.zeros(64, np.float32), np.zeros(64, np.float32)
    oracle.lib().orc_valuenet_forward(oracle.ptr(np.ascontiguousarray(z[pk])), oracle.ptr(np.ascontiguousarray(z["states"].reshape(-1, 200)[:64])),
                                      64, oracle.ptr(ov), oracle.ptr(ovar))
    assert v.cpu().numpy().tobytes() == ov.tobytes() and var.cpu().numpy().tobytes() == ovar.tobytes()
    # batch invariance: a state's output does not depend on its neighbours
    big = states.repeat(40, 1)[torch.randperm(64 * 40, device="cuda")]
    v1, r1 = m.inference_device(big.contiguous())
    v2, r2 = m.inference_device(big[:5].contiguous())
    assert torch.equal(v1[:5], v2) and torch.equal(r1[:5], r2)
    # the two tilings of the fc1 kernel (32-state tiles below 8192 states, 64-state tiles from there on: valuenet.hip k_vn_fc1)
    # are the same arithmetic: a large ragged batch against itself in pieces, and its first 64 states against the oracle's bits
    huge = states.repeat(141, 1)[torch.randperm(64 * 141, device="cuda")][:9001].contiguous()
    vh, rh = m.inference_device(huge)
    vh, rh = vh.clone(), rh.clone()
    for lo in range(0, 9001, 3000):
        vp, rp = m.inference_device(huge[lo:lo + 3000].contiguous())
        assert torch.equal(vh[lo:lo + 3000], vp) and torch.equal(rh[lo:lo + 3000], rp), lo
    s64 = np.ascontiguousarray(huge[:64].cpu().numpy())
    ov, ovar = np.zeros(64, np.float32), np.zeros(64, np.float32)
    oracle.lib().orc_valuenet_forward(oracle.ptr(np.ascontiguousarray(z[pk])), oracle.ptr(s64), 64, oracle.ptr(ov), oracle.ptr(ovar))
    assert vh[:64].cpu().numpy().tobytes() == ov.tobytes() and rh[:64].cpu().numpy().tobytes() == ovar.tobytes()


def test_torch_backend_within_tolerance(golden_dir):
    import torch
    from tetris_mcts_amd.model import Model_VV
    z = np.load(os.path.join(golden_dir, "ref_valuenet.npz"))
    m = Model_VV(backend="torch")
    m.set_flat_params(z["params"])
    v, var = m.inference_device(torch.from_numpy(z["states"].reshape(-1, 200)).cuda())
    assert np.abs(v.cpu().numpy() - z["out"][:, 0]).max() <= TOL
    assert np.abs(var.cpu().numpy() - z["out"][:, 1]).max() <= TOL
    out = m.inference(z["states"][:3, None].astype(np.float32))  # reference signature
    assert out[0].shape == (3, 1) and out[1].shape == (3, 1)


def test_fence_free_handoffs_under_stress(golden_dir):
    """k_vn_fc1's folded output layer (and k_dn_fc's hidden layer) hand a tile's partial results from workgroup to workgroup with
    write-through stores and an arrival counter, no release / acquire (valuenet.hip; the form MI355X_MICROARCH.md lists as
    valid - the acquire-release form measured + 2.4 us).  Every parity test holds the result to the oracle's bits once; this one
    holds it 4 000 times in a row, both tilings (32- and 64-state tiles), ragged sizes, while another stream keeps the chip busy
    with k_sim_step launches on an unrelated store: the outputs of every launch are the first launch's, bit for bit."""
    import torch
    from tetris_mcts_amd import agents, store as st
    from tetris_mcts_amd.model import Model_VV
    from tetris_mcts_amd.model_distributional import Model_Dist
    from tetris_mcts_amd.pyTetris import Tetris
    z = np.load(os.path.join(golden_dir, "ref_valuenet.npz"))
    m = Model_VV(backend="hip")
    m.set_flat_params(z["params"])
    base = torch.from_numpy(z["states"].reshape(-1, 200)).cuda()
    # the unrelated store: 512 games of plain UCT with rollouts (no evaluator), launched on a stream of its own
    env_args = ((20, 10), 1, 0, 0)
    game = Tetris(*env_args, seed=7, n_games=512)
    other = agents.Vanilla(sims=10, env=Tetris, env_args=env_args, n_games=512, max_nodes=4000)
    other.update_root(game)
    side = torch.cuda.Stream()
    both = st.SIM_BACKUP | st.SIM_FRONT
    other.store.move_begin(10 ** 6)
    torch.cuda.synchronize()
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    for n_states, rounds in ((9001, 1500), (4001, 2500)):          # 64-state tiles from 8192 states on, 32-state tiles below
        batch = base.repeat((n_states + 63) // 64, 1)[torch.randperm(64 * ((n_states + 63) // 64), device="cuda")][:n_states].contiguous()
        v0, r0 = m.inference_device(batch)
        v0, r0 = v0.clone(), r0.clone()
        v, r = torch.empty_like(v0), torch.empty_like(r0)
        for i in range(rounds):
            if i % 8 == 0:
                with torch.cuda.stream(side):
                    other.store.sim_step(both)
            m.inference_device(batch, v, r)
            bad += (v.view(torch.int32) != v0.view(torch.int32)).sum() + (r.view(torch.int32) != r0.view(torch.int32)).sum()
    # the distributional head's fc kernel: the same hand-off inside one workgroup's tile
    md = Model_Dist(atoms=50, seed=0, backend="hip")
    batch = base.repeat(65, 1)[:4099].contiguous()
    d0 = md.inference_device(batch).clone()
    out = torch.empty_like(d0)
    for i in range(1000):
        if i % 8 == 0:
            with torch.cuda.stream(side):
                other.store.sim_step(both)
        md.inference_device(batch, out)
        bad += (out.view(torch.int32) != d0.view(torch.int32)).sum()
    torch.cuda.synchronize()
    assert int(bad.item()) == 0
    assert (other.store.errors() & ~1).sum().item() == 0
