"""Where a fit's time goes (one MI355X): train_data on a synthetic replay set of the online run's size, eager and replayed from
the graph, and its parts on their own (a validation pass, a checkpoint save).  python scripts/fit_timing.py [tuples]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_mcts_amd import model as M, train as T  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
rng = np.random.default_rng(0)
states = torch.from_numpy(rng.integers(-1, 2, size=(n, 1, 20, 10)).astype(np.float32)).cuda()
values = (states.sum(dim=(1, 2, 3)) * 0.5 + 20).reshape(-1, 1)
variances = torch.full((n, 1), 4.0, device="cuda")
weights = torch.from_numpy(rng.integers(10, 200, size=(n, 1)).astype(np.float32)).cuda()
os.chdir("/tmp")
if os.environ.get("TM_FIT_BENCHMARK") == "1":      # MIOpen's search for the convolutions' kernels instead of its default pick
    torch.backends.cudnn.benchmark = True
print("cudnn.benchmark (MIOpen find)", torch.backends.cudnn.benchmark, flush=True)
for mode in ("0", "1"):
    os.environ["TM_TRAIN_GRAPH"] = mode
    mdl = M.Model_VV(backend="torch", seed=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = mdl.train_data([states, values, variances, weights], iters_per_val=100, batch_size=1024, max_iters=1300, log=False,
                         early_stopping=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("graph=%s: %d iterations in %.3f s = %.3f ms per iteration (13 validations of %d rows and the saves included) %s" % (mode, res["iters"], dt, 1e3 * dt / res["iters"], n // 10, res), flush=True)
    val = [d[-n // 10:] for d in (states, values, variances, weights / weights.mean())]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): T.validation_loss(mdl.model, val, True)
    torch.cuda.synchronize(); print("   one validation pass: %.1f ms" % (1e3 * (time.perf_counter() - t0) / 5))
    t0 = time.perf_counter()
    for _ in range(5): mdl.save(verbose=False)
    print("   one checkpoint save: %.1f ms" % (1e3 * (time.perf_counter() - t0) / 5))
    # the iterations alone
    os.environ["TM_TRAIN_GRAPH"] = mode
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = mdl.train_data([states, values, variances, weights], iters_per_val=10 ** 9, batch_size=1024, max_iters=1000, log=False, early_stopping=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("   1000 iterations without validation: %.3f ms per iteration" % (1e3 * dt / 1000), flush=True)
