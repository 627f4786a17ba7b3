"""The log lines the reference's dashboards parse (web/parseLog.py:58-66) keep their format: restated regexes (the
reference tree is not available where the tests run) against the lines our training loop / replay memory / play.py
print."""
import io
import re
from contextlib import redirect_stderr

import torch

SCORE_RE = (r'Episode:\s*(?P<episode>\d*)\s*Score:\s*(?P<score>\d*)\s*Lines Cleared:\s*(?P<lines>\d*)')          # parseLog.py:59-61
TRAIN_RE = (r'Iteration:\s*(?P<iter>\d*)\s*training loss:\s*(?P<t_loss>\d*\.\d*)\s*'
            r'validation loss:\s*(?P<v_loss>\d*\.\d*)±\s*(?P<v_loss_err>\d*\.\d*|nan)\s*'
            r'gradient norm:\s*(?P<g_norm>\d*\.\d*)')                                                           # parseLog.py:62-65
DATASIZE_RE = r'Training data size:\s*(?P<tsize>\d*)\s*Validation data size:\s*(?P<vsize>\d*)'                   # parseLog.py:66-67
QUEUE_RE = r'Memory usage: (?P<filled>\d*) / (?P<size>\d*).*'                                                    # parseLog.py:68


def test_training_lines_parse():
    from tetris_mcts_amd import train as T
    from tetris_mcts_amd.model import Net
    torch.manual_seed(0)
    net = Net()
    n = 300
    data = [torch.randint(-1, 2, (n, 1, 20, 10)).float(), torch.rand(n, 1) * 50, torch.rand(n, 1) * 100 + 1,
            torch.randint(1, 30, (n, 1)).float()]
    buf = io.StringIO()
    with redirect_stderr(buf):
        import importlib
        importlib.reload(T)          # train.py binds sys.stderr at import
        T.train_data(net, T.Yogi(net.parameters(), lr=1e-3, eps=1e-3), data, batch_size=32, iters_per_val=5, max_iters=10)
    importlib.reload(T)
    lines = buf.getvalue().splitlines()
    m = [re.search(DATASIZE_RE, l) for l in lines]
    assert any(m) and [x for x in m if x][0].groupdict() == dict(tsize="270", vsize="30")
    t = [re.search(TRAIN_RE, l) for l in lines]
    t = [x for x in t if x]
    assert len(t) == 2 and t[0].group("iter") == "5" and float(t[0].group("g_norm")) > 0


def test_replay_memory_lines_parse():
    from tetris_mcts_amd.replay import ReplayMemory
    mem = ReplayMemory(2, 100, 5, 10)
    out = []
    keys = torch.zeros(40, 12, dtype=torch.int32)
    stats = torch.ones(40, 4)
    assert mem.absorb(keys, stats, 0, log=out.append) is None
    m = re.search(QUEUE_RE, out[0])
    assert m and m.groupdict() == dict(filled="40", size="100")
    assert mem.absorb(keys, stats, 5, log=out.append) is not None
    assert any("proceed to training" in l for l in out)      # parseLog.py:113


def test_episode_line_parses():
    line = 'Episode: {:>5} Score: {:>10} Lines Cleared: {:>10}'.format(3, 1200, 17)     # play.py:164 of the reference
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "play.py")).read()
    assert "'Episode: {:>5} Score: {:>10} Lines Cleared: {:>10}'" in src
    assert re.search(SCORE_RE, line).groupdict() == dict(episode="3", score="1200", lines="17")
