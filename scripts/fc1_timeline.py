"""k_vn_fc1's stations in time (a -DTM_FC1_TIMELINE build: scripts/build_variant.sh fc1tl valuenet.hip '1i #define TM_FC1_TIMELINE 1'):
thread 0 of every workgroup stamps s_memrealtime (100 MHz) at entry (0), prologue issued (1), first chunk staged (2), K loop done (3),
h stored and visible (4), arrived on the tile's counter (5), - the last workgroup of a tile only - the other parts loaded (6), outputs
written (7).  Printed: microseconds after the launch's first stamp, over the workgroups.
    TETRIS_MCTS_LIB=$PWD/build_variants/fc1tl.so python scripts/fc1_timeline.py [B=1867] [out.json]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_mcts_amd.model import Model_VV  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1867
torch.manual_seed(0)
states = (torch.randint(0, 3, (B, 200), device="cuda") - 1).to(torch.int8)
m = Model_VV(backend="hip", seed=0)
rows = []
for it in range(6):
    v, var = m.inference_device(states)
    torch.cuda.synchronize()
    sc = m._scratch[:B].view(torch.int32).cpu().numpy()
    tile = 64 if B >= 8192 else 32            # (valuenet.hip: the 64-state shape from 8 192 rows on - the leaf-parallel kinds' request slots)
    st = []
    for t0 in range(0, B, tile):
        for y in range(4):
            if t0 + y < B: st.append(sc[t0 + y, 2049:2057].astype(np.uint32).astype(np.int64))
    st = np.array(st)
    ref = st[:, 0].min()
    us = (st - ref) / 100.0
    last = st[:, 7] > st[:, 5]                # wrote the outputs in this launch
    pc = lambda x: [round(float(x.mean()), 2), round(float(np.percentile(x, 50)), 2), round(float(x.max()), 2)]
    row = {"workgroups": int(len(st)), "last_of_their_tile": int(last.sum()),
           "entry": pc(us[:, 0]), "prologue_issued": pc(us[:, 1]), "first_chunk_staged": pc(us[:, 2]), "k_loop_done": pc(us[:, 3]),
           "h_visible": pc(us[:, 4]), "arrived": pc(us[:, 5]),
           "others_loaded_last_only": pc(us[last, 6]) if last.any() else None, "outputs_written_last_only": pc(us[last, 7]) if last.any() else None,
           "k_loop_us": pc(us[:, 3] - us[:, 2]), "stage_first_chunk_us": pc(us[:, 2] - us[:, 1]), "prologue_us": pc(us[:, 1] - us[:, 0]),
           "epilogue_last_us": pc(us[last, 7] - us[last, 5]) if last.any() else None}
    row["entry_deciles_us"] = [round(float(np.percentile(us[:, 0], q)), 1) for q in range(0, 101, 10)]
    row["entry_by_y_mean_us"] = [round(float(us[y::4, 0].mean()), 1) for y in range(4)]
    cs = sc[:, 2057:2063].astype(np.uint32).astype(np.int64)        # the convolution's waves (one state each at these sizes): entry,
    cu = (cs - cs[:, 0].min()) / 100.0                                #  loop entered, input rendered, conv1, conv2, conv3 + stored
    row["conv"] = {"states": int(B), "entry": pc(cu[:, 0]), "constants_and_request_resolved": pc(cu[:, 1]), "input_rendered": pc(cu[:, 2]),
                   "conv1_done": pc(cu[:, 3]), "conv2_done": pc(cu[:, 4]), "conv3_stored": pc(cu[:, 5]),
                   "per_wave_us": {"prologue": pc(cu[:, 1] - cu[:, 0]), "render": pc(cu[:, 2] - cu[:, 1]), "conv1": pc(cu[:, 3] - cu[:, 2]),
                                   "conv2": pc(cu[:, 4] - cu[:, 3]), "conv3": pc(cu[:, 5] - cu[:, 4])}}
    if it >= 2:
        rows.append(row)
        print(json.dumps(row), flush=True)
if len(sys.argv) > 2:
    json.dump({"what": __doc__, "states": B, "columns": "mean, median, max over the workgroups", "rows": rows}, open(sys.argv[2], "w"), indent=1)
