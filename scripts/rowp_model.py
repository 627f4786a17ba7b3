"""Row P (north_star: "one persistent wavefront per concurrent game"; the verdict's item 3): what a schedule WITHOUT the launch-wide
barrier between tree kernel, convolution and fc1 could gain, from measurements only (no GPU time):
  profiles/r05_rowp_timeline_*.json   rocprofv3 kernel traces of the headline workload as one batch and as two sub-batches on two
                                      streams, with the stock value net and with the co-residency-shaped one (scripts/timeline.py)
  profiles/r05_wave_cycles_before.npz per-wave phase cycles of k_sim_step (scripts/wave_cycles.py)
-> profiles/r05_rowp_model.json and a table on stdout."""
import json
import os

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(R, "profiles", n)  # noqa: E731
one = json.load(open(P("r05_rowp_timeline_one_batch.json")))
two = json.load(open(P("r05_rowp_timeline_two_sub_batches.json")))
two_c = json.load(open(P("r05_rowp_timeline_two_sub_batches_coresident_valuenet.json")))
one_c = json.load(open(P("r05_rowp_timeline_one_batch_coresident_valuenet.json")))
w = np.load(P("r05_wave_cycles_before.npz"))
wave = (w["CYC_BACK"] + w["CYC_SELECT"] + w["CYC_EXPAND"])[5:].astype(np.float64) / 2400.0      # us at 2.4 GHz
k = lambda d, n: d["kernels"][n]["avg_us"]  # noqa: E731
names = ("k_sim_step", "k_vn_conv", "k_vn_fc1")
out = {"kernel_us": {}, "notes": []}
for tag, d in (("one batch of 4096 games", one), ("two sub-batches of 2048, two streams", two),
               ("two sub-batches, value net shaped for co-residency (84 KB LDS: one conv workgroup per CU)", two_c),
               ("one batch, co-residency-shaped value net", one_c)):
    out["kernel_us"][tag] = {n: k(d, n) for n in names}
    out["kernel_us"][tag].update(sum=sum(k(d, n) for n in names), two_or_more_kernels_running=d["two_or_more"],
                                 tree_overlapped_with_evaluator=d["tree_time_overlapped_with_evaluator"], idle=d["idle"])
lock = sum(k(one, n) for n in names)
half = sum(k(two, n) for n in names)
# a kernel's duration = a floor (its slowest wave / workgroup chain) + a part that shrinks with the batch: two points each
fit = {}
for n, units in (("k_sim_step", (4096, 2048)), ("k_vn_conv", (1867, 934)), ("k_vn_fc1", (59, 30))):
    b = (k(one, n) - k(two, n)) / (units[0] - units[1])
    fit[n] = {"floor_us": k(one, n) - b * units[0], "us_per_unit": b, "units": {"k_sim_step": "games", "k_vn_conv": "states", "k_vn_fc1": "32-state tiles"}[n]}
out["fit"] = fit
out["lockstep_us_per_simulation"] = lock
out["two_streams_us_per_simulation"] = half          # each stream advances its half by one simulation in this time, both at once
out["two_streams_gain"] = 1.0 - half / lock
# the barrier-free bound: every game advances at its own pace = its own wave + the evaluator's latency for ONE state
conv_alone, fc1_tile = 19.9 + 10.0, 12.0 + 10.0      # MFMA issue of one state + render/conv1/epilogue; matrix time of a tile + staging/arrival
out["per_game_chain_us"] = {"mean_wave": float(wave.mean()), "p99_wave": float(np.percentile(wave, 99)), "max_wave": float(wave.max()),
                            "conv_one_state": conv_alone, "fc1_one_tile": fc1_tile,
                            "sum": float(wave.mean()) + conv_alone + fc1_tile}
out["notes"] = [
    "the kernels of two sub-batches DO run at the same time (two or more kernels %.0f %% of the window, tree kernel under evaluator kernels %.0f %%) - "
    "co-residency is there, with the stock value net and with the shaped one" % (100 * two["two_or_more"], 100 * two["tree_time_overlapped_with_evaluator"]),
    "but a kernel over half the games takes as long as over all of them: every one of the three is a chain of dependent round trips "
    "(floor) plus a small part that scales - see fit - so two half-batches side by side cost what one batch costs (%.1f vs %.1f us per simulation, %.1f %%)"
    % (half, lock, 100 * out["two_streams_gain"]),
    "the shaped value net (one conv workgroup per CU so that tree workgroups fit beside it) is slower by itself (conv %.1f vs %.1f us for the whole batch): one wave per SIMD "
    "leaves the matrix pipe idle through a wave's render / conv1 / epilogue" % (k(one_c, "k_vn_conv"), k(one, "k_vn_conv")),
    "a schedule with no barrier at all is bounded by a game's own chain (mean wave + one state's convolution + one tile's fc1 = %.0f us per simulation, "
    "%.0f %% of the lockstep's %.0f us) but needs 4096 tree waves (4 x 96 registers per SIMD lane), the convolution's waves (120) and fc1's workgroups "
    "(2 x 152) resident together: 384 + 120 + 304 > 512 registers - fc1 does not fit beside the other two, and the collectors' hand-over relies on kernel boundaries"
    % (out["per_game_chain_us"]["sum"], 100 * out["per_game_chain_us"]["sum"] / lock, lock)]
json.dump(out, open(P("r05_rowp_model.json"), "w"), indent=1)
for tag, v in out["kernel_us"].items():
    print("%-95s tree %5.1f conv %5.1f fc1 %5.1f sum %6.1f  >=2 kernels %4.0f%%" % (tag, v["k_sim_step"], v["k_vn_conv"], v["k_vn_fc1"], v["sum"], 100 * v["two_or_more_kernels_running"]))
for n, f in fit.items():
    print("  %-10s floor %5.1f us + %.4f us per %s" % (n, f["floor_us"], f["us_per_unit"], f["units"]))
print("lockstep %.1f us/sim, two streams %.1f (gain %.1f %%), barrier-free bound %.1f" % (lock, half, 100 * out["two_streams_gain"], out["per_game_chain_us"]["sum"]))
