"""The two summarisers that turn rocprofv3 csv output into what profiles/ holds (scripts/kernel_stats.py,
scripts/pmc_traffic.py) and bench.py's reader of the committed PMC file, on tiny synthetic traces."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, header, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(header)
        w.writerows(rows)


def test_kernel_stats_summary(tmp_path):
    hdr = ["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
           "LDS_Block_Size", "Scratch_Size"]
    rows = [["KERNEL_DISPATCH", "void tmcts::k_sim_step<false>(tm_store, int)", 100, 190, 64, 0, 112, 16896, 32],
            ["KERNEL_DISPATCH", "void tmcts::k_sim_step<false>(tm_store, int)", 300, 410, 64, 0, 112, 16896, 32],
            ["KERNEL_DISPATCH", "tmcts_vn::k_vn_fc1(float const*)", 500, 540, 52, 0, 32, 33792, 0]]
    _write(str(tmp_path / "run" / "1_kernel_trace.csv"), hdr, rows)
    out = tmp_path / "stats.csv"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "kernel_stats.py"), str(tmp_path / "run"), str(out)])
    got = list(csv.DictReader(open(out)))
    assert [g["Name"] for g in got] == ["void tmcts::k_sim_step<false>(tm_store, int)", "tmcts_vn::k_vn_fc1(float const*)"]
    assert got[0]["Calls"] == "2" and float(got[0]["AverageNs"]) == 100.0 and got[0]["MinNs"] == "90" and got[0]["MaxNs"] == "110"
    assert abs(float(got[0]["Percentage"]) - 100.0 * 200 / 240) < 1e-2 and got[1]["LDS"] == "33792"
    # --last N: only the last N launches of every kernel (the timed window of a bench run)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "kernel_stats.py"), str(tmp_path / "run"), str(out), "--last", "1"])
    got = list(csv.DictReader(open(out)))
    assert got[0]["Calls"] == "2" and got[0]["CallsAveraged"] == "1" and float(got[0]["AverageNs"]) == 110.0


def test_pmc_traffic_summary_and_bench_reader(tmp_path, monkeypatch):
    hdr = ["Kernel_Name", "Counter_Name", "Counter_Value"]
    for counter, vals in (("FETCH_SIZE", (10.0, 30.0)), ("WRITE_SIZE", (4.0, 8.0))):
        rows = [["void tmcts::k_sim_step<false>(tm_store, int)", counter, v] for v in vals]
        rows.append(["__amd_rocclr_copyBuffer", counter, 1.0])           # foreign kernels are left out
        _write(str(tmp_path / counter / "p_counter_collection.csv"), hdr, rows)
    oj, oc = tmp_path / "t.json", tmp_path / "t.csv"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "pmc_traffic.py"), str(oj), str(oc),
                           str(tmp_path / "FETCH_SIZE"), str(tmp_path / "WRITE_SIZE"), "--workload-key", "K1"])
    k = json.load(open(oj))["kernels"]
    assert list(k) == ["tmcts::k_sim_step<false>"]
    assert k["tmcts::k_sim_step<false>"]["FETCH_SIZE_KB_mean"] == 20.0 and k["tmcts::k_sim_step<false>"]["launches"] == 2
    # bench.py reads the same structure, and only for the workload the passes were recorded on
    sys.path.insert(0, ROOT)
    import bench
    os.makedirs(tmp_path / "profiles")
    os.replace(oj, tmp_path / "profiles" / "r05_pmc_traffic_of_this_test.json")      # (a file bench.PMC_FILE's pattern matches)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic(["tmcts::k_sim_step<false>"], "K1") == 1024.0 * (20.0 + 6.0)
    assert bench.pmc_traffic(["tmcts::k_sim_step<false>"], "K1", 2.0) == 1024.0 * (40.0 + 6.0)
    assert bench.pmc_traffic(["tmcts::k_sim_step<false>"], "another workload") is None
    assert bench.pmc_traffic(["tmcts_vn::k_vn_conv"], "K1") is None


def test_committed_bench_line_keeps_the_contract():
    """The driver's line as this round's final run printed it (profiles/r05_bench_valuesim_4096x500.json): the keys of the bench
    contract, the two objects of the hot-path tier (roofline, cpu_baseline) and this repo's additions, with consistent numbers."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_valuesim_4096x500.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] is not None
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    # the per-kernel figures of one simulation step times the simulations of a move fit inside the step
    per_sim = d["roofline"]["avg_launch_ms"] + d["roofline_other"]["avg_launch_ms"]
    assert per_sim * d["config"]["sims_per_move"] <= d["ms_per_step"]
    # expansions per second x seconds per step = expansions per step <= games x simulations
    assert d["value"] * d["ms_per_step"] * 1e-3 <= d["config"]["games_per_gpu"] * d["config"]["sims_per_move"]
    assert set(d["other_configs"]) == {"ValueSimLP", "DistValueSim", "Vanilla"} and all("error" not in v or v["error"] is None for v in d["other_configs"].values())
    assert d["steady_state"]["ms_per_step"] < 1.1 * d["ms_per_step"]        # the steady state within 10 % of the headline
    # the metric's second half: the same windows under the committed checkpoint, and every other line's number as a top-level scalar
    t = d["trained_net"]
    assert os.path.isfile(os.path.join(ROOT, t["checkpoint"]["file"])) and t["error_games"] == 0 and t["steady_state"]["error_games"] == 0
    assert t["steady_state"]["lines_per_1000_moves"] > 50 and t["mean_trace_len"] < d["mean_trace_len"]
    import hashlib
    assert hashlib.sha256(open(os.path.join(ROOT, t["checkpoint"]["file"]), "rb").read()).hexdigest()[:16] == t["checkpoint"]["sha256_16"]
    for key, src in (("steady_value", d["steady_state"]), ("lp_value", d["other_configs"]["ValueSimLP"]), ("dist_value", d["other_configs"]["DistValueSim"]),
                     ("vanilla_value", d["other_configs"]["Vanilla"]), ("trained_value", t)):
        assert d[key] == src["value"], key
    assert d["trained_steady_lines_per_1000_moves"] == t["steady_state"]["lines_per_1000_moves"]


def test_bench_gpus_n_without_n_devices_says_so():
    """`python bench.py --gpus N` on a box with fewer than N GPUs ends with a message that names both numbers (it used to die
    in an assert about --nproc-per-node); a rank count that differs from --gpus is named as well."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs 64 visible GPUs" in r.stderr and "Traceback" not in r.stderr
