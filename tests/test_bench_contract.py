"""The benchmark line's contract, checked on the lines this tree's profiles/ holds (each written by bench.py on an
MI355X; tools/profile_round.sh): the keys a reader of the line relies on exist, have the types and units stated, and
the derived figures follow from the primary ones.  No GPU needed: the lines are data."""
import glob
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(REPO, "profiles", "r03*_bench*.json")))


def _load(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_lines_exist():
    assert len(LINES) >= 3


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_bench_line_contract(path):
    d = _load(path)
    assert d["metric"] == "RANSAC 8-pt hypotheses scored/sec @N=4096 matches" and d["unit"] == "hypotheses/s"
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    cfg = d["config"]
    assert "config2" in cfg["workload"] and cfg["n_matches"] == 4096 and cfg["hypotheses"] == 2048
    assert "model" not in cfg
    # value = hypotheses of one step / time of one step (one GPU)
    assert d["value"] == pytest.approx(cfg["hypotheses"] / (d["ms_per_step"] * 1e-3), rel=1e-9)
    assert d["residual_evals_per_sec"] == pytest.approx(d["value"] * cfg["n_matches"], rel=1e-9)
    assert d["bursts"] >= 50 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    assert r["flop_per_launch"] == 18 * 2048 * 4096
    assert r["achieved"] == pytest.approx(r["flop_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12, rel=1e-9)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-12) and 0.0 < r["frac"] < 1.0
    # counter bytes are quoted only for the library build they were measured with; otherwise null and the reason
    if r["traffic"] is None:
        assert "build" in r["traffic_source"]
    else:
        assert isinstance(r["traffic"], int) and r["traffic"] >= r["algorithmic_bytes_per_launch"] > 0
    assert r["bound_unit"] == "valu_fp64"
    s_ = d["aux_steady_state"]
    assert s_["steps_per_burst"] == 2000 and s_["ms_per_step_min"] <= s_["ms_per_step"] <= s_["ms_per_step_max"]
    if "state" in s_:   # (lines written since the field exists) which of the device's two step rates the record was taken in
        spread = (s_["ms_per_step_max"] - s_["ms_per_step_min"]) / s_["ms_per_step"]
        assert s_["state"] == ("steady" if spread < 0.03 else "noisy")
    if "cpu_baseline" in d:  # (the short runs of a profile set skip it: --no-cpu-baseline)
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["unit"] == "hypotheses/s" and c["cores"] == 1 and c["value"] > 0
        assert isinstance(c["sample"], str) and c["winner_matches_gpu"] is True
    for aux in ("aux_large_h", "aux"):
        if aux in d:
            a = d[aux]["roofline"]
            assert 0.0 < a["frac"] < 1.0 and a["frac"] == pytest.approx(a["achieved"] / a["peak"], rel=1e-12)


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_multi_gpu_run_starts_its_own_ranks():
    """`python bench.py --gpus N` started plainly runs torch.distributed.run with N ranks as a child process and
    returns its exit code; a rank of such a job (RANK set) and a one-GPU run do not relaunch."""
    bench = _bench_module()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert bench.maybe_relaunch(8, argv, {"PATH": "/usr/bin"}, call=fake_call) == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    k = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[k + 1:] == argv                       # the ranks get the same arguments
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "RANK" not in seen["env"]
    called = []
    assert bench.maybe_relaunch(8, argv, {"RANK": "3", "WORLD_SIZE": "8"}, call=lambda *a, **k: called.append(1)) is None
    assert bench.maybe_relaunch(1, ["--gpus", "1"], {}, call=lambda *a, **k: called.append(1)) is None
    assert not called


def test_traffic_is_quoted_for_the_build_it_was_measured_with_only(tmp_path):
    bench = _bench_module()
    f = tmp_path / "traffic.json"
    f.write_text(json.dumps({"build_hash": "0123456789abcdef", "k_ransac_bytes_per_launch": 5, "source": "x"}))
    t, why = bench.load_traffic("0123456789abcdef", str(f))
    assert t["k_ransac_bytes_per_launch"] == 5 and why is None
    t, why = bench.load_traffic("fedcba9876543210", str(f))
    assert t == {} and "0123456789abcdef" in why and "fedcba9876543210" in why
    t, why = bench.load_traffic("x", str(tmp_path / "missing.json"))
    assert t == {} and "not readable" in why
    # the tracked file carries the stamp
    assert "build_hash" in json.load(open(os.path.join(REPO, "profiles", "traffic.json")))


def test_traffic_file_matches_the_last_counter_set():
    t = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
    tag = t["source"].split("profiles/")[1].split("_counters.json")[0]
    c = json.load(open(os.path.join(REPO, "profiles", tag + "_counters.json")))
    k = c["k_ransac (N=4096, H=2048)"]
    assert t["k_ransac_bytes_per_launch"] == int(2 * 1024 * k["FETCH_SIZE_KB"] + 1024 * k["WRITE_SIZE_KB"])
    assert t["k_ransac_algorithmic_bytes"] == 4096 * 32 + 2048 * (256 + 72 + 12)
