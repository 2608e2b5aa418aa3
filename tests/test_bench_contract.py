"""bench.py prints exactly one JSON line on stdout with the contract's keys (run small and fast)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nx", "512", "--steps", "6", "--warmup", "2",
                        "--jacobi-sweeps-timed", "20", "--cpu-seconds", "0.5", "--sustained-steps", "200", "--profile-steps", "14"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "cell-updates/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert abs(d["value"] - 512 * 512 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]                             # the Jacobi kernel the step runs, from the in-situ profile
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "k_jacobi_tb"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["algorithmic_bytes_per_launch"] == 3 * 8 * 512 * 512 and rf["sweeps_per_launch"] == 5
    assert "traffic" in rf and isinstance(rf["traffic_note"], str) and rf["duration_source"].startswith("in-situ profile") and "14 steps" in rf["duration_source"]
    assert d["config"]["step_schedule"] == "one-chain batch graphs"          # (512^2: chains start at 6 M cells)
    cfg = d["config"]                              # the step's own kernel list: 6 + 2 x 3 + 7 passes, and the three fractions side by side
    assert abs(cfg["arrays_per_cell_update"] - 19.0) < 1e-6 and abs(cfg["bytes_per_cell_update_algorithmic"] - 152.0) < 1e-6
    assert set(cfg["step_kernel_list"]) == {"k_momentum", "k_jacobi_tb", "k_transport"} and cfg["step_kernel_list"]["k_jacobi_tb"]["launches_per_step"] == 2.0
    fr = cfg["step_frac_of_peak"]
    assert abs(fr["own_kernel_list"] - d["step_frac_of_peak_algorithmic"]) < 1e-12 and abs(fr["own_kernel_list"] - fr["reference_schedule_19_passes"]) < 1e-9
    assert abs(fr["own_kernel_list"] - 19 * 8 * 512 * 512 / (d["ms_per_step"] * 1e-3) / 1e9 / 8000.0) < 1e-9
    assert "counter_traffic" in fr and isinstance(fr["counter_traffic_note"], str)
    assert cfg["tm_choice"] == -1 and cfg["sustained_ms_per_step"] > 0 and cfg["sustained_steps"] == 200   # (the rule applies from 6 M cells)
    assert d["courant_violations"] == 0
    assert abs(rf["us_per_launch"] - d["step_kernels"]["k_jacobi_tb"]["us_per_launch_dispatch"]) < 1e-9
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["us_per_launch"] * 1e-6) / 1e9) < 1e-6 * rf["achieved"]
    one = rf["north_star_single_sweep"]            # the single-sweep kernel of the north star's wording: a sub-record
    assert one["kernel"] == "k_jacobi" and abs(one["frac"] - one["achieved"] / 8000.0) < 1e-12 and one["us_per_launch"] > 0
    low = rf["in_step_lowest"]                     # the kernel of the step itself that is furthest below the peak
    assert low["kernel"] in ("k_momentum", "k_jacobi_tb", "k_transport") and abs(low["frac_of_peak"] - low["achieved"] / 8000.0) < 1e-12
    assert low["frac_of_peak"] == min(v["frac_of_peak"] for v in d["step_kernels"].values())
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "oracle/vof_oracle.c" in cb["sample"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert cb["threads_1"]["cores"] == 1 and cb["threads_1"]["value"] > 0 and cb["jacobi_GBs_24B_rule"] > 0
    sus = d["sustained"]
    assert sus["steps"] == 200 and len(sus["ms_per_step_blocks"]) == 2 and sus["value"] > 0
    wc = cfg["worst_case"]                         # the same grid holding the rising bubble (2 % gas): the spread of the step beside the headline
    assert "-ic 2" in wc["workload"] and wc["ms_per_step"] > 0 and wc["steps"] >= 40
    assert abs(wc["slowdown_vs_the_headline_workload"] - wc["ms_per_step"] / d["ms_per_step"]) < 1e-9
    for k in ("k_momentum", "k_jacobi_tb", "k_transport"):
        assert 0 < d["step_kernels"][k]["frac_of_peak"] < 1.5      # 512^2 sits in cache: may exceed the HBM figure


@pytest.mark.gpu
def test_bench_default_workload_reports_the_1024_residual_solve():
    """BASELINE configs[1] as a bench leg (small step counts elsewhere to keep it short)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2",
                        "--jacobi-sweeps-timed", "20", "--no-cpu-baseline", "--no-scaling-reference",
                        "--sustained-steps", "100", "--profile-steps", "12"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    rs = d["residual_solve_1024"]
    assert rs["converged"] and rs["residual"] <= 1e-6 and 100000 <= rs["iterations"] < 3000000 and rs["sweeps_per_s"] > 1e4


def test_bench_defaults_time_a_window_of_at_least_200_steps():
    """One headline, not two: the default timed window is long enough (0.12 s at 4096^2) that it and the 1000-step
    `sustained` record tell the same story; the in-situ profile behind `roofline` reaches into the tiny-value front."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    assert a.steps >= 200 and a.warmup >= 2 and a.profile_steps >= 300 and a.gpus == 1


def test_bench_refuses_a_gpu_count_that_does_not_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)


def _clean_env():
    return {k: v for k, v in os.environ.items()
            if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VOF2D_RDZV_DIR", "VOF2D_RDZV_TAG")}


def test_bench_starts_its_own_workers_when_no_launcher_did():
    """`python bench.py --gpus N` with WORLD_SIZE unset: the (GPU-free) parent spawns N workers with
    the launcher's environment, they meet in a private rendezvous directory, rank 0's line is relayed.
    --dry-run: no GPU work, so this runs here."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True,
                       text=True, timeout=120, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and [x["rank"] for x in d["ranks"]] == [0, 1, 2, 3]
    assert [x["local_rank"] for x in d["ranks"]] == [0, 1, 2, 3] and all(x["world"] == 4 for x in d["ranks"])
    assert len({x["token"] for x in d["ranks"]}) == 1 and len({x["master"] for x in d["ranks"]}) == 1
    assert d["ranks"][0]["master"].startswith("127.0.0.1:")
    # --same-device (the one-GPU rehearsal of the N > 1 path): every rank is handed device 0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--same-device"], capture_output=True,
                       text=True, timeout=120, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert [x["local_rank"] for x in d["ranks"]] == [0, 0] and all(x["same_device"] for x in d["ranks"])


def test_self_started_run_ends_when_a_worker_dies():
    """A worker that exits non-zero takes the launch down with its code instead of leaving the
    others waiting in the rendezvous (VOF2D_BENCH_TEST_DIE_RANK: that worker exits with code 7)."""
    env = dict(_clean_env(), VOF2D_BENCH_TEST_DIE_RANK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode == 7 and "worker 1 exited with code 7" in r.stderr


def test_rendezvous_directory_must_be_private(tmp_path):
    from vof2d.comms import EnvComm
    d = tmp_path / "shared"
    d.mkdir(mode=0o755)
    os.chmod(d, 0o755)
    with pytest.raises(PermissionError):
        EnvComm(0, 1, 0, rdzv_dir=str(d))
    link = tmp_path / "link"
    target = tmp_path / "target"
    target.mkdir(mode=0o700)
    link.symlink_to(target)
    with pytest.raises(PermissionError):
        EnvComm(0, 1, 0, rdzv_dir=str(link))
    ok = EnvComm(0, 1, 0, rdzv_dir=str(tmp_path / "fresh"))
    assert (os.stat(ok.dir).st_mode & 0o777) == 0o700
    # payloads are data: JSON or .npy, never pickle
    import numpy as np
    assert ok.gather_object({"a": [1, 2.5]}) == [{"a": [1, 2.5]}]
    assert np.array_equal(ok.gather_object(np.arange(6.0).reshape(2, 3))[0], np.arange(6.0).reshape(2, 3))
    with open(os.path.join(ok.dir, "bcast1"), "wb") as f:
        import pickle
        f.write(pickle.dumps({"evil": 1}))
    ok2 = EnvComm(1, 2, 1, rdzv_dir=ok.dir)
    with pytest.raises(ValueError):
        ok2.broadcast_object(None)


def _run_bench(args, env=None, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=dict(_clean_env(), **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), r.stderr


@pytest.mark.gpu
def test_bench_strip_path_with_one_rank_native_carrier_and_cost_probe():
    """The N > 1 code path of bench.py driven with one rank (--force-dist): supervisor -> child, EnvComm
    rendezvous, StripSolver, the cost probe + re-cut block (VOF2D_BENCH_TEST_BALANCE), timed region,
    single-GPU reference leg skipped for a non-default grid."""
    d, err = _run_bench(["--force-dist", "--nx", "1024", "--steps", "6", "--warmup", "2", "--jacobi-sweeps-timed", "20"],
                        env={"VOF2D_BENCH_TEST_BALANCE": "1"})
    assert d["n_gpus"] == 1 and d["config"]["rows_per_rank"] == [1024] and d["value"] > 0
    assert d["config"]["exchange"] == "none" and "cost probe failed" not in err


@pytest.mark.gpu
def test_bench_strip_path_with_one_rank_torch_carrier():
    d, err = _run_bench(["--force-dist", "--exchange", "torch", "--nx", "1024", "--steps", "6", "--warmup", "2",
                         "--jacobi-sweeps-timed", "20"])
    assert d["n_gpus"] == 1 and d["config"]["rows_per_rank"] == [1024] and d["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,nx,overlap", [("f64", 1024, 5), ("f32", 768, 4)])
def test_bench_two_ranks_rehearsed_on_one_device(dtype, nx, overlap):
    """`bench.py --gpus 2 --same-device`: the N > 1 path end to end on the one GPU of the test box -- the GPU-free launcher starts
    two fresh workers, each rank's supervisor its child; the children meet in the rendezvous directory (EnvComm), RCCL refuses a
    communicator of two ranks on one device, the halos travel by the second carrier (gloo, rows staged through host memory);
    cost probe and re-cut, warm-up, timed region, the same-grid single-GPU leg, the N > 1 line.  The fields the two strips end
    with (SHA-256 in the line) are the single domain's after the same number of steps."""
    import hashlib
    from vof2d._lib import hip_api
    from vof2d.engine import Engine, make_desc
    steps, warmup = 7, 4
    d, err = _run_bench(["--gpus", "2", "--same-device", "--digest", "--nx", str(nx), "--dtype", dtype, "--steps", str(steps), "--warmup", str(warmup),
                         "--jacobi-sweeps-timed", "20", "--overlap", str(overlap), "--attempt-timeout", "240"], timeout=1500)
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["value"] > 0 and d["scaling"] == "strong"
    c = d["config"]
    assert c["same_device_rehearsal"] is True and c["multi_gpu_hardware_verified"] is False
    assert len(c["rows_per_rank"]) == 2 and sum(c["rows_per_rank"]) == nx and min(c["rows_per_rank"]) >= 18
    assert c["exchange"].startswith("torch/gloo") and c["overlap"] == overlap
    assert "native RCCL exchange unavailable" in err                       # (the first carrier was tried: rendezvous + vof_comm_init)
    ref1 = d["strong_scaling_reference_n1"]
    assert ref1 and ref1["value"] > 0 and d["speedup_same_grid"] > 0       # the same-grid single-GPU leg ran beside it
    api = hip_api()
    e = Engine(api, make_desc(api, nx, nx, dtype, "f32", device=0, dt=c["dt"]))
    e.set_init_F(1)
    e.step(steps + warmup)
    got = d["fields_sha256"]
    assert got["istep"] == steps + warmup and got["shape"] == [nx + 2, nx + 2]
    for f in ("F", "u", "v", "p"):
        assert got[f] == hashlib.sha256(e.get(f).tobytes()).hexdigest(), "field %s of the two strips differs from the single domain's" % f
    e.close()


@pytest.mark.gpu
def test_bench_watchdog_falls_back_to_the_second_carrier():
    """A native attempt that hangs (VOF2D_BENCH_TEST_HANG) is killed by the supervisor after
    --attempt-timeout and the torch.distributed carrier produces the line."""
    d, err = _run_bench(["--force-dist", "--nx", "512", "--steps", "4", "--warmup", "2", "--jacobi-sweeps-timed", "20",
                         "--attempt-timeout", "8"], env={"VOF2D_BENCH_TEST_HANG": "native"})
    assert "native attempt exceeded" in err and d["value"] > 0


def test_traffic_is_quoted_only_for_the_kernel_sources_it_was_measured_on(tmp_path):
    """roofline.traffic comes from committed rocprofv3 --pmc passes, not from the run: bench.py repeats it
    only while the profile's recorded hash of csrc/ equals the sources the library is built from."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from vof2d._lib import kernel_source_hash
    rec = {"nx": 4096, "ny": 4096, "dtype": "f64", "tag": "t", "hbm_bytes_per_launch": {"single": 4.1e8, "tb": 4.9e8},
           "kernel_source_sha256": kernel_source_hash()}
    p = tmp_path / "pmc.json"
    p.write_text(json.dumps(rec))
    got, note = bench.load_pmc_traffic(4096, 4096, "f64", str(p))
    assert got == rec["hbm_bytes_per_launch"] and "same kernel sources" in note
    assert bench.load_pmc_traffic(2048, 4096, "f64", str(p)) == ({}, "the committed PMC profile (t) is of another workload")
    rec["kernel_source_sha256"] = "0" * 64
    p.write_text(json.dumps(rec))
    got, note = bench.load_pmc_traffic(4096, 4096, "f64", str(p))
    assert got == {} and "other kernel sources" in note
    del rec["kernel_source_sha256"]                       # a profile from before the hash existed
    p.write_text(json.dumps(rec))
    assert bench.load_pmc_traffic(4096, 4096, "f64", str(p))[0] == {}
    assert bench.load_pmc_traffic(4096, 4096, "f64", str(tmp_path / "none.json")) == ({}, "no committed PMC profile")


@pytest.mark.gpu
def test_bench_reports_the_form_the_handle_keeps():
    """A large fp64 grid: the handle chooses its batch form by a rule on the state (a dam-break: k_tm + k_jacobi_pair;
    config.step_schedule, config.tm_choice, config.gas_share); `roofline` then names k_tm (8 algorithmic passes), carries
    the Jacobi kernel of the step and the record of the four classic kernels one at a time, and config's per-step byte
    count is the kept form's own kernel list (launches per step x passes), not the four-kernel schedule's 19 passes."""
    d, _ = _run_bench(["--nx", "3072", "--steps", "16", "--warmup", "2", "--no-cpu-baseline", "--no-extras", "--no-scaling-reference",
                    "--profile-steps", "24"])
    cfg, rf = d["config"], d["roofline"]
    assert cfg["handle_warm_steps"] == 19 and cfg["handle_warm_ms"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["bound"] == "hbm"
    assert cfg["step_schedule"].startswith("k_tm") and cfg["tm_choice"] == 1 and 0.8 < cfg["gas_share"] < 0.85
    assert rf["kernel"] == "k_tm" and rf["algorithmic_passes"] == 8 and rf["algorithmic_bytes_per_launch"] == 8 * 8 * 3072 * 3072
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["us_per_launch"] * 1e-6) / 1e9) < 1e-3 * rf["achieved"]   # (us rounded to 0.01)
    run = d["step_kernels_as_run"]
    assert set(run) >= {"k_tm", "k_jacobi_pair"} and abs(run["k_jacobi_pair"]["launches_per_step"] - 1.0) < 1e-9
    assert 0.8 <= run["k_tm"]["launches_per_step"] < 1.0 and rf["jacobi_kernel_of_the_step"]["sweeps_per_launch"] == 10
    assert rf["one_kernel_at_a_time"]["kernel"] == "k_jacobi_tb" and 0 < rf["one_kernel_at_a_time"]["frac"] < 1
    own = sum(v["launches_per_step"] * v["algorithmic_passes"] for v in run.values())
    assert abs(cfg["arrays_per_cell_update"] - own) < 2e-3 and 11.0 < own < 13.0 and cfg["arrays_per_cell_update_four_kernel_schedule"] == 19
    assert set(cfg["step_kernel_list"]) == set(run)
    fr = cfg["step_frac_of_peak"]
    assert abs(fr["own_kernel_list"] * 19.0 / own - fr["reference_schedule_19_passes"]) < 1e-9 and fr["own_kernel_list"] == d["step_frac_of_peak_algorithmic"]
    assert rf["north_star_single_sweep"]["kernel"] == "k_jacobi"
