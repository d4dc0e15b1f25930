"""CPU tests: outlier filter + statistics against the reference's golden output, and the
shard/all-gather layer on a world_size-2 gloo group (the GPU engine is replaced by the CPU oracle
as the per-rank runner: what is under test here is the partitioning and the collective)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from erpl_monte_carlo_sim_amd import _abi, analysis, dist, flatten, models

import helpers as H


def test_analyze_matches_reference_stats():
    g = H.load_json("stats.json")
    inp = g["inputs"]
    n = len(inp["apogee_altitude"])
    params = flatten.generate_parameter_samples(H.UNCERTAINTY, n)
    results = [{"apogee_altitude": inp["apogee_altitude"][i], "range": inp["range"][i],
                "flight_time": inp["flight_time"][i], "simulation_id": i, "parameters": params[i]} for i in range(n)]
    results[inp["none_index"]] = None
    out = analysis.analyze(results)
    assert out["n_samples"] == g["n_samples"] and out["n_failed"] == g["n_failed"] and out["n_outliers"] == g["n_outliers"]
    for key in ("apogee_altitude", "range", "flight_time"):
        for stat in ("mean", "std", "min", "max"):
            assert out[key][stat] == pytest.approx(g[key][stat], rel=1e-14), (key, stat)
        assert np.allclose(out[key]["percentiles"], g[key]["percentiles"], rtol=1e-14)
    assert [r["simulation_id"] for r in out["results"]] == g["valid_ids"]
    assert [r["simulation_id"] for r in out["outliers"]] == g["outlier_ids"]
    assert [r["outlier_reasons"] for r in out["outliers"]] == g["outlier_reasons"]
    pr, gr = out["parameter_ranges_observed"], g["parameter_ranges_observed"]
    assert set(pr) == set(gr)
    for k in gr:
        assert np.allclose(pr[k]["min"], gr[k]["min"], rtol=0, atol=0) and np.allclose(pr[k]["max"], gr[k]["max"], rtol=0, atol=0)


def test_analyze_error_behaviour():
    with pytest.raises(ValueError, match="No valid simulation results"):
        analysis.analyze([None, None])
    bad = [{"apogee_altitude": 9e4, "range": 1.0, "flight_time": 10.0, "parameters": {}}]
    with pytest.raises(ValueError, match="No physically reasonable"):
        analysis.analyze(bad)


def test_outlier_mask_equals_reason_list():
    rng = np.random.RandomState(3)
    a = np.concatenate([rng.normal(25000, 30000, 500), [np.nan, np.inf, 50.0, 100.0, 80000.0, 88073.4]])
    r = np.abs(np.concatenate([rng.normal(1e5, 1e5, 500), [1.0, 2.0, np.nan, 200000.0, 200000.1, 5.0]]))
    f = np.concatenate([rng.normal(400, 200, 500), [1.0, np.nan, 600.0, 600.1, 5.0, 5.0]])
    m = analysis.outlier_mask(a, r, f)
    for i in range(len(a)):
        assert bool(m[i]) == bool(analysis.outlier_reasons(a[i], r[i], f[i])), i


def test_device_statistics_equal_host_statistics():
    """torch implementation (runs on the GPU in production) == the NumPy/reference-pinned one."""
    rng = np.random.RandomState(11)
    n = 5000
    summ = np.zeros((16, n))
    summ[_abi.SUM_APOGEE_ALT] = rng.normal(25000, 20000, n)
    summ[_abi.SUM_RANGE] = np.abs(rng.normal(50000, 90000, n))
    summ[_abi.SUM_FLIGHT_TIME] = rng.normal(300, 150, n)
    summ[_abi.SUM_APOGEE_ALT, :7] = [np.nan, np.inf, 50.0, 100.0, 80000.0, 88073.4, -np.inf]
    summ[_abi.SUM_RANGE, 7:10] = [np.nan, 200000.0, 200000.1]
    status = rng.randint(0, 4, n).astype(np.int32)
    res = [{"apogee_altitude": summ[0, i], "range": summ[4, i], "flight_time": summ[5, i], "parameters": {}}
           for i in range(n)]
    ref = analysis.analyze(res)
    got = analysis.device_statistics(torch.from_numpy(summ), torch.from_numpy(status))
    assert got["n_samples"] == ref["n_samples"] and got["n_outliers"] == ref["n_outliers"]
    for key in ("apogee_altitude", "range", "flight_time"):
        for stat in ("mean", "std", "min", "max"):
            assert got[key][stat] == pytest.approx(ref[key][stat], rel=1e-12), (key, stat)
        assert np.allclose(got[key]["percentiles"], ref[key]["percentiles"], rtol=1e-12)
    assert sum(got["termination_counts"].values()) == n
    with pytest.raises(ValueError):
        analysis.device_statistics(torch.full((16, 4), float("nan"), dtype=torch.float64))


def test_report_writer_matches_reference_format(tmp_path):
    """monte_carlo_report.txt / .json in the reference's on-disk format: the statistics block of one of
    the reference's own committed reports must render to the very same text lines."""
    import json
    from erpl_monte_carlo_sim_amd import reports
    g = H.load_json("report_format.json")

    class A:  # the four objects + uncertainty table a MonteCarloAnalyzer carries
        rocket, motor = models.Rocket(), models.LiquidMotor()
        atmosphere, wind_model = models.StandardAtmosphere(), models.WindModel()
        uncertainty_params = H.UNCERTAINTY

    analysis = dict(g["analysis"])
    analysis["results"] = [{"simulation_id": 7, "apogee_altitude": np.float64(1.5), "trajectory": {"time": np.arange(3.0)}}]
    rep = reports.save_report(A, analysis, str(tmp_path))
    assert rep["simulation_summary"]["success_rate"] == g["success_rate"]
    txt = open(tmp_path / "monte_carlo_report.txt").read().split("\n")
    assert [l for l in txt if not l.startswith("Generated:")] == g["txt_lines"]
    on_disk = json.load(open(tmp_path / "monte_carlo_report.json"))
    assert list(on_disk.keys()) == g["json_keys"]
    sim = json.load(open(tmp_path / "simulation_results" / "sim_7.json"))
    assert sim["trajectory"]["time"] == [0.0, 1.0, 2.0]
    # with a performance block (run_optimized_monte_carlo, monte_carlo.py:555-560)
    analysis["performance"] = {"total_time": 1.234, "simulations_per_second": 1e6, "cores_used": 16}
    lines = reports.report_text(reports.build_report(A, analysis))
    assert "  Simulations per second: 1000000.0" in lines and "  Cores used: 16" in lines


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 1000, 131072):
        for ws in (1, 2, 3, 8):
            got = []
            for r in range(ws):
                lo, hi, per = dist.shard_bounds(n, r, ws)
                assert 0 <= hi - lo <= per
                got += list(range(lo, hi))
            assert got == list(range(n))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    pl = flatten.generate_parameter_samples(H.UNCERTAINTY, n)
    hb = flatten.dispersed_batch(models.Rocket(), models.LiquidMotor(), models.WindModel(), H.EXAMPLE_IC, pl,
                                 H.CSV_ALT, H.CSV_WIND)
    cfg = H.make_config("liquid")
    calls = []

    def runner(shard):
        calls.append(shard.n)
        s, t = orc.run_batch(cfg, shard, threads=1)
        return torch.from_numpy(s), torch.from_numpy(t)

    summ, status = dist.run_sharded(hb, runner)
    q.put((rank, calls, summ, status))
    td.barrier()
    td.destroy_process_group()


def _worker_local(rank, world, port, n, q):
    """Each rank builds ONLY its own shard on the host (per-sample seeds) and hands it to run_local_shard."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    lo, hi, _ = dist.shard_bounds(n, rank, world)
    arr = flatten.generate_parameter_arrays(H.UNCERTAINTY, n)
    mine = {k: v[lo:hi] for k, v in arr.items()}
    hb = flatten.dispersed_batch(models.Rocket(), models.LiquidMotor(), models.WindModel(), H.EXAMPLE_IC, mine,
                                 H.CSV_ALT, H.CSV_WIND) if hi > lo else None
    cfg = H.make_config("liquid")

    def runner(shard):
        s, t = orc.run_batch(cfg, shard, threads=1)
        return torch.from_numpy(s), torch.from_numpy(t)

    summ, status = dist.run_local_shard(n, hb, runner)
    q.put((rank, 0 if hb is None else hb.n, summ, status))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("n", [7, 1])
def test_two_rank_gloo_local_shards(n):
    """Host preparation proportional to n / world: every rank builds only samples [lo, hi) (an EMPTY shard on
    rank 1 when n = 1) and the gathered result equals the single-process batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_local, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=180) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from oracle import oracle as orc
    pl = flatten.generate_parameter_samples(H.UNCERTAINTY, n)
    hb = flatten.dispersed_batch(models.Rocket(), models.LiquidMotor(), models.WindModel(), H.EXAMPLE_IC, pl,
                                 H.CSV_ALT, H.CSV_WIND)
    ref_s, ref_t = orc.run_batch(H.make_config("liquid"), hb, threads=2)
    per = -(-n // 2)
    assert [o[1] for o in outs] == [per, n - per]
    for rank, _, summ, status in outs:
        assert np.array_equal(summ, ref_s, equal_nan=True), rank
        assert np.array_equal(status, ref_t), rank


@pytest.mark.parametrize("n", [9, 2])
def test_two_rank_gloo_shard_and_gather(n):
    """N > 1 path: each rank integrates only its shard; after the all-gather every rank holds the
    same [16, n] summaries, equal to the single-process result."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=180) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from oracle import oracle as orc
    pl = flatten.generate_parameter_samples(H.UNCERTAINTY, n)
    hb = flatten.dispersed_batch(models.Rocket(), models.LiquidMotor(), models.WindModel(), H.EXAMPLE_IC, pl,
                                 H.CSV_ALT, H.CSV_WIND)
    ref_s, ref_t = orc.run_batch(H.make_config("liquid"), hb, threads=2)
    per = -(-n // 2)
    assert outs[0][1] == [per] and outs[1][1] == [n - per]
    for rank, calls, summ, status in outs:
        assert np.array_equal(summ, ref_s, equal_nan=True), rank
        assert np.array_equal(status, ref_t), rank


def _table_of(n, seed=5):
    """A SampleTable with healthy, outlier and non-finite rows, real parameter arrays and one trajectory."""
    from erpl_monte_carlo_sim_amd import results
    rng = np.random.RandomState(seed)
    summ = rng.normal(0, 1, (16, n))
    summ[_abi.SUM_APOGEE_ALT] = rng.normal(25000, 20000, n)
    summ[_abi.SUM_RANGE] = np.abs(rng.normal(50000, 90000, n))
    summ[_abi.SUM_FLIGHT_TIME] = rng.normal(300, 150, n)
    summ[_abi.SUM_APOGEE_ALT, :7] = [np.nan, np.inf, 50.0, 100.0, 80000.0, 88073.4, -np.inf]
    summ[_abi.SUM_RANGE, 7:10] = [np.nan, 200000.0, 200000.1]
    summ[_abi.SUM_STEPS] = rng.randint(100, 60000, n)
    status = (rng.randint(0, 5, n) | (rng.randint(0, 2, n) << 9)).astype(np.int32)
    params = flatten.generate_parameter_arrays(H.UNCERTAINTY, n)
    traj = {11: {"time": np.arange(3.0), "altitude": np.arange(3.0), "position": np.zeros((3, 3))}}
    return results.SampleTable(summ, status, params, traj)


def test_lazy_results_equal_the_eager_list():
    """`analysis['results']` / `['outliers']` are lazy sequences; they must yield the very dicts the
    eager construction gives (same keys in the same order, same values and types), and the column-wise analysis
    must equal analysis.analyze (itself pinned to the reference's numbers above) on that eager list."""
    from erpl_monte_carlo_sim_amd import results
    n = 1000
    table = _table_of(n)
    eager = [table.record(i) for i in range(n)]
    ref = analysis.analyze([dict(r) for r in eager])
    got = results.analyze_table(table)
    for k in ("n_samples", "n_failed", "n_outliers", "apogee_altitude", "range", "flight_time", "parameter_ranges_observed"):
        assert got[k] == ref[k], k
    assert len(got["results"]) == ref["n_samples"] and len(got["outliers"]) == ref["n_outliers"]
    assert got["results"] == ref["results"] and got["outliers"] == ref["outliers"]          # record by record
    assert [r["simulation_id"] for r in got["outliers"]] == [r["simulation_id"] for r in ref["outliers"]]
    assert got["outliers"][0]["outlier_reasons"] == ref["outliers"][0]["outlier_reasons"]
    # list behaviour the reference's consumers rely on: indexing, negative indices, slices, iteration, `+`
    lz = got["results"]
    assert list(lz[3].keys()) == list(ref["results"][3].keys())
    assert lz[-1]["simulation_id"] == ref["results"][-1]["simulation_id"]
    assert [r["simulation_id"] for r in lz[5:9]] == [r["simulation_id"] for r in ref["results"][5:9]]
    both = got["results"] + got["outliers"]
    assert isinstance(both, list) and len(both) == n
    assert type(lz[0]["apogee_altitude"]) is float and type(lz[0]["n_steps"]) is int
    assert isinstance(lz[0]["parameters"]["initial_velocity_offset"], np.ndarray)
    withtraj = [r for r in both if "trajectory" in r]
    assert len(withtraj) == 1 and withtraj[0]["simulation_id"] == 11
    assert np.array_equal(lz.column("apogee_altitude"), [r["apogee_altitude"] for r in ref["results"]])
    with pytest.raises(ValueError, match="No physically reasonable"):
        bad = _table_of(16)
        bad.summary[_abi.SUM_APOGEE_ALT] = 9e4
        results.analyze_table(bad)


def test_lazy_analysis_equals_the_reference_statistics():
    """analyze_table on the inputs of tests/golden/stats.json (the reference's own _analyze_results output; the
    failed sample of that fixture has no counterpart on the GPU path and is left out on both sides)."""
    from erpl_monte_carlo_sim_amd import results
    g = H.load_json("stats.json")
    inp = g["inputs"]
    keep = [i for i in range(len(inp["apogee_altitude"])) if i != inp["none_index"]]
    summ = np.zeros((16, len(keep)))
    summ[_abi.SUM_APOGEE_ALT] = np.array(inp["apogee_altitude"], dtype=np.float64)[keep]
    summ[_abi.SUM_RANGE] = np.array(inp["range"], dtype=np.float64)[keep]
    summ[_abi.SUM_FLIGHT_TIME] = np.array(inp["flight_time"], dtype=np.float64)[keep]
    P = flatten.generate_parameter_arrays(H.UNCERTAINTY, len(inp["apogee_altitude"]))
    P = {k: v[keep] for k, v in P.items()}
    out = results.analyze_table(results.SampleTable(summ, np.zeros(len(keep), dtype=np.int32), P))
    assert out["n_samples"] == g["n_samples"] and out["n_outliers"] == g["n_outliers"]
    for key in ("apogee_altitude", "range", "flight_time"):
        for stat in ("mean", "std", "min", "max"):
            assert out[key][stat] == pytest.approx(g[key][stat], rel=1e-14), (key, stat)
        assert np.allclose(out[key]["percentiles"], g[key]["percentiles"], rtol=1e-14)
    assert [r["outlier_reasons"] for r in out["outliers"]] == g["outlier_reasons"]
    pr, gr = out["parameter_ranges_observed"], g["parameter_ranges_observed"]
    for k in gr:
        assert np.allclose(pr[k]["min"], gr[k]["min"], rtol=0, atol=0) and np.allclose(pr[k]["max"], gr[k]["max"], rtol=0, atol=0)


def _bench_cmd(*extra):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return [sys.executable, os.path.join(root, "bench.py"), *extra], env, root


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 8` as the driver types it: the launcher counts the devices BEFORE starting ranks and
    says what is wrong (here: no GPU at all; on a one-GPU box: "8 GPUs requested, 1 visible"), rc != 0."""
    import subprocess
    cmd, env, root = _bench_cmd("--gpus", "8", "--steps", "1", "--warmup", "0")
    env.pop("ERPL_BENCH_BACKEND", None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=root, timeout=300)
    assert r.returncode != 0 and "8 GPUs requested" in r.stderr and "visible" in r.stderr, (r.stdout[-500:], r.stderr[-1500:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="needs a box WITHOUT a GPU: the ranks must fail")
def test_bench_launcher_returns_nonzero_when_a_rank_fails():
    """The self-launch path (no torch.distributed.run around it): two child ranks are started; without a GPU both
    fail, the launcher ends whatever is left and reports a non-zero exit code, and prints no JSON line."""
    import subprocess
    cmd, env, root = _bench_cmd("--gpus", "2", "--steps", "1", "--warmup", "0", "--samples-per-gpu", "64")
    env["ERPL_BENCH_BACKEND"] = "gloo"     # (skips the one-GPU-per-rank check so that ranks are actually started)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode != 0, (r.stdout[-500:], r.stderr[-1500:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def _worker_failing(rank, world, port, q):
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)

    def runner(shard):
        if rank == 1:
            raise RuntimeError("rank 1 lost its GPU")
        return torch.zeros((16, shard.n), dtype=torch.float64), torch.ones((shard.n,), dtype=torch.int32)

    class Shard:
        n = 3
    try:
        summ, status = dist.run_local_shard(6, Shard(), runner)
        # the healthy rank gets the gathered block back - with the failed rank's samples marked incomplete
        from erpl_monte_carlo_sim_amd.engine import TrajectoryEngine
        try:
            TrajectoryEngine.raise_if_incomplete(status)
            q.put((rank, "no error", None))
        except _abi.IncompleteBatch as e:
            q.put((rank, "incomplete", int(np.sum((status & _abi.ST_INCOMPLETE) != 0))))
    except RuntimeError as e:
        q.put((rank, "raised", str(e)))
    td.barrier()
    td.destroy_process_group()


def test_a_failing_rank_still_takes_part_in_the_gather():
    """an exception inside one rank's integration must not leave the other ranks waiting in the all-gather.
    The failing rank contributes a shard marked ERPL_ST_INCOMPLETE, re-raises its own error afterwards, and the healthy
    rank refuses the gathered result (raise_if_incomplete) instead of hanging or handing on half a batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_failing, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict((o[0], o[1:]) for o in [q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert outs[1] == ("raised", "rank 1 lost its GPU")
    assert outs[0] == ("incomplete", 3)


def test_lazy_results_hand_out_the_same_record_again():
    """in-place annotations of a record survive (the reference returns plain lists of dicts), and tolist()
    is that plain list."""
    from erpl_monte_carlo_sim_amd import results
    got = results.analyze_table(_table_of(200))
    lz = got["results"]
    for r in lz:
        r["my_note"] = r["simulation_id"] * 2
    assert all(r["my_note"] == r["simulation_id"] * 2 for r in lz)
    assert lz[5] is lz[5] and lz[2:4][1] is lz[3]
    lst = lz.tolist()
    assert isinstance(lst, list) and len(lst) == len(lz) and lst[7] is lz[7]
    lst.sort(key=lambda r: -r["apogee_altitude"])      # list operations work on the list
    lst.append({"simulation_id": -1})
    assert len(lz) == len(lst) - 1
