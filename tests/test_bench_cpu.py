"""bench.py's N-rank control flow on the CPU (gloo, world_size 2): both workloads run with the oracle injected as the
index (the HIP index needs a GPU), rank 0 prints one JSON-able line whose collective really spans 2 ranks, and the
row-sharded leg's own exactness check (per-rank exact hits -> all_gather -> plain-torch merge) passes."""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception:   # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put((rank, "ERROR", traceback.format_exc()))
        raise


def _worker_body(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "ICD_BENCH_BACKEND": "gloo", "ICD_BENCH_DEVICE": "cpu"})
    import bench
    import bench_cpu_engine as eng
    ctx = bench.Ctx()
    assert ctx.world == world and ctx.dist.get_world_size() == world
    args = argparse.Namespace(steps=2, warmup=1, settle_ms=0.0, repeats=1, profile_every=1, nq=40, n=900, k=5, mode="auto", rows_per_gpu=700, rowshard_queries=50,
                              rowshard_slice=32, rowshard_steps=2, no_cpu_baseline=True, no_extras=True, no_family=True,
                              config3_queries=70, config3_steps=1)
    rs = bench.run_rowshard(ctx, args, index_factory=eng.index_factory, sharded_factory=eng.sharded_factory)
    rp = bench.run_replicated(ctx, args, index_factory=eng.index_factory)
    c3 = bench.run_config3(ctx, args, index_factory=eng.index_factory)
    if rp is not None:
        rp["config3"] = c3
    q.put((rank, json.dumps(rs) if rs else None, json.dumps(rp) if rp else None))
    ctx.dist.barrier()
    ctx.dist.destroy_process_group()


def test_bench_two_ranks_on_cpu():
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = 29950 + (os.getpid() % 40)
    procs = [mpctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict((r, (a, b)) for r, a, b in (q.get(timeout=240) for _ in range(2)))
    for r, (a, b) in results.items():
        assert a != "ERROR", b
    for p in procs:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert results[1] == (None, None)                      # only rank 0 reports
    rs, rp = json.loads(results[0][0]), json.loads(results[0][1])
    assert rs["n_gpus"] == 2 and rs["config"]["collective_ranks"] == 2 and rs["config"]["corpus_rows_total"] == 1400
    assert rs["ids_exact_on_sample"] and rs["raw_scores_exact_on_sample"] and rs["adjusted_sorted"]
    assert rs["scaling"] == "weak" and rs["metric"] == "queries_per_sec" and rs["value"] > 0 and "roofline" in rs
    assert "configs[4]" in rs["config"]["workload"]
    assert rp["n_gpus"] == 2 and rp["ids_exact"] and rp["recall_at_10"] == 1.0 and rp["adjusted_scores_exact"]
    assert rp["parity_checked_queries"] == 20   # (every 2nd query at N = 2)
    # the N > 1 `value` is the HEADLINE workload replicated and says so; configs[3] is its own object at its own size
    assert "configs[1] replicated on every GPU" in rp["config"]["workload"] and rp["config"]["nq_per_gpu"] == 40
    c3 = rp["config3"]
    assert c3["config"]["workload"].startswith("BASELINE configs[3]") and c3["config"]["nq_per_gpu"] == 70 and c3["config"]["queries_total"] == 140
    assert c3["config"]["slice"] == 32 and c3["ids_exact"] and c3["adjusted_scores_exact"] and c3["parity_checked_queries"] == 70
    assert abs(c3["value"] - 2 * 70 * c3["steps"] / (c3["ms_per_step"] * c3["steps"] / 1e3)) / c3["value"] < 1e-6
    # the row-shard sample covers EVERY slice of the pass, the short last one included (50 queries in slices of 32: 32 + 18)
    assert rs["sample_slices"] == 2 and rs["sample_queries"] == 16
    assert abs(rp["value"] - 2 * 40 * 2 / (rp["ms_per_step"] * 2 / 1e3)) / rp["value"] < 1e-6
    for line in (rs, rp):
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in line, key


# ---- `python bench.py --gpus N` as the driver runs it: the parent starts the ranks itself -------------------------------
SMALL = ["--steps", "2", "--warmup", "1", "--settle-ms", "0", "--repeats", "1", "--nq", "40", "--n", "900", "--k", "5",
         "--rows-per-gpu", "700", "--rowshard-queries", "50", "--rowshard-slice", "32", "--rowshard-steps", "2", "--no-cpu-baseline",
         "--config3-queries", "70", "--config3-steps", "1", "--family-families", "6", "--family-rows", "20"]


def _run_bench(extra_args, extra_env=None, timeout=420):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"ICD_BENCH_BACKEND": "gloo", "ICD_BENCH_DEVICE": "cpu",
                "ICD_BENCH_TEST_ENGINE": os.path.join(ROOT, "tests", "bench_cpu_engine.py")})
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args + SMALL, env=env, capture_output=True,
                          text=True, timeout=timeout, cwd=ROOT)


def test_bench_gpus_2_starts_two_ranks_itself():
    p = _run_bench(["--gpus", "2"])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout     # ONE JSON line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["collective_ranks"] == 2 and "configs[1] replicated on every GPU" in line["config"]["workload"]
    assert line["ids_exact"] and line["adjusted_scores_exact"]
    c3 = line["config3"]
    assert c3["n_gpus"] == 2 and c3["config"]["workload"].startswith("BASELINE configs[3]") and c3["config"]["nq_per_gpu"] == 70
    assert c3["ids_exact"] and c3["adjusted_scores_exact"] and c3["value"] > 0
    rs = line["rowshard"]
    assert rs["n_gpus"] == 2 and rs["config"]["collective_ranks"] == 2 and rs["config"]["corpus_rows_total"] == 1400
    assert rs["ids_exact_on_sample"] and rs["raw_scores_exact_on_sample"] and rs["adjusted_scores_exact_on_sample"]
    assert "oracle" in rs["sample_checked_against"] and rs["sample_slices"] == 2 and rs["sample_queries"] == 16


def test_bench_under_torch_distributed_run_exactly_as_the_driver_launches_n_gt_1():
    """The driver's N > 1 command: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` - the ranks come from the launcher (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in the environment), bench.py must not start any itself, and rank 0 prints the ONE line."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"ICD_BENCH_BACKEND": "gloo", "ICD_BENCH_DEVICE": "cpu",
                "ICD_BENCH_TEST_ENGINE": os.path.join(ROOT, "tests", "bench_cpu_engine.py")})
    port = 29890 + (os.getpid() % 40)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ["--corpus-rows" if a == "--n" else a for a in SMALL]   # (the launcher's parser rejects `--n` as an ambiguous prefix of its own options)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["config"]["collective_ranks"] == 2
    assert line["ids_exact"] and line["adjusted_scores_exact"] and line["value"] > 0
    assert line["config3"]["n_gpus"] == 2 and line["rowshard"]["config"]["collective_ranks"] == 2
    assert line["rowshard"]["ids_exact_on_sample"] and line["rowshard"]["config"]["engine"]


def test_bench_gpus_1_is_a_single_process_line_with_the_extras():
    p = _run_bench(["--gpus", "1"])
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert line["n_gpus"] == 1 and line["config"]["collective_ranks"] == 1 and "rowshard" not in line
    assert "configs[1]" in line["config"]["workload"]
    ex = line["extra"]
    assert ex["windows"]["min"] <= ex["windows"]["median"] <= ex["windows"]["max"]
    for key, rows in (("real_size", 40474), ("clustered", 900), ("exact_mode", 900), ("k20", 900), ("family", 120), ("k100", 900), ("dim1024", 900)):
        assert ex[key]["corpus_rows"] == rows and ex[key]["ids_exact"] and ex[key]["ms_per_step"] > 0 and "fallback_queries" in ex[key]
        assert "last_second_pass" in ex[key]
    assert ex["k100"]["top_k"] == 100 and "single_query" not in ex   # (the single-query latencies are the library's: GPU runs only)
    assert ex["k20"]["top_k"] == 20 and ex["family"]["first_batch"]["ms"] > 0 and "config3" not in line


def test_bench_child_failure_fails_the_parent():
    p = _run_bench(["--gpus", "2", "--no-rowshard"], {"ICD_BENCH_TEST_FAIL_RANK": "1"})
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]   # no result line from a failed run


def test_bench_hung_rank_fails_the_parent_within_the_limit():
    """a rank that never returns (stuck in a collective the others left, a hung device): the parent kills the child's process
    group at --rank-timeout and exits non-zero instead of waiting for the caller's own limit"""
    import time
    t0 = time.time()
    p = _run_bench(["--gpus", "2", "--no-rowshard", "--no-config3", "--rank-timeout", "25"], {"ICD_BENCH_TEST_HANG_RANK": "1"}, timeout=200)
    assert p.returncode == 124 and "rank-timeout" in p.stderr and time.time() - t0 < 150
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]


def test_bench_gpus_8_dry_run_on_cpu():
    """the driver's one-shot `python bench.py --gpus 8`, rehearsed on the CPU engine at reduced sizes: eight gloo ranks, ONE line
    with value / config3 / rowshard whose collectives span 8 ranks, parent and ranks gone inside the limit"""
    import time
    t0 = time.time()
    p = _run_bench(["--gpus", "8", "--rank-timeout", "400"], timeout=500)
    took = time.time() - t0
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["value"] > 0 and line["config"]["collective_ranks"] == 8 and line["ids_exact"]
    assert line["parity_checked_queries"] == 5                   # every 8th of rank 0's 40 queries
    c3, rs = line["config3"], line["rowshard"]
    assert c3["n_gpus"] == 8 and c3["config"]["queries_total"] == 8 * 70 and c3["ids_exact"] and c3["value"] > 0
    assert rs["n_gpus"] == 8 and rs["config"]["collective_ranks"] == 8 and rs["config"]["corpus_rows_total"] == 8 * 700
    assert rs["ids_exact_on_sample"] and rs["raw_scores_exact_on_sample"] and rs["adjusted_scores_exact_on_sample"] and rs["value"] > 0
    assert rs["config"]["engine"] == "torch.distributed" and rs["sample_slices"] == 2 and rs["sample_queries"] == 8   # 4 per slice above two ranks
    assert took < 400, took


def test_bench_rank_with_a_hung_native_trial_does_not_strand_the_others():
    """ONE rank's C-ABI trial timed out (a thread stuck in a GPU collective) - the ranks agree on that over the gloo
    side channel and ALL leave without the teardown barrier; the line is out, the parent exits 0 at once, nobody waits for
    --rank-timeout"""
    import time
    t0 = time.time()
    p = _run_bench(["--gpus", "2", "--no-config3", "--rank-timeout", "120"], {"ICD_BENCH_TEST_TRIAL_HUNG_RANK": "1"}, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert time.time() - t0 < 100
    line = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert line["n_gpus"] == 2 and line["rowshard"]["ids_exact_on_sample"]
    assert "exits without the teardown barrier" in p.stderr


def test_bench_watchdog_relays_a_line_that_was_already_printed():
    """a rank that hangs AFTER rank 0 printed its line (the teardown): the watchdog kills the tree, the parent exits 124 AND relays
    the measurement it had already read"""
    p = _run_bench(["--gpus", "2", "--no-rowshard", "--no-config3", "--rank-timeout", "40"], {"ICD_BENCH_TEST_HANG_AT_EXIT_RANK": "1"}, timeout=200)
    assert p.returncode == 124 and "rank-timeout" in p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` on a box with fewer than N GPUs (none here) must not print an N = 1 line: it exits non-zero
    before starting anything"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ICD_BENCH_DEVICE", "ICD_BENCH_ONE_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env, capture_output=True, text=True,
                       timeout=120, cwd=ROOT)
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr and not p.stdout.strip()


def test_visible_gpus_counts_without_touching_the_runtime(monkeypatch, tmp_path):
    """spawn_ranks' device count comes from the visibility variables or the KFD topology (nodes with SIMDs), never from the
    HIP runtime: the parent of the ranks must not open the device"""
    sys.path.insert(0, ROOT)
    import bench
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,3,5")
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.visible_gpus() == 1
    # both kinds set: HIP indices are relative to the ROCR subset, the smaller list bounds the count
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpus() == 1
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    # no variable: the topology directory (absent in this container -> None, and spawn_ranks asks torch instead)
    got = bench.visible_gpus()
    assert got is None or isinstance(got, int)
    # a fake topology: two GPU nodes (simd_count > 0) and one CPU node
    base = tmp_path / "nodes"
    for i, simd in enumerate((0, 1024, 1024)):
        d = base / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 64\nsimd_count {simd}\nmem_banks_count 1\n")
    import builtins, os as _os
    real_listdir, real_open = _os.listdir, builtins.open
    monkeypatch.setattr(_os, "listdir", lambda p: real_listdir(str(base)) if p == "/sys/class/kfd/kfd/topology/nodes" else real_listdir(p))
    monkeypatch.setattr(builtins, "open", lambda p, *a, **k: real_open(str(p).replace("/sys/class/kfd/kfd/topology/nodes", str(base)), *a, **k))
    assert bench.visible_gpus() == 2


def test_bench_adopts_the_c_abi_group_only_when_every_ranks_trial_passed():
    """round 6: the row-sharded leg reports the C-ABI group's number (config.engine says so, the torch engine's stays next to it)
    when the trial passed on EVERY rank - agreed over the CPU side channel; one rank whose output differed keeps all of them on
    the torch.distributed number. The trial itself is the test engine's stand-in (tests/bench_cpu_engine.py native_trial)."""
    p = _run_bench(["--gpus", "2", "--no-config3", "--rank-timeout", "200"], {"ICD_BENCH_TEST_NATIVE_TRIAL": "ok"}, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    rs = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])["rowshard"]
    assert rs["config"]["engine"] == "icd_group (C ABI)" and "icd_group_search" in rs["config"]["collective"]
    assert rs["native_group_trial"]["status"] == "ok" and rs["native_group_trial"]["adopted_as_the_legs_engine"] is True
    assert rs["value"] == rs["native_group_trial"]["value"] and rs["ms_per_step"] == 2.0
    assert rs["torch_distributed_engine"]["value"] > 0 and rs["ids_exact_on_sample"]
    p = _run_bench(["--gpus", "2", "--no-config3", "--rank-timeout", "200"], {"ICD_BENCH_TEST_NATIVE_TRIAL": "rank1_differs"}, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    rs = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])["rowshard"]
    assert rs["config"]["engine"] == "torch.distributed" and rs["torch_distributed_engine"] is None
    assert rs["native_group_trial"]["status"] == "ok" and rs["native_group_trial"]["adopted_as_the_legs_engine"] is False   # (rank 0's own trial passed; rank 1's did not)
    assert rs["value"] != rs["native_group_trial"]["value"] and rs["ids_exact_on_sample"]
