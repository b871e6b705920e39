"""`bench.py --gpus N` without a launcher must start N ranks itself: rehearsed here on the CPU over gloo
(`--rehearse` swaps the GPU work for a token step; rendezvous, barriers and the MAX-over-ranks reduction are the real ones)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=env,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                     # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    line = _run("--gpus", "2", "--rehearse", "--steps", "2", "--warmup", "0")
    assert line["n_gpus"] == 2 and line["rank_sum"] == 3.0 and line["steps"] == 2


def test_the_line_carries_what_the_collectives_saw():
    """n_gpus must come from the process group, with the evidence beside it -- a SUM all-reduce of ones
    (ranks_seen), the all-gathered rank / device / PID lists, taken after init and again after the timed region, and every
    rank's own step time beside the MAX.  (The real run adds the same object as `rccl`: same function.)"""
    line = _run("--gpus", "2", "--rehearse", "--steps", "2", "--warmup", "0")
    ev = line["rccl"]
    assert ev["world"] == 2 and ev["ranks_seen"] == 2 and ev["backend"] == "gloo" and ev["ranks"] == [0, 1]
    assert len(ev["device_ids"]) == 2 and len(set(ev["pids"])) == 2            # two processes, not one counted twice
    after = ev["after_timed_region"]
    assert after["ranks_seen"] == 2 and after["pids"] == ev["pids"]
    ms = line["ms_per_step_by_rank"]
    assert len(ms["all"]) == 2 and ms["min"] <= ms["max"] and ms["min"] > 0
    one = _run("--gpus", "1", "--rehearse", "--steps", "1")
    assert one["rccl"]["ranks_seen"] == 1 and one["rccl"]["world"] == 1 and len(one["ms_per_step_by_rank"]["all"]) == 1


def test_gpus_1_stays_single_process():
    line = _run("--gpus", "1", "--rehearse", "--steps", "2")
    assert line["n_gpus"] == 1 and line["rank_sum"] == 1.0


def test_under_a_launcher_world_size_wins():
    """The driver's form: torch.distributed.run sets WORLD_SIZE; bench.py must not spawn again."""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--rehearse", "--steps", "1"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_a_dead_rank_stops_the_launcher_with_its_status():
    """the parent used to wait() for the ranks in order with no timeout -- one dead rank left the others in the
    collective and the parent never returned.  Now it polls, stops the survivors and reports the failure."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["TMDIFF_BENCH_REHEARSE_FAIL_RANK"] = "1"
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 7, (p.returncode, p.stderr[-2000:])
    assert "rank 1 exited with status 7" in p.stderr
    assert time.monotonic() - t0 < 120
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]          # no result line from a failed run


def test_visible_gpus_needs_no_gpu_runtime():
    """The launcher parent counts devices from the environment / the KFD topology: it never imports torch."""
    code = ("import sys, os; sys.argv=['bench.py']; os.environ['HIP_VISIBLE_DEVICES']='0,1,2'; "
            f"sys.path.insert(0, {ROOT!r}); import bench; assert bench.visible_gpus() == 3; "
            "os.environ['ROCR_VISIBLE_DEVICES']='4'; assert bench.visible_gpus() == 1; "
            "assert 'torch' not in sys.modules; print('ok')")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stderr[-2000:]
