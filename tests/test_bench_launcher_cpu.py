"""CPU: `python bench.py --gpus N` must launch its own ranks (the driver's N>1 path also runs under an external
torchrun; both are driven here on gloo with the stubbed step, MOBI_BENCH_STUB=1)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(MOBI_BENCH_STUB="1", MOBI_BENCH_BACKEND="gloo")
    return env


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout                      # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_flag_spawns_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _line(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["scaling"] == "weak"
    # max over ranks: rank 1 sleeps 4 ms per step, rank 0 only 2 ms
    assert out["ms_per_step"] >= 3.9
    # steps_per_s is a per-rank rate (not multiplied by the world size); value is the whole-job aggregate
    assert abs(out["steps_per_s"] - 1e3 / out["ms_per_step"]) / out["steps_per_s"] < 0.02
    assert abs(out["value"] - out["steps_per_s"] * 16 * 2) / out["value"] < 0.02


def test_eight_ranks_self_verifying_line():
    """The line of an N > 1 run says what the collective backend saw: world size, one record per rank (rank, local rank,
    device, its own step time); the headline time is the maximum of the per-rank times."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _line(r.stdout)
    assert out["n_gpus"] == 8 == out["ranks_seen"] and out["backend"] == "gloo"
    assert [x["rank"] for x in out["ranks"]] == list(range(8)) and sorted(x["local_rank"] for x in out["ranks"]) == list(range(8))
    per = out["ms_per_step_by_rank"]
    assert len(per) == 8 and per[7] >= 15.9 and per[0] < per[7]          # rank r sleeps 2 (r + 1) ms per step
    assert out["ms_per_step"] >= max(per) - 0.5
    assert abs(out["value"] - out["steps_per_s"] * 16 * 8) / out["value"] < 0.02


def test_under_external_torchrun_and_mismatch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
            "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")]
    r = subprocess.run(base + ["--gpus", "2", "--steps", "3", "--warmup", "0"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 2
    # a --gpus that disagrees with the launcher's world size is an error, not a silently wrong record
    env = _env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "disagrees" in r.stderr


def test_single_process_default():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 1
