"""CPU: `python bench.py --gpus N` with no torchrun around it must start N ranks itself: the launcher
starts N fresh children with the rendezvous environment, relays rank 0's JSON line and propagates a failing rank's exit code.
--rendezvous-only keeps the children off the GPU: they join a gloo group and count themselves with an all-reduce."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(dict({"IRON_BENCH_BACKEND": "gloo"}, **kw))
    return env


def test_gpus_n_without_torchrun_starts_n_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert r.stdout.strip() == lines[0], "stdout must carry the JSON line and nothing else (gloo's connection notes go to stderr): %r" % r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["ranks_counted"] == 2
    assert d["launcher"] == "self" and d["backend"] == "gloo"


def test_three_ranks_and_the_torchrun_form_agree():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--rendezvous-only"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["ranks_counted"] == 3
    # the driver's form for N > 1: torchrun around the same script
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", BENCH, "--gpus", "2", "--rendezvous-only"], env=_env(), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ranks_counted"] == 2 and d["launcher"] == "torchrun"
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1, "rank 0 prints one line under torchrun too: %r" % r.stdout


def test_a_failing_rank_fails_the_launcher():
    # every rank raises in init_process_group: the launcher must exit non-zero and relay no JSON line
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], env=_env(IRON_BENCH_BACKEND="no_such_backend"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_more_ranks_than_gpus_is_refused_up_front():
    # this container has no GPU: the real (non-rehearsal) N>1 bench must say so instead of printing an N=1 line
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"], env=_env(), capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() >= 8:
        return
    assert r.returncode != 0 and "GPU(s)" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
