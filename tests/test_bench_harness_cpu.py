"""bench.py's own rank fan-out: `--gpus N` without a launcher must start N ranks itself, before any GPU
call, and must refuse to report fewer ranks than requested.  No GPU here: TTV_BENCH_DRYRUN=1 runs the harness (spawn, rendezvous on
127.0.0.1, one gloo collective, rank 0's JSON line) without the measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks_and_reports_them():
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], TTV_BENCH_DRYRUN="1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"]["world_size"] == 2 and line["dry_run"] is True and line["value"] is None


def test_gpus_beyond_the_visible_devices_is_refused():
    # no GPU in this container (and on a one-GPU box --gpus 2 must fail the same way): never a silent n_gpus: 1
    import torch
    if torch.cuda.device_count() >= 2:
        return
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert "GPU(s) visible" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_refused():
    p = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", TTV_BENCH_DRYRUN="1")
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_train_dp_fan_out_hands_every_rank_the_same_run_token():
    """tools/train_dp.py --gpus N starts its own ranks; they meet at a shard directory named after the run token, so the token must be
    one per run, not one per rank."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_dp_tool", os.path.join(ROOT, "tools", "train_dp.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    envs = mod.fan_out_envs(4, 29999)
    assert len({e["TTV_RUN_TOKEN"] for e in envs}) == 1
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and all(e["WORLD_SIZE"] == "4" for e in envs)
    assert all(e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
