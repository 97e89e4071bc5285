"""bench.py's N-rank launch path on a machine without GPUs: `python bench.py --gpus 2 --dry-run` (no launcher, no
WORLD_SIZE in the environment — the way the driver may call it) must start two ranks by itself, shard the global batch
with `partition_molecules`, run one collective (gloo here; RCCL on the GPU box) and print ONE JSON line with n_gpus = 2."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    env["OMP_NUM_THREADS"] = "2"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), p.stderr


def test_self_launch_two_ranks_dry_run():
    line, err = _run(["--gpus", "2", "--dry-run", "--batch", "6", "--atoms", "12"])
    assert "re-executing under torch.distributed.run" in err
    assert line["n_gpus"] == 2 and line["dry_run"] is True
    assert line["every_molecule_owned_once"] is True
    assert sum(line["shard_sizes"]) == 12 and min(line["shard_sizes"]) >= 4    # balanced by triplet count
    assert line["collective"] == "gloo all_reduce"


def test_single_rank_dry_run_needs_no_launcher():
    line, err = _run(["--gpus", "1", "--dry-run", "--batch", "4", "--atoms", "12"])
    assert "re-executing" not in err
    assert line["n_gpus"] == 1 and line["shard_sizes"] == [4]
