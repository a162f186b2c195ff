"""CPU test of bench.py's launcher contract: `python bench.py --gpus 2` with no external launcher starts two
child ranks itself (torch.distributed.run, gloo in --dry-run: kernels skipped, embeddings fabricated as in
test_distributed_cpu.py), the printed line says "n_gpus": 2, and a WORLD_SIZE / --gpus mismatch exits non-zero."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env=None, timeout=300):
    e = dict(os.environ)
    e.pop("RANK", None); e.pop("WORLD_SIZE", None); e.pop("LOCAL_RANK", None)
    if env:
        e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=e, timeout=timeout, cwd=ROOT)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_gpus2_self_launch_dry_run():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--utterances-per-step", "7"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout               # rank 0 only
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["dry_run"] is True and line["gather_ok"] is True
    assert line["scaling"] == "strong" and line["metric"].startswith("real-time factor")


def test_gpus1_dry_run_single_process():
    r = _run(["--gpus", "1", "--dry-run", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout)[0]["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--dry-run"], env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
