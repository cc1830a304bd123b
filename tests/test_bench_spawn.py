"""`python bench.py --gpus N` without a torchrun environment starts its N ranks itself (the driver's command shape;
reference launch: Makefile:21-23 `torchrun --nproc_per_node=N`). CPU test with a stub worker: every rank gets the
torchrun variables, exactly ONE line reaches the parent's stdout, and a failing rank fails the launch."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, body, gpus=2):
    worker = tmp_path / "worker.py"
    worker.write_text(textwrap.dedent(body))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DFINE_BENCH_WORKER"] = str(worker)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1"],
                          env=env, capture_output=True, text=True, timeout=120)


def test_gpus2_spawns_ranks_and_prints_one_line(tmp_path):
    r = _run(tmp_path, """
        import json, os, sys
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1"
        assert int(os.environ["MASTER_PORT"]) > 0 and "--gpus" in sys.argv
        print(json.dumps({"rank": rank, "n_gpus": world, "argv": sys.argv[1:]}))  # rank 0 -> stdout, others -> the parent's stderr
    """)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["rank"] == 0 and out["n_gpus"] == 2 and out["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert '"rank": 1' in r.stderr


def test_failing_rank_fails_the_launch(tmp_path):
    r = _run(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(60)  # stands for a rank stuck in a collective: the launcher must end it
    """)
    assert r.returncode == 3
    assert r.stdout.strip() == ""
