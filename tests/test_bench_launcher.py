"""bench.py --gpus N without a launcher starts its own rank processes (bench.spawn_ranks):
the argument and environment plumbing, on CPU.  The end-to-end N = 2 line itself is a GPU
rehearsal (tests/test_gpu_sharded.py::test_bench_self_launch_two_ranks_share_the_gpu)."""
import json
import subprocess
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

import bench  # noqa: E402  (imports neither torch nor the library at module level)

CHILD = ("import os, json, sys; r = os.environ['RANK'];"
         "print(json.dumps({k: os.environ.get(k) for k in "
         "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')} | {'argv': sys.argv[1:]}))")


def test_bench_module_import_touches_neither_torch_nor_the_gpu():
    code = "import sys; sys.path.insert(0, %r); import bench; assert 'torch' not in sys.modules, 'torch imported'" % str(REPO)
    subprocess.run([sys.executable, "-c", code], check=True)


def test_spawn_ranks_environment_and_rank0_stdout():
    rc, out = bench.spawn_ranks([sys.executable, "-c", CHILD, "--gpus", "3", "--steps", "2"], 3)
    assert rc == 0
    lines = [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "only rank 0's stdout is relayed"
    e = lines[0]
    assert e["RANK"] == "0" and e["LOCAL_RANK"] == "0" and e["WORLD_SIZE"] == "3" and e["LOCAL_WORLD_SIZE"] == "3"
    assert e["MASTER_ADDR"] == "127.0.0.1" and 1024 <= int(e["MASTER_PORT"]) < 65536
    assert e["argv"] == ["--gpus", "3", "--steps", "2"]


def test_spawn_ranks_first_failure_stops_the_others_and_is_reported():
    child = "import os, sys, time; sys.exit(3) if os.environ['RANK'] == '1' else time.sleep(120)"
    t0 = time.monotonic()
    rc, _ = bench.spawn_ranks([sys.executable, "-c", child], 3)
    assert rc == 3
    assert time.monotonic() - t0 < 60, "the surviving ranks were not stopped"


def test_spawn_ranks_timeout():
    rc, _ = bench.spawn_ranks([sys.executable, "-c", "import time; time.sleep(120)"], 2, timeout=1.0)
    assert rc == 124


def test_bench_gpus_2_without_a_gpu_fails_loudly_with_empty_stdout():
    """Here (no GPU) the two self-launched ranks must fail, the parent must exit non-zero and
    print nothing on stdout (the one JSON line is all stdout may ever carry)."""
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--bodies", "4096"], capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():   # on a GPU box this call would simply work
        return
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "needs a GPU" in r.stderr
