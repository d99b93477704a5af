"""bench.py --gpus N started bare (no torch.distributed.run around it): the parent starts the N ranks itself, as fresh child
processes, before anything touches the GPU -- argument handling only, no GPU needed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_importing_bench_does_not_import_torch():
    code = "import sys; sys.path.insert(0, %r); import bench; assert 'torch' not in sys.modules, 'bench.py imported torch at module level'" % ROOT
    subprocess.run([sys.executable, "-c", code], check=True)


def test_needs_launch_rules():
    import bench
    assert bench.needs_launch(8, {})
    assert bench.needs_launch(2, {"PATH": "/bin"})
    assert not bench.needs_launch(1, {})
    assert not bench.needs_launch(8, {"WORLD_SIZE": "8", "RANK": "3"})  # a rank torch.distributed.run started
    assert not bench.needs_launch(8, {"RANK": "0"})


def test_launch_command_is_the_documented_form():
    import bench
    cmd = bench.launch_command(["--gpus", "4", "--steps", "20", "--warmup", "5"], 4, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]  # the caller's flags reach every rank unchanged
    assert 1024 < bench.free_port() < 65536


def test_bare_two_rank_run_starts_two_ranks_and_relays_their_exit_code():
    """No GPU here: each rank must get as far as its own device check (so WORLD_SIZE = 2 was set for it by the launcher the parent
    started) and the parent must come back non-zero -- never the old 'launch with: python -m torch.distributed.run' refusal."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    import torch
    if torch.cuda.is_available():  # on a GPU box the bare run is exercised by the rehearsal line under profiles/ instead
        return
    assert p.returncode != 0
    assert "launch with" not in p.stderr
    assert p.stderr.count("bench.py needs a HIP device") >= 2, p.stderr[-2000:]
    assert "--gpus 2 but WORLD_SIZE" not in p.stderr


def test_mismatched_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=4" in p.stderr
