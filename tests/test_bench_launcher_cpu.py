"""bench.py --gpus N without an external launcher: the parent starts N ranks as a child process, relays rank 0's line
and refuses to pass off a run whose world size is not N (north_star "rays/sec at 1, 2, 4 and 8 GPUs").
CPU only: the ranks here are a stand-in script on gloo, the launcher logic is bench.py's own."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_parent_only_when_no_launcher_has_run():
    assert bench.needs_launch(8, {})
    assert not bench.needs_launch(1, {})
    assert not bench.needs_launch(8, {"WORLD_SIZE": "8"})      # already a rank of torch.distributed.run
    assert not bench.needs_launch(2, {"WORLD_SIZE": "1"})      # a rank: check_world() refuses it, it is not relaunched


def test_child_command_is_one_rank_per_gpu_with_the_same_arguments():
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "2", "--workload", "fern"], 29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "2", "--workload", "fern"]


STAND_IN = """
import json, os, sys
import torch.distributed as dist
dist.init_process_group("gloo")
world, rank = dist.get_world_size(), dist.get_rank()
dist.barrier()
mode = sys.argv[1]
if mode == "fail" and rank == 1:
    sys.exit(7)
if rank == 0 and mode != "silent":
    print(json.dumps({"n_gpus": world if mode != "lie" else 1, "world": world, "argv": sys.argv[2:]}), flush=True)
dist.destroy_process_group()
"""


@pytest.fixture
def stand_in(tmp_path):
    p = tmp_path / "ranks.py"
    p.write_text(textwrap.dedent(STAND_IN))
    return str(p)


def test_two_ranks_are_started_and_rank0_line_relayed(stand_in, capfd):
    rc = bench.launch_ranks(2, ["ok", "--steps", "2"], script=stand_in)
    out = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")]
    assert rc == 0 and len(out) == 1
    line = json.loads(out[0])
    assert line["n_gpus"] == 2 and line["world"] == 2 and line["argv"] == ["--steps", "2"]


def test_a_run_without_an_n_gpu_line_is_an_error(stand_in, capfd):
    assert bench.launch_ranks(2, ["silent"], script=stand_in) == 3
    assert bench.launch_ranks(2, ["lie"], script=stand_in) == 3      # a line that says n_gpus 1 is not an N = 2 result
    capfd.readouterr()


def test_a_failing_rank_fails_the_run(stand_in, capfd):
    assert bench.launch_ranks(2, ["fail"], script=stand_in) != 0
    capfd.readouterr()


def test_world_size_mismatch_is_refused_before_any_gpu_work():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and "--gpus 2 but WORLD_SIZE is 3" in r.stderr and r.stdout.strip() == ""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and r.stdout.strip() == ""
