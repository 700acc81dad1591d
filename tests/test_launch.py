"""`bench.py --gpus N` starts its own ranks (tron_amd/launch.py): the spawner, the gloo barrier / max-reduce the
ranks use instead of an RCCL group, and the refusal to print a line for an incomplete run -- all on CPU."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from tron_amd import launch
from tron_amd.shard import partition

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(tmp_path, body):
    f = tmp_path / "child.py"
    f.write_text("import os, sys\nsys.path.insert(0, %r)\n" % ROOT + textwrap.dedent(body))
    return str(f)


@pytest.mark.timeout(300)
def test_spawn_ranks_env_barrier_and_max(tmp_path):
    child = _child(tmp_path, """
        from tron_amd import launch
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
        g = launch.HostGroup(rank, world)
        g.barrier()
        slow = g.max(1.0 + rank)            # max over ranks of a per-rank time
        tot = g.sum(float(rank + 1))
        g.close()
        if rank == 0:
            print('{"metric": "x", "n_gpus": %d, "max": %g, "sum": %g}' % (world, slow, tot))
    """)
    code, out = launch.spawn_ranks([child], 3, timeout=240)
    assert code == 0
    line = json.loads(out.strip().splitlines()[-1])
    assert line == {"metric": "x", "n_gpus": 3, "max": 3.0, "sum": 6.0}


@pytest.mark.timeout(300)
def test_spawn_ranks_reports_a_failing_rank(tmp_path):
    child = _child(tmp_path, """
        if os.environ["RANK"] == "1":
            sys.exit(7)
        print('{"metric": "x", "n_gpus": 2}')
    """)
    code, out = launch.spawn_ranks([child], 2, timeout=120)
    assert code == 7 and '"n_gpus": 2' in out        # rank 0's line exists but the caller must not print it


@pytest.mark.timeout(600)
def test_bench_gpus_2_without_gpus_fails_loudly():
    """On a box without GPUs `python bench.py --gpus 2` must exit non-zero and print no result line
    (round 1 silently ran one rank and printed n_gpus: 1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=540)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present: this is the no-GPU behaviour test")
    assert r.returncode != 0
    assert '"metric"' not in r.stdout
    assert "no result line printed" in r.stderr


def test_strong_scaling_partition_matches_config4():
    # BASELINE config 4: 256 slices in total over 8 GPUs = 32 contiguous slices each, global angle index z*npe
    blocks = [partition(256, 8, r) for r in range(8)]
    assert blocks == [(32 * r, 32) for r in range(8)]


@pytest.mark.timeout(120)
def test_spawn_ranks_does_not_wait_for_a_rendezvous_that_cannot_happen(tmp_path):
    """Rank 1 dies before the rendezvous; rank 0 would sit in gloo's 30-minute timeout.  The spawner ends the run."""
    child = _child(tmp_path, """
        if os.environ["RANK"] == "1":
            sys.exit(3)
        from tron_amd import launch
        launch.HostGroup(0, 2)          # blocks: its peer is gone
    """)
    import time
    t0 = time.monotonic()
    code, out = launch.spawn_ranks([child], 2, timeout=100)
    assert code == 3 and time.monotonic() - t0 < 60


def test_spawn_ranks_retries_on_a_taken_port(tmp_path, monkeypatch):
    """free_port() closes its probe socket before rank 0 binds the port, so another process can take it; a run whose rank 0
    dies with "address already in use" is started again on a fresh port."""
    marker = tmp_path / "tried"
    child = _child(tmp_path, """
        marker = %r
        if os.environ["RANK"] == "0" and not os.path.exists(marker):
            open(marker, "w").write("x")
            sys.stderr.write("RuntimeError: The server socket has failed to listen on any local network address. "
                             "EADDRINUSE: address already in use\\n")
            sys.exit(1)
        if os.environ["RANK"] == "0":
            print('{"metric": "x", "n_gpus": 2}')
    """ % str(marker))
    ports = []
    real = launch.free_port
    monkeypatch.setattr(launch, "free_port", lambda: ports.append(real()) or ports[-1])
    code, out = launch.spawn_ranks([child], 2, timeout=120)
    assert code == 0 and '"n_gpus": 2' in out and len(ports) == 2


def test_spawn_ranks_has_a_finite_default_timeout():
    import inspect
    assert inspect.signature(launch.spawn_ranks).parameters["timeout"].default == launch.DEFAULT_TIMEOUT_S < 3600


@pytest.mark.timeout(120)
def test_ranks_bind_to_the_cpus_of_their_gpus_numa_node(tmp_path):
    """launch.bind_near_gpu: a rank that feeds its GPU from host buffers stays on that GPU's socket (PCI bus id -> sysfs numa_node ->
    cpulist -> sched_setaffinity), here against a fake sysfs tree with two "GPUs" on two nodes that split this machine's CPUs; an
    unknown node or a node none of whose CPUs this process may use leaves the process where it is.  (In a child: affinity is
    inherited by everything pytest starts later.)"""
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 2:
        pytest.skip("needs two CPUs")
    half = len(cpus) // 2
    nodes = {"0000:01:00.0": (0, cpus[:half]), "0000:c1:00.0": (1, cpus[half:])}
    for bus, (node, cl) in nodes.items():
        dev = tmp_path / "bus" / "pci" / "devices" / bus
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text(f"{node}\n")
        nd = tmp_path / "devices" / "system" / "node" / f"node{node}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(",".join(str(c) for c in cl) + "\n")
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:e1:00.0"            # a GPU whose node is unknown (-1: single-node hosts, VMs)
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("-1\n")
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:f1:00.0"            # ... and one on a node whose CPUs this process may not use
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("7\n")
    nd = tmp_path / "devices" / "system" / "node" / "node7"
    nd.mkdir(parents=True)
    (nd / "cpulist").write_text("100000-100003\n")
    child = _child(tmp_path, """
        import json
        from tron_amd import launch
        root = sys.argv[1]
        before = sorted(os.sched_getaffinity(0))
        res = {}
        for bus in ("0000:e1:00.0", "0000:f1:00.0"):
            res[bus] = [launch.bind_near_gpu(0, sysroot=root, bus_id=bus), sorted(os.sched_getaffinity(0)) == before]
        for bus in ("0000:C1:00.0", "0000:01:00.0"):                        # HIP reports upper-case hex
            os.sched_setaffinity(0, before)
            res[bus] = [launch.bind_near_gpu(0, sysroot=root, bus_id=bus), sorted(os.sched_getaffinity(0))]
        print(json.dumps(res))
    """)
    r = subprocess.run([sys.executable, child, str(tmp_path)], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["0000:e1:00.0"] == [[], True] and res["0000:f1:00.0"] == [[], True]
    assert res["0000:C1:00.0"] == [cpus[half:], cpus[half:]]
    assert res["0000:01:00.0"] == [cpus[:half], cpus[:half]]
