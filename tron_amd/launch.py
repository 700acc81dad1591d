"""One process per GPU without an external launcher, and the only cross-rank traffic the path
needs: a barrier and a max-reduce of wall times, over gloo on the host.

Slices are independent (src/tron.cu:732-783) and the reference's own multi-GPU sketch is a slice
round-robin with no inter-GPU traffic (src/tron.cu:582-597,735-736), so no RCCL group is opened
anywhere: ranks meet on the host only to line up their clocks.

`spawn_ranks` must run BEFORE the calling process touches the GPU (no torch.cuda / HIP call):
children are fresh interpreters started with subprocess, never an exec of a GPU-initialised one.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def bind_near_gpu(local_rank: int, sysroot: str = "/sys", bus_id: str | None = None):
    """Binds the calling process (and every thread it starts later) to the CPUs of the NUMA node its GPU hangs off: a rank feeds its
    GPU from host buffers (python -m tron_amd.shard, the host-buffer entry points), and on a two-socket 8-GPU node half the ranks would
    otherwise run across the socket link.  sysfs: <sysroot>/bus/pci/devices/<bus id>/numa_node -> .../node<N>/cpulist (parsed by
    tron_host_numa_cpulist, as for the in-process workers of tron_recon_radial2d_multi).  Best effort: returns the CPUs bound to, or []
    when the node is unknown (single-node hosts, VMs) or none of its CPUs is available to this process -- the process then stays put.
    `bus_id` is looked up through HIP (tron_device_pci_bus_id) unless given, so call it in the rank, not in the spawning parent."""
    from tron_amd import lib
    if bus_id is None:
        bus_id = lib.device_pci_bus_id(local_rank)
    cpus = sorted(set(lib.numa_cpulist(sysroot, bus_id)) & os.sched_getaffinity(0))
    if cpus:
        os.sched_setaffinity(0, cpus)
    return cpus


DEFAULT_TIMEOUT_S = 1800.0   # a run that has not finished by then is ended (gloo's own rendezvous timeout is 30 minutes too)


def spawn_ranks(argv, world: int, timeout=DEFAULT_TIMEOUT_S, attempts=3):
    """Starts `world` children running ``python argv...`` with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set.
    Returns (exit_code, rank0_stdout): exit_code is 0 only if EVERY rank exited 0; rank 0's stdout is
    captured (not forwarded) so the caller can decide to print it only for a complete run.
    The rendezvous port is picked by binding and closing a socket, so another process can take it before rank 0 binds it:
    a run whose rank 0 dies with "address already in use" is started again on a fresh port (up to `attempts` times)."""
    code, out, err0 = 1, "", ""
    for _ in range(max(1, attempts)):
        code, out, err0 = _spawn_once(argv, world, timeout)
        if code == 0 or not any(t in err0 for t in ("EADDRINUSE", "Address already in use", "address already in use")):
            break
    if err0:                                     # the last attempt's rank-0 stderr, also when every attempt lost its port
        sys.stderr.write(err0)
    return code, out


def _spawn_once(argv, world: int, timeout):
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=rank_env(r, world, port),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=subprocess.PIPE if r == 0 else None, text=True))
    import threading
    import time
    out0_parts, err0_parts = [], []
    reader = threading.Thread(target=lambda: out0_parts.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    ereader = threading.Thread(target=lambda: err0_parts.append(procs[0].stderr.read()), daemon=True)
    ereader.start()
    code = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        # a rank that dies before the rendezvous would leave the others waiting for it (gloo's own timeout is 30 min):
        # watch all of them and end the run as soon as one fails
        while True:
            states = [p.poll() for p in procs]
            bad = [rc for rc in states if rc not in (None, 0)]
            if bad:
                code = bad[0]
                break
            if all(rc == 0 for rc in states):
                break
            if deadline is not None and time.monotonic() > deadline:
                code = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:          # exact PIDs we started, nothing else
            if p.poll() is None:
                p.kill()
            p.wait()
    reader.join(timeout=10)
    ereader.join(timeout=10)
    return code, (out0_parts[0] if out0_parts else ""), (err0_parts[0] if err0_parts else "")


class HostGroup:
    """Barrier + max-reduce over gloo (CPU tensors).  world == 1 needs no process group."""

    def __init__(self, rank: int, world: int):
        self.rank, self.world = rank, world
        self._dist = None
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self._dist is not None:
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None
