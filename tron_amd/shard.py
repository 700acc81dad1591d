"""Slice sharding across GPUs: one process per GPU, no collective on the data path, a host-side
gather assembles the output volume.

The reference's only multi-GPU provision is a compiled-out slice round-robin over streams
(``MULTI_GPU``, src/tron.h:49, src/tron.cu:582-597,735-736).  Slices ``z`` are independent
(src/tron.cu:732-783), so rank r of W takes the contiguous block ``[r*nz//W, (r+1)*nz//W)``;
its input is the spoke range those windows touch (neighbouring ranks overlap by
``npe1work - prof_slide`` spokes, read-only, taken from the host buffer), and the angle index
stays global (``pe + skip_angles + z*prof_slide``), so results do not depend on W.

Run as a program under torch.distributed.run to reconstruct a .ra file on several GPUs::

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m tron_amd.shard -a -G -u 0.4 -d 21 in.ra out.ra
"""
from __future__ import annotations

import os
import sys

import numpy as np


def partition(nz: int, world: int, rank: int):
    """Contiguous slice block of `rank`: (zfirst, zcount)."""
    z0 = rank * nz // world
    z1 = (rank + 1) * nz // world
    return z0, z1 - z0


def spoke_range(prof_slide: int, npe1work: int, zfirst: int, zcount: int):
    """(first spoke, number of spokes) of the stream that slices [zfirst, zfirst+zcount) read
    (src/tron.cu:738-739,747-748)."""
    if zcount <= 0:
        return zfirst * prof_slide, 0
    return zfirst * prof_slide, (zcount - 1) * prof_slide + npe1work


def hip_compute(cfg, dims):
    """Default per-rank compute: the HIP path through the C ABI (no CPU fallback)."""
    from . import lib
    plan = lib.Plan(cfg, dims)

    def run(flat_in, zfirst, zcount, out):
        plan.recon(flat_in, zfirst=zfirst, zcount=zcount, out=out)
    run.close = plan.close
    run.plan = plan
    return run


def gather_blocks(block, zc: int, slice_elems: int, nz: int, rank: int, world: int, group=None):
    """Host-side gather of per-rank image blocks (rank r holds slices partition(nz, world, r)) on rank 0; returns the
    assembled flat output there, None elsewhere.  The only cross-rank traffic of a sharded run."""
    if world == 1:
        return np.ascontiguousarray(block[: nz * slice_elems])
    import torch
    import torch.distributed as dist
    maxc = max(partition(nz, world, r)[1] for r in range(world))
    padded = np.zeros(maxc * slice_elems, np.complex64)
    padded[: zc * slice_elems] = block[: zc * slice_elems]
    t = torch.from_numpy(padded.view(np.float32))
    if rank == 0:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.gather(t, parts, dst=0, group=group)
        out = np.zeros(nz * slice_elems, np.complex64)
        for r, part in enumerate(parts):
            rz0, rzc = partition(nz, world, r)
            out[rz0 * slice_elems: (rz0 + rzc) * slice_elems] = part.numpy().view(np.complex64)[: rzc * slice_elems]
        return out
    dist.gather(t, None, dst=0, group=group)
    return None


def recon_file_sharded(infile, cfg, rank: int, world: int, compute_block, group=None):
    """One rank's share of `tron -a` on a .ra file: reads ONLY the spokes its slice block touches (spoke_range),
    reconstructs them with ``compute_block(block_in, zfirst, zcount) -> zcount images`` and gathers on rank 0.
    Returns (flat output or None, dims, bytes this rank read from the file)."""
    from . import lib, ra
    hdr = ra.read_header(infile)
    dims = lib.derive_dims(cfg, hdr.dims)
    z0, zc = partition(dims.nz, world, rank)
    s0, ns = spoke_range(dims.prof_slide, dims.npe1work, z0, zc)
    block_in, _, nbytes = ra.read_spokes(infile, s0, ns)
    slice_elems = dims.nt * dims.nx * dims.ny
    block_out = compute_block(block_in, z0, zc, dims) if zc > 0 else np.zeros(0, np.complex64)
    return gather_blocks(block_out, zc, slice_elems, dims.nz, rank, world, group), dims, nbytes


def recon_sharded(flat_in: np.ndarray, out_elems: int, slice_elems: int, nz: int, compute,
                  rank: int = 0, world: int = 1, group=None):
    """Every rank reconstructs its block with ``compute(flat_in, zfirst, zcount, out)`` (which
    writes slices at their global offsets into the full-size ``out``); rank 0 gathers the blocks
    on the host and returns the assembled flat output (other ranks return None)."""
    out = np.zeros(out_elems, np.complex64)
    z0, zc = partition(nz, world, rank)
    if zc > 0:
        compute(flat_in, z0, zc, out)
    if world == 1:
        return out
    import torch
    import torch.distributed as dist
    maxc = max(partition(nz, world, r)[1] for r in range(world))
    block = np.zeros(maxc * slice_elems, np.complex64)
    block[: zc * slice_elems] = out[z0 * slice_elems: (z0 + zc) * slice_elems]
    t = torch.from_numpy(block.view(np.float32))
    if rank == 0:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.gather(t, parts, dst=0, group=group)
        for r, part in enumerate(parts):
            rz0, rzc = partition(nz, world, r)
            out[rz0 * slice_elems: (rz0 + rzc) * slice_elems] = part.numpy().view(np.complex64)[: rzc * slice_elems]
        return out
    dist.gather(t, None, dst=0, group=group)
    return None


def main(argv=None):
    import getopt
    argv = sys.argv[1:] if argv is None else argv
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)     # host-side gather only
    from . import launch, lib, ra
    if world > 1:
        try:
            launch.bind_near_gpu(local_rank)         # this rank reads its spokes from the file and feeds its GPU: stay on that GPU's socket
        except Exception:                            # (no GPU here -- the CPU tests -- or no NUMA information: stay put)
            pass
    opts, args = getopt.getopt(argv, "3aB:d:g:Ghi:k:o:r:s:T:u:v")      # the reference's flags, src/tron.cu:822
    # same default as the tron binary (tron_main.cpp): fast Kaiser-Bessel unless TRON_OPTIONS holds kb=exact
    kw = dict(device=local_rank, kb_mode=lib.KB_EXACT if "kb=exact" in os.environ.get("TRON_OPTIONS", "").split(",") else lib.KB_FAST)
    for o, v in opts:
        if o == "-a": kw["adjoint"] = 1
        elif o == "-G": kw["golden_angle"] = 1
        elif o == "-3": kw["koosh"] = 1
        elif o == "-v": kw["verbose"] = 1 if rank == 0 else 0
        elif o == "-d": kw["prof_slide"] = int(v)
        elif o == "-s": kw["skip_angles"] = int(v)
        elif o == "-i": kw["niter"] = int(v)
        elif o == "-k": kw["kernwidth"] = float(v)
        elif o == "-o": kw["gridos"] = float(v)
        elif o == "-u": kw["data_undersamp"] = float(v)
    if not args:
        print("usage: python -m tron_amd.shard [tron flags] <infile.ra> [outfile.ra]", file=sys.stderr)
        return 1
    infile, outfile = args[0], (args[1] if len(args) > 1 else "img_tron.ra")
    if not kw.get("adjoint"):
        raise SystemExit("sharding is defined for the adjoint (-a): forward runs have one image")
    hdr = ra.read_header(infile)
    if hdr.eltype == ra.RA_TYPE_COMPLEX and hdr.elbyte == 4:
        kw["input_half"] = 1
    cfg = lib.default_config(**kw)
    plans = {}

    def compute_block(block_in, z0, zc, dims):
        # the plan is made from the FULL file's dims (global slice and angle indices); the rank holds only its own spokes
        plan = plans.setdefault("p", lib.Plan(cfg, dims))
        return plan.recon_block(block_in, z0, zc)                   # = tron_recon_radial2d_block

    out, dims, _ = recon_file_sharded(infile, cfg, rank, world, compute_block)
    if "p" in plans:
        plans["p"].close()
    if rank == 0:
        ra.write(outfile, out.reshape(tuple(int(x) for x in dims.out_dims), order="F"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
