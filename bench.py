#!/usr/bin/env python3
"""Throughput of the radial gridding reconstruction (adjoint NUFFT) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--coils C] [--slices S] [--kb fast|exact] [--scaling weak|strong]

One "step" = one pass of the hot path (density compensation + Kaiser-Bessel gridding + 2-D FFT +
crop/deapodise/coil-combine) over one batch of S slices of synthetic golden-angle radial k-space,
512 readout x 402 spokes x C coils per slice onto a 512^2 grid -> 256^2 images, inputs and
outputs resident in HBM.  For N > 1 there is one process per GPU -- started by torch.distributed.run
(RANK/LOCAL_RANK/WORLD_SIZE in the environment) or, when those are absent, spawned by this script
itself before anything touches the GPU.  Slices are independent, so there is no collective on the
data path; ranks meet over gloo on the host only for the barrier and the max-over-ranks time.
--scaling weak (default): every rank its own S slices; --scaling strong: S slices in total, rank r
takes the contiguous block partition(S, N, r) (BASELINE config 4 literally).

Prints ONE JSON line (rank 0): whole-job slices/s plus
  roofline      the dominant kernel's algorithmic bytes / its mean hipEvent duration, vs 8 TB/s
  cpu_baseline  the CPU oracle (a port of the reference's algorithm) timed on this host on a
                bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NRO, NPE, NXOS, NX = 512, 402, 512, 256
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def algorithmic_bytes(nc, half=False):
    """bytes that must cross HBM per coil-slice and per slice."""
    grid = (4 if half else 8) * NRO * NPE + 8 * NXOS * NXOS   # read samples once + write grid once
    fft = 2 * 8 * NXOS * NXOS                       # one read + one write
    post_cs = 8 * NX * NX                           # read centre crop
    per_cs = grid + fft + post_cs                   # 8 462 336
    per_slice = nc * per_cs + 8 * NX * NX           # + image write
    return dict(grid=grid, fft=fft, post=post_cs + 8 * NX * NX / nc, per_cs=per_cs, per_slice=per_slice)


def cpu_baseline(nc, sample_slices, undersamp=0.7852):
    """The CPU oracle (oracle/, a port of the reference's own algorithm: point-driven gather
    over every spoke, src/tron.cu:465-536) on `sample_slices` slices of the same shape."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    from oracle import pyoracle
    threads = os.cpu_count() or 1
    data = synth.kspace(nc, NRO, NPE * sample_slices, seed=synth.SEED_BASE)
    t0 = time.perf_counter()
    out, p = pyoracle.recon(data, adjoint=1, golden=1, data_undersamp=undersamp, prof_slide=NPE)
    dt = time.perf_counter() - t0
    assert p.nz == sample_slices and p.npe1work == NPE
    return dict(value=sample_slices / dt, unit="slices/s", cores=threads, kind="port",
                sample=f"{sample_slices} slice(s) x {nc} coil(s) of 512x{NPE} golden-angle through the full oracle pipeline "
                       f"(OpenMP over grid points, {threads} threads), {dt:.1f} s wall"), out


def irt_baseline(nc):
    """The comparator the reference measures itself against (README.md:20-21, "75x"): Fessler's IRT NUFFT, bundled as
    MATLAB under contrib/irt, here its C++/OpenMP restatement oracle/irt_nufft.cpp (double precision, min-max KB
    interpolation with 16 non-zeros per sample), used as the reference's own scripts use it
    (src/RUNME4_others_grid_slcmt.m:112-130): nufft_init PER SLICE (the trajectory rotates with the window), density
    weights multiplied in by the caller, nufft_adj per coil, root-sum-of-squares.  Timed on all host cores (slices dealt
    to threads) and on one core."""
    from oracle import irt_cpp
    cores = os.cpu_count() or 1
    threads = min(cores, 64)                      # ~70 MB of interpolation tables per slice in flight
    one = irt_cpp.bench_golden(NX, NRO, NPE, nc, 2, 1)
    per_slice = one["wall_s"] / 2
    n_all = max(threads, min(4 * threads, int(12.0 / per_slice) * threads // max(threads, 1) or threads))
    n_all = max(threads, min(n_all, 4 * threads))
    allc = irt_cpp.bench_golden(NX, NRO, NPE, nc, n_all, threads)
    return dict(value=round(n_all / allc["wall_s"], 3), unit="slices/s", cores=threads, kind="port",
                value_1core=round(2 / one["wall_s"], 4),
                init_share=round(allc["init_s"] / (allc["init_s"] + allc["adj_s"]), 3),
                sample=f"contrib/irt restated in C++/OpenMP (oracle/irt_nufft.cpp): {n_all} slices x {nc} coils of 512x{NPE} golden-angle on "
                       f"{threads} threads ({allc['wall_s']:.1f} s wall; nufft_init per slice = {allc['init_s'] / n_all:.2f} s, {nc} nufft_adj = "
                       f"{allc['adj_s'] / n_all:.2f} s per slice and thread), and 2 slices on 1 thread ({one['wall_s']:.1f} s); double precision")


def forward_bench(args, rank, local_rank, world, torch, group, lib):
    """Secondary line (not the headline metric): the forward NUFFT, src/tron.cu:639-649 -- pad, deapodise, FFT, degrid --
    on `--slices` images of 256^2 x C coils per step -> 512 readout x 512 golden-angle spokes each, device resident."""
    import ctypes
    import numpy as np
    nc, nimg = args.coils, min(args.slices, 64)
    cfg = lib.default_config(adjoint=0, golden_angle=1, device=local_rank, kb_mode=lib.KB_FAST if args.kb == "fast" else lib.KB_EXACT)
    dims = lib.derive_dims(cfg, (nc, 1, NX, NX, 1))
    nro, npe = dims.nro, dims.npe1work
    plan = lib.Plan(cfg, dims)
    g = torch.Generator(device="cuda")
    g.manual_seed(0x54524F4E + 100 + rank)
    imgs = torch.rand(2 * nc * NX * NX * nimg, device="cuda", generator=g, dtype=torch.float32) * 2 - 1
    out = torch.empty(2 * nc * nro * npe * nimg, device="cuda", dtype=torch.float32)
    d_in, d_out = ctypes.c_void_p(imgs.data_ptr()), ctypes.c_void_p(out.data_ptr())

    def fence():
        torch.cuda.synchronize()
        group.barrier()
        torch.cuda.synchronize()
    torch.cuda.synchronize()          # inputs were generated on torch's stream, the library runs on its own
    for _ in range(args.warmup):
        plan.forward_device(d_out, d_in, nimg)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.forward_device(d_out, d_in, nimg)
    fence()
    dt = time.perf_counter() - t0
    plan.sync()
    dt = group.max(dt)
    value = world * nimg * args.steps / dt
    result = None
    if rank == 0:
        plan.timing(True)
        plan.timing_reset()
        reps = 3
        for _ in range(reps):
            plan.forward_device(d_out, d_in, nimg)
        stages = {}
        for st, name in {lib.STAGE_PRE: "pre", lib.STAGE_FFT: "fft", lib.STAGE_DEGRID: "degrid"}.items():
            ms, n = plan.timing_get(st)
            if n:
                stages[name] = (ms, n)
        plan.timing(False)
        # SURVEY 8(d), forward: image + padded grid write + FFT read/write + grid read + samples
        per_ci = 8 * NX * NX + 8 * NXOS * NXOS + 16 * NXOS * NXOS + 8 * NXOS * NXOS + 8 * nro * npe
        ms, n = stages["degrid"]
        alg = 8 * NXOS * NXOS + 8 * nro * npe                    # degridding alone: grid read + sample write
        achieved = alg * nc * nimg * reps / n / (ms / n * 1e-3) / 1e9
        fwd_units = nc * nimg * reps / n
        ftraffic, fstale, fnote = traffic_capture(workload_key(args), ("degrid_stream_kernel", "degrid_tile_kernel", "degrid_kernel"), fwd_units)
        err = None
        if not args.no_check:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle import pyoracle
            host = imgs[: 2 * nc * NX * NX].cpu().numpy().view(np.complex64).reshape((nc, 1, NX, NX, 1), order="F")
            want, _ = pyoracle.recon(host, adjoint=0, golden=1)
            got = out[: 2 * nc * nro * npe].cpu().numpy().view(np.complex64)
            err = float(np.linalg.norm(got - want.reshape(-1, order="F")) / np.linalg.norm(want))
        gbps = per_ci * nc * value / world / 1e9
        result = {
            "metric": "2D images/sec degridded (256^2 image -> 512^2 grid -> 512x512 golden-angle spokes) + achieved HBM GB/s",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"forward NUFFT: {nimg} images/GPU/step x {nc} coils, 256^2 -> 512^2 grid -> {nro} readout x {npe} golden-angle spokes (tron -G)",
                       "coils": nc, "images_per_gpu": nimg, "kb_mode": args.kb},
            "algorithmic_gbps_per_gpu": round(gbps, 1), "algorithmic_frac_of_peak": round(gbps / HBM_PEAK_GBPS, 4),
            "parity_rel_l2_vs_oracle": err,
            "roofline": {"bound": "hbm", "kernel": plan.degrid_kernel_name(), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": ftraffic, "traffic_stale": fstale, "traffic_source": fnote,
                         "bytes_per_launch": int(alg * fwd_units), "units_per_launch": fwd_units, "launch_ms": round(ms / n, 4),
                         "stage_share": {k: round(v[0] / sum(x[0] for x in stages.values()), 3) for k, v in stages.items()}},
            "cpu_baseline": None,
        }
    plan.close()
    group.close()
    if rank == 0:
        print(json.dumps(result), flush=True)


def one_coil_field(args, lib, torch, local_rank, undersamp):
    """SURVEY 8(d) M0 names nc in {1, 8}: the driver's line is the 8-coil workload, and this is the SAME measurement at one coil --
    same slices, spokes, fences and clock, a plan and buffers of its own -- as a field of that line, with the
    gridding stage's roofline fraction and a spot check against the oracle."""
    import ctypes
    import numpy as np
    nz = args.slices
    cfg = lib.default_config(adjoint=1, golden_angle=1, data_undersamp=undersamp, prof_slide=NPE, device=local_rank, kb_mode=lib.KB_FAST,
                             chunk_slices=args.chunk)
    dims = lib.derive_dims(cfg, (1, 1, NRO, NPE * nz, 1))
    plan = lib.Plan(cfg, dims)
    g = torch.Generator(device="cuda")
    g.manual_seed(0x54524F4E + 7)
    kspace = torch.rand(2 * NRO * NPE * nz, device="cuda", generator=g, dtype=torch.float32) * 2 - 1
    images = torch.empty(2 * NX * NX * nz, device="cuda", dtype=torch.float32)
    d_in, d_out = ctypes.c_void_p(kspace.data_ptr()), ctypes.c_void_p(images.data_ptr())
    torch.cuda.synchronize()

    def step():
        plan.adjoint_device(d_out, d_in, 0, nz, combine=1)

    def fence():
        torch.cuda.synchronize()
        plan.sync()
    for _ in range(max(1, args.warmup)):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    value = nz * args.steps / dt
    n_sus = max(args.steps, int(min(args.sustain, 2.0) / (dt / args.steps)) + 1) if args.sustain > 0 else 0
    sus = None
    if n_sus:
        t0 = time.perf_counter()
        for _ in range(n_sus):
            step()
        fence()
        sus = nz * n_sus / (time.perf_counter() - t0)
    plan.timing(True)
    plan.timing_reset()
    for _ in range(3):
        step()
    ms, n = plan.timing_get(lib.STAGE_GRID)
    plan.timing(False)
    ab = algorithmic_bytes(1)
    grid_gbps = ab["grid"] * nz * 3 / n / (ms / n * 1e-3) / 1e9 if n else None
    err = None
    if not args.no_check:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import pyoracle
        host = kspace[: 2 * NRO * NPE].cpu().numpy().view(np.complex64).reshape((1, 1, NRO, NPE, 1), order="F")
        want, _ = pyoracle.recon(host, adjoint=1, golden=1, data_undersamp=undersamp, prof_slide=NPE)
        got = images[: 2 * NX * NX].cpu().numpy().view(np.complex64)
        err = float(np.linalg.norm(got - want.reshape(-1, order="F")) / np.linalg.norm(want))
        if not err <= 1e-5:
            raise SystemExit(f"one-coil output disagrees with the oracle: rel L2 {err:.3e}")
    kname = plan.grid_kernel_name()
    plan.close()
    alg = ab["per_slice"] * value / 1e9
    return dict(value=round(value, 1), unit="slices/s", sustained_slices_per_s=round(sus, 1) if sus else None, steps=args.steps,
                ms_per_step=round(dt / args.steps * 1e3, 3), algorithmic_frac_of_peak=round(alg / HBM_PEAK_GBPS, 4),
                target_40pct_slices_per_s=round(0.4 * HBM_PEAK_GBPS * 1e9 / ab["per_slice"], 0),
                roofline=dict(kernel=kname, achieved=round(grid_gbps, 1) if grid_gbps else None, peak=HBM_PEAK_GBPS, unit="GB/s",
                              frac=round(grid_gbps / HBM_PEAK_GBPS, 4) if grid_gbps else None, launch_ms=round(ms / n, 4) if n else None),
                parity_rel_l2_vs_oracle=err, workload=f"{nz} slices x 1 coil, otherwise the line's own workload")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--coils", type=int, default=8)
    ap.add_argument("--slices", type=int, default=256, help="slices per GPU per step (weak) or in total (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank its own --slices; strong: --slices in total, contiguous blocks per rank (BASELINE config 4)")
    ap.add_argument("--spokes", type=int, default=402, help="spokes per slice (402 = the metric's shape; BASELINE config 4 has 804)")
    ap.add_argument("--kb", choices=["fast", "exact"], default="fast")
    ap.add_argument("--half", action="store_true", help="k-space stored as complex-half in HBM (BASELINE config 5); fp32 accumulate and FFT")
    ap.add_argument("--chunk", type=int, default=0, help="slices per internal batch (0 = auto)")
    ap.add_argument("--cpu-slices", type=int, default=-1,
                    help="slices of the CPU-baseline sample (0 = skip; -1 = sized for about 12 s of wall time, 2..32 slices)")
    ap.add_argument("--no-irt", action="store_true", help="skip the contrib/irt comparator (cpu_baseline is then the oracle port)")
    ap.add_argument("--forward", action="store_true",
                    help="measure the forward (degridding) direction instead: --slices images of 256^2 -> 512 x 512 golden-angle spokes")
    ap.add_argument("--linear", action="store_true",
                    help="secondary line: linear-angle spokes (no -G): every slice shares one trajectory, so with few coils several "
                         "slices share one pass of the gridding kernel")
    ap.add_argument("--sustain", type=float, default=3.0,
                    help="seconds of back-to-back steps after the timed region for sustained_slices_per_s (0 = skip)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-fresh", action="store_true", help="skip fresh_trajectory_slices_per_s (tron_plan_retarget before every step)")
    ap.add_argument("--fresh-cycle", type=int, default=4,
                    help="fresh trajectories: step k grids the spokes whose angle index starts at (k mod this) x slices x spokes, i.e. every "
                         "step's tables are new; bounded because the reference multiplies the index in fp32 (src/tron.cu:509, SURVEY Q6)")
    ap.add_argument("--no-one-coil", action="store_true", help="skip the one_coil field of the default (8-coil) line")
    return ap.parse_args(argv)


def launch_ranks(args):
    """`python bench.py --gpus N` without an external launcher: N children, one per GPU, started before this
    process has made any torch / HIP call.  The JSON line of rank 0 is printed only if every rank exited 0."""
    from tron_amd import launch
    code, out0 = launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
    line = None
    for ln in out0.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if code != 0 or line is None:
        sys.stderr.write(f"bench.py: {args.gpus}-rank run failed (exit code {code}); no result line printed\n")
        return code if code != 0 else 1
    if json.loads(line).get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: result line does not carry the requested rank count; refusing to print it\n")
        return 1
    print(line, flush=True)
    return 0


GRID_STAGE_KERNELS = ("grid_arc_kernel", "grid_scatter_kernel", "grid_centre_kernel", "grid_binned_kernel", "grid_tile_kernel", "grid_reduce_parts_kernel")


def workload_key(args):
    """Names the capture file of a workload: profiles/traffic_<key>.json (tools/traffic.sh <key> [bench.py arguments])."""
    if args.forward:
        return f"forward_nc{args.coils}"
    return (f"nc{args.coils}_npe{args.spokes}_nz{args.slices}" + ("_half" if args.half else "") + ("_linear" if args.linear else "")
            + ("_exact" if args.kb == "exact" else ""))


def traffic_capture(key, kernel_prefixes, units_per_launch_now):
    """HBM bytes per launch of a stage (all its kernels) from the PMC capture of THIS workload under profiles/ -- used only
    when the capture was taken on THESE kernel sources (tools/traffic.sh stamps it with buildinfo.kernel_source_hash()) and
    with launches of THIS size.  Returns (bytes_per_launch or None, stale flag, note)."""
    from tron_amd.buildinfo import kernel_source_hash
    path = os.path.join(ROOT, "profiles", f"traffic_{key}.json")
    try:
        tj = json.load(open(path))
    except Exception:
        return None, True, f"no capture for this workload (tools/traffic.sh {key} ...)"
    if tj.get("source_hash") != kernel_source_hash():
        return None, True, f"capture {tj.get('source_hash')} != sources {kernel_source_hash()}"
    if abs(float(tj.get("units_per_launch", -1)) - float(units_per_launch_now)) > 0.5:
        return None, True, f"capture has {tj.get('units_per_launch')} units per launch, this run {units_per_launch_now}"
    per_launch = 0.0
    for name, c in tj.get("kernels", {}).items():
        if name.startswith(kernel_prefixes) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            # per launch of the STAGE: a stage launches each of its kernels once, so sum the per-dispatch means
            per_launch += (2.0 * c["FETCH_SIZE"]["kib"] / c["FETCH_SIZE"]["dispatches"]
                           + c["WRITE_SIZE"]["kib"] / c["WRITE_SIZE"]["dispatches"]) * 1024.0
    if per_launch > 0.0:
        return int(per_launch), False, tj.get("command", "")
    return None, True, "kernel not in capture"


_BURN_IN = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import synth
from tron_amd import lib
if lib.device_count() < 1:
    sys.exit(0)
for data, adjoint, fl in ((synth.kspace(2, 64, 60, seed=1), True, dict(golden_angle=1, data_undersamp=0.5, prof_slide=14)),
                          (synth.kspace(8, 512, 402 * 2, seed=2), True, dict(golden_angle=1, data_undersamp=0.7852, prof_slide=402)),
                          (synth.image(2, 256, seed=4), False, dict(golden_angle=1, data_undersamp=0.125))):
    a, _ = lib.recon(data, adjoint=adjoint, **fl)
    b, _ = lib.recon(data, adjoint=adjoint, **fl)
    assert np.isfinite(a).all() and np.array_equal(a, b)
"""


def burn_in():
    """The first GPU process on a freshly leased box is not like the later ones (tests/conftest.py): a child process
    runs each pipeline family twice first, so that a cold-start fault costs a retry there and not the measurement.  Untimed.
    Returns the attempts' exit codes (None = timed out), e.g. [0]: they go on the result line as burn_in_attempts, so a retry is seen."""
    import subprocess
    codes = []
    for attempt in range(3):
        try:
            r = subprocess.run([sys.executable, "-c", _BURN_IN % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            codes.append(None)
            continue
        codes.append(r.returncode)
        if r.returncode == 0:
            break
        sys.stderr.write(f"bench.py: burn-in attempt {attempt + 1} failed (rc {r.returncode}): {r.stderr[-300:]}\n")
    return codes


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ or os.environ.get("RANK", "0") == "0":
        if os.environ.get("TRON_BENCH_NO_BURN_IN") != "1" and "TRON_BENCH_BURNT" not in os.environ:
            codes = burn_in()                    # before this process (or the ranks it spawns) touches the GPU
            os.environ["TRON_BENCH_BURNT"] = json.dumps(codes)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # torch first: it brings the HIP runtime every later library (ours included) binds to
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    ndev = torch.cuda.device_count()
    share = os.environ.get("TRON_BENCH_SHARE_GPU") == "1"      # plumbing tests on a 1-GPU box only; flagged on the line
    if ndev <= local_rank:
        if not share:
            raise SystemExit(f"rank {rank}: {ndev} GPU(s) visible, --gpus {args.gpus} requested")
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    torch.zeros(1, device="cuda")
    from tron_amd import launch
    bound_cpus = []
    if world > 1 and not share:                  # one rank per GPU: run on that GPU's socket (best effort: a box without NUMA information stays put)
        try:
            bound_cpus = launch.bind_near_gpu(local_rank)
        except Exception as e:
            sys.stderr.write(f"bench.py: rank {rank}: no NUMA binding ({e})\n")
    group = launch.HostGroup(rank, world)      # gloo on the host: barrier + max of wall times, nothing on the data path

    import ctypes
    import numpy as np
    from tron_amd import lib
    from tron_amd.shard import partition

    if args.forward:
        return forward_bench(args, rank, local_rank, world, torch, group, lib)
    global NPE
    NPE = args.spokes
    undersamp = 0.7852 if NPE == 402 else (NPE + 0.5) / NRO    # tron -u: npe1work = int(nro*u), src/tron.cu:925
    nc = args.coils
    if args.scaling == "strong":
        zfirst, nz = partition(args.slices, world, rank)       # contiguous block of the ONE job (src/tron.cu:735-736 made contiguous)
        if nz < 1:
            raise SystemExit(f"rank {rank}: --scaling strong with {args.slices} slices leaves this rank without work")
        total_slices = args.slices
    else:
        zfirst, nz = 0, args.slices
        total_slices = world * nz
    golden = 0 if args.linear else 1
    cfg = lib.default_config(adjoint=1, golden_angle=golden, data_undersamp=undersamp, prof_slide=NPE, device=local_rank,
                             kb_mode=lib.KB_FAST if args.kb == "fast" else lib.KB_EXACT, chunk_slices=args.chunk,
                             input_half=1 if args.half else 0, skip_angles=zfirst * NPE)    # global angle index of this rank's first spoke
    dims = lib.derive_dims(cfg, (nc, 1, NRO, NPE * nz, 1))
    assert (dims.nz, dims.npe1work, dims.nxos, dims.nx) == (nz, NPE, NXOS, NX)
    t_plan0 = time.perf_counter()
    plan = lib.Plan(cfg, dims)
    plan_wall_s = time.perf_counter() - t_plan0          # tron_plan_create: NOT inside the timed region (a NUFFT plan: made once per trajectory)
    plan_times = plan.create_times()

    # synthetic k-space, uniform [-1,1) re/im, laid out [c + nc*(ro + nro*spoke)], resident in HBM
    g = torch.Generator(device="cuda")
    g.manual_seed(0x54524F4E + rank)
    kspace = torch.rand(2 * nc * NRO * NPE * nz, device="cuda", generator=g, dtype=torch.float32) * 2 - 1
    if args.half:
        kspace = kspace.to(torch.float16)        # round-to-nearest-even, as src/float16.cu
    images = torch.empty(2 * NX * NX * nz, device="cuda", dtype=torch.float32)
    d_in, d_out = ctypes.c_void_p(kspace.data_ptr()), ctypes.c_void_p(images.data_ptr())
    torch.cuda.synchronize()                     # inputs were generated on torch's stream, the library runs on its own

    def step():
        plan.adjoint_device(d_out, d_in, 0, nz, combine=1)

    def fence():
        torch.cuda.synchronize()
        plan.sync()                              # the library's stream is non-blocking: wait for it explicitly
        group.barrier()

    first_step_s = None
    if args.warmup >= 1:                         # the first warm-up step, clocked on its own: cold_slices_per_s
        fence()
        t_first0 = time.perf_counter()
        step()
        fence()
        first_step_s = time.perf_counter() - t_first0
    for _ in range(args.warmup - 1):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    plan.sync()   # surfaces a device-side error flag, if any
    dt = group.max(dt)
    value = total_slices * args.steps / dt

    # Sustained rate: the same step() back to back for >= --sustain seconds (the timed region above is tens of milliseconds on a
    # chip that holds its clock by power and settles slowly), the shader clock read in-kernel (s_memtime /
    # s_memrealtime, tron_plan_shader_clock) right behind the timed steps and right behind the sustained ones.
    sustained = None
    if args.sustain > 0:
        clock0 = plan.shader_clock_mhz()
        n_sus = max(args.steps, int(args.sustain / (dt / args.steps)) + 1)
        fence()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            step()
        clock1 = plan.shader_clock_mhz()         # queued behind the last step on the gridding stream; synchronises it
        fence()
        dt_sus = group.max(time.perf_counter() - t0)
        sustained = dict(value=round(total_slices * n_sus / dt_sus, 1), seconds=round(dt_sus, 3), steps=n_sus,
                         ratio_to_value=round(total_slices * n_sus / dt_sus / value, 4),
                         shader_clock_mhz_start=round(clock0, 0), shader_clock_mhz_end=round(clock1, 0))

    # Fresh trajectories: the timed steps above reuse ONE plan's angle tables, as a NUFFT plan made once per trajectory does; a
    # continuing golden-angle acquisition never repeats a window, and the reference's one published time (src/tron.cu:973-978,
    # RUNME4:219) and the cpu_baseline beside `value` both include their set-up.  Here EVERY step is preceded by tron_plan_retarget: new
    # angle index, new (cos, sin) table (host libm), new sorted lists and run tables (device, on the build stream, while the previous
    # step is still being gridded); nothing that does not depend on the angles is redone.  Same fences, same clock as `value`.
    fresh = None
    if golden and not args.no_fresh:
        cyc = max(2, args.fresh_cycle)
        skip_of = lambda k: (zfirst + (k % cyc) * nz) * NPE

        def fresh_step(k):
            plan.retarget(skip_of(k))
            step()
        for k in range(max(2, args.warmup)):
            fresh_step(k)
        fence()
        n_f = max(args.steps, int(1.0 / (dt / args.steps)) + 1)
        t0 = time.perf_counter()
        for k in range(n_f):
            fresh_step(k + 2)
        fence()
        dt_f = group.max(time.perf_counter() - t0)
        rt = plan.retarget_times()
        fresh = dict(value=round(total_slices * n_f / dt_f, 1), steps=n_f, seconds=round(dt_f, 3), ratio_to_value=round(total_slices * n_f / dt_f / value, 4),
                     distinct_trajectories=cyc, retarget_host_ms=round(rt["call"] * 1e3, 3), trig_table_host_ms=round(rt["trig"] * 1e3, 3),
                     note=f"tron_plan_retarget(skip_angles) before every step, skip = (k mod {cyc}) x {nz} x {NPE}: tables of the next step built on the "
                          "device beside the current step's gridding; bytes equal a fresh plan's (tests/test_gpu_retarget.py)")
        if rank == 0 and not args.no_check:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle import pyoracle
            pyoracle.set_threads(max(1, (os.cpu_count() or 1) // world))
            k_last = n_f + 1
            host = kspace[: 2 * nc * NRO * NPE].float().cpu().numpy().view(np.complex64).reshape((nc, 1, NRO, NPE, 1), order="F")
            want, _ = pyoracle.recon(host, adjoint=1, golden=1, data_undersamp=undersamp, prof_slide=NPE, skip_angles=skip_of(k_last))
            got = images[: 2 * NX * NX].cpu().numpy().view(np.complex64)
            fresh["parity_rel_l2_vs_oracle"] = float(np.linalg.norm(got - want.reshape(-1, order="F")) / np.linalg.norm(want))
            if not fresh["parity_rel_l2_vs_oracle"] <= 1e-5:
                raise SystemExit(f"retargeted output disagrees with the oracle: rel L2 {fresh['parity_rel_l2_vs_oracle']:.3e}")
        plan.retarget(zfirst * NPE)              # back to the plan's own angles for what follows
        step()
        fence()

    # per-kernel durations, measured live with hipEvents on the library's own stream
    ab = algorithmic_bytes(nc, args.half)
    roofline = None
    stages = {}
    if rank == 0:
        reps = max(2, min(args.steps, 5))
        # (the oracle checks above left the GPU idle for seconds and the chip's clock takes tens of milliseconds of load to settle,
        #  a quarter of a second of untimed steps first, so that the kernels are clocked as in the timed region)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.25:
            step()
        plan.sync()                              # (rank 0 only here: no barrier)
        plan.timing(True)
        plan.timing_reset()
        for _ in range(reps):
            step()
        for st, name in {lib.STAGE_GRID: "grid", lib.STAGE_FFT: "fft", lib.STAGE_POST: "post"}.items():
            ms, n = plan.timing_get(st)
            if n:
                stages[name] = (ms, n)
        plan.timing(False)
        tot = sum(ms for ms, _ in stages.values())
        dom = max(stages, key=lambda k: stages[k][0])
        ms, n = stages[dom]
        units_per_launch = nz * nc * reps / n                            # coil-slices per launch
        # "fft" = the fused pruned FFT + crop + deapodise + SoS when nxos=512 (then there is no separate post stage)
        alg = {"grid": ab["grid"], "fft": ab["fft"] + (ab["post"] if "post" not in stages else 0), "post": ab["post"]}[dom]
        bytes_per_launch = alg * units_per_launch
        achieved = bytes_per_launch / (ms / n * 1e-3) / 1e9
        kname = {"grid": plan.grid_kernel_name(),
                 "fft": "fft512_rows_kernel + fft512_cols_post_kernel" if "post" not in stages else "rocFFT 512x512 C2C inverse (batched)",
                 "post": "post_kernel"}[dom]
        # HBM bytes the dominant kernel actually moved: rocprofv3 PMC capture of THIS build (FETCH_SIZE x2 + WRITE_SIZE,
        # separate passes, MI355X_MICROARCH.md HBM section), else null + traffic_stale
        traffic, stale, note = (None, True, "the dominant stage is not gridding")
        if dom == "grid":
            traffic, stale, note = traffic_capture(workload_key(args), GRID_STAGE_KERNELS, units_per_launch)
        roofline = dict(bound="hbm", kernel=kname, achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=traffic, traffic_stale=stale, traffic_source=note,
                        bytes_per_launch=int(bytes_per_launch), units_per_launch=units_per_launch,
                        launch_ms=round(ms / n, 4), stage_share={k: round(v[0] / tot, 3) for k, v in stages.items()},
                        measured="hipEvents round the stage's launches on the plan's stream (one stream: every kernel runs alone)")

    one_coil = None
    if (rank == 0 and world == 1 and nc == 8 and golden and not args.half and args.kb == "fast" and NPE == 402 and not args.no_one_coil):
        one_coil = one_coil_field(args, lib, torch, local_rank, undersamp)

    result = None
    if rank == 0:
        cpu = None
        if args.cpu_slices != 0 and world == 1:               # the CPU baseline is reported at N = 1 only
            n_cpu = args.cpu_slices
            if n_cpu < 0:                                     # one probe slice sizes the bounded sample
                probe, _ = cpu_baseline(nc, 1, undersamp)
                n_cpu = max(2, min(32, int(round(12.0 * probe["value"]))))
            cpu, _ = cpu_baseline(nc, n_cpu, undersamp)
            cpu["value"] = round(cpu["value"], 4)
        irt_cpu = irt_baseline(nc) if (not args.no_irt and world == 1 and args.cpu_slices != 0) else None
        if not args.no_check:
            # the timed path produced real images: spot-check the first and the last slice of this rank against the oracle
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle import pyoracle
            pyoracle.set_threads(max(1, (os.cpu_count() or 1) // world))      # torchrun exports OMP_NUM_THREADS=1
            err = 0.0
            for z in sorted({0, nz - 1} if world == 1 else {0}):
                lo = 2 * nc * NRO * NPE * z
                host = kspace[lo: lo + 2 * nc * NRO * NPE].float().cpu().numpy().view(np.complex64).reshape((nc, 1, NRO, NPE, 1), order="F")
                want, _ = pyoracle.recon(host, adjoint=1, golden=golden, data_undersamp=undersamp, prof_slide=NPE, skip_angles=(zfirst + z) * NPE)
                got = images[2 * NX * NX * z: 2 * NX * NX * (z + 1)].cpu().numpy().view(np.complex64)
                err = max(err, float(np.linalg.norm(got - want.reshape(-1, order="F")) / np.linalg.norm(want)))
            if not err <= 1e-5:
                raise SystemExit(f"bench output disagrees with the oracle: rel L2 {err:.3e}")
        else:
            err = None
        alg_gbps = ab["per_slice"] * value / world / 1e9      # per GPU, SURVEY 8(d)'s algorithmic bytes (not measured traffic)
        per_gpu = f"{nz} slices/GPU/step" if args.scaling == "weak" else f"{total_slices} slices/step in total ({nz} on rank 0)"
        result = {
            "metric": "2D slices/sec gridded (512^2 grid, 512x402 golden-angle) + achieved HBM GB/s",
            "value": round(value, 1), "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32" if not args.half else "f32 (k-space stored as f16)", "data": "synthetic",
            "config": {"workload": f"adjoint gridding recon: {per_gpu} x {nc} coils, 512 readout x {NPE} {'golden' if golden else 'linear'}-angle spokes "
                                   f"-> 512^2 oversampled grid -> 256^2 image (tron -a {'-G ' if golden else ''}-u {undersamp:.4f} -d {NPE})",
                       "coils": nc, "slices_per_gpu": nz, "kb_mode": args.kb,
                       "kb_mode_note": ("fast = tabulated Kaiser-Bessel weights (arc and centre kernels: quadratic pieces, the reference's exact support) + their own summation order: within 1e-5 of the reference arithmetic (measured on the "
                                        "line: parity_rel_l2_vs_oracle), NOT bit-identical; --kb exact is the bit-identical gather, ~2.7x slower"
                                        if args.kb == "fast" else "exact = the reference's expression tree and summation order, bit-identical interpolation"),
                       "parallelism": f"slices sharded over {world} GPU(s), one process each, no collective on the data path (gloo barrier only)"
                                      + (f"; rank 0 bound to the {len(bound_cpus)} CPUs of its GPU's NUMA node" if bound_cpus else ""),
                       **({"ranks_share_one_gpu": True} if share and ndev < world else {})},
            "algorithmic_gbps_per_gpu": round(alg_gbps, 1), "algorithmic_frac_of_peak": round(alg_gbps / HBM_PEAK_GBPS, 4),
            "copy_ceiling_guide_gbps": 6290.0,      # MI355X_MICROARCH.md: float4 copy, 79 % of the 8 TB/s spec (round 5's own torch copy probe read 5.0 TB/s and is gone)
            "coil_slices_per_s": round(value * nc, 1),
            # the same step() for >= --sustain seconds right after the timed steps (value / ms_per_step above are the driver's K steps)
            "sustained_slices_per_s": sustained["value"] if sustained else None, "sustained": sustained,
            # every step on NEW spoke angles (tron_plan_retarget inside the clock): VERDICT round 5, item 1
            "fresh_trajectory_slices_per_s": fresh["value"] if fresh else None, "fresh_trajectory": fresh,
            "one_coil": one_coil,
            "burn_in_attempts": json.loads(os.environ["TRON_BENCH_BURNT"]) if os.environ.get("TRON_BENCH_BURNT", "").startswith("[") else None,
            "parity_rel_l2_vs_oracle": err,
            # plan creation is outside the timed region (value = steady state of a plan made once per trajectory); the reference's one
            # published time clocks tron_init too (src/tron.cu:973-978), so the plan's cost and the rate of ONE cold job are stated here
            "plan_ms": {"total": round(plan_wall_s * 1e3, 1), "hip_runtime_and_code_objects": round(plan_times["runtime"] * 1e3, 1),
                        "tables": round(plan_times["tables"] * 1e3, 1), "run_tables_of_the_gridding_kernels": round(plan_times["run_tables"] * 1e3, 1),
                        "work_buffers": round(plan_times["work_buffers"] * 1e3, 1)},
            "cold_slices_per_s": round(nz * world / (plan_wall_s + first_step_s), 1) if first_step_s is not None else None,
            "cold_note": f"one job from nothing: tron_plan_create + the first step of {nz} slices (first launches of every kernel included), rank 0's clock",
            # cpu_baseline = the comparator north_star names (contrib/irt on the host cores); oracle_baseline = the CPU oracle,
            # i.e. the reference's own point-driven algorithm without a GPU (kind "port")
            "roofline": roofline, "cpu_baseline": irt_cpu if irt_cpu is not None else cpu, "oracle_baseline": cpu if irt_cpu is not None else None,
        }
    plan.close()
    group.close()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
