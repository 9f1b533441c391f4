#!/usr/bin/env python3
"""Headline benchmark: space-time DoF/s of one matrix-free vmult (3D heat, Q4 x cG(2)).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one SystemMatrix::vmult (reference include/operators.h:536-559) of the all-at-once
space-time system  dst = (Alpha (x) K + Beta (x) M) src  on synthetic data resident in HBM.
N = 1: BASELINE.json configs[1] (72^3 cells, Q4, cG(2) -> 2 blocks, 48 275 138 space-time DoFs).
N > 1: BASELINE configs[2] - the fixed 144^3-cell mesh with vertices perturbed by 0.15 h (general-geometry
path, 383 M space-time DoFs) cut into N z-slabs: STRONG scaling, as north_star asks (>= 6x at 8 GPUs); one
packed interface-plane exchange per vmult over RCCL send/recv, no other collective on the data path.  The
line also carries the time of the SAME mesh on one GPU (rank 0 runs it after the timed region), because the
default N = 1 line is configs[1] and not the one-GPU point of this curve.  --scaling weak: every rank owns a
72 x 72 x 72-cell slab of a 72 x 72 x 72N mesh instead.  The JSON says which ran.

Prints ONE JSON line (rank 0) with the driver's contract plus "roofline" and "cpu_baseline".
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def cpu_baseline(stfem, degree, r, cells, threads):
    """CPU restatement of the reference's algorithm IN ITS OWN STRUCTURE (oracle/stfem_cpu_baseline.c,
    kind "port": 2 n_blocks spatial cell loops + axpys per vmult, include/operators.h:536-559, run the
    way deal.II's MatrixFree runs them: Cartesian-compressed geometry, SIMD across cells, OpenMP over
    all granted cores) on the SAME mesh as the GPU run, a bounded number of vmults."""
    import numpy as np
    from oracle import oracle
    nc = (cells,) * 3
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, r, 1.0 / 144, 1)
    nb = Alpha.shape[0]
    cb = oracle.CpuBaseline(degree, nc, threads=threads)
    rng = np.random.default_rng(1234)
    X = rng.uniform(-1, 1, (nb, cb.n_dofs))
    Y = np.empty_like(X)
    cb.st_vmult(Alpha, Beta, X, Y)  # warm-up (first touch of Y and the scratch vector)
    reps, t0 = 0, time.perf_counter()
    while True:
        cb.st_vmult(Alpha, Beta, X, Y)
        reps += 1
        el = time.perf_counter() - t0
        if el > 12.0 or reps >= 40:
            break
    dofs = nb * cb.n_dofs
    return {"value": dofs * reps / el, "unit": "space-time DoF/s", "cores": threads,
            "kind": "port", "algorithmic_GBps": 16.0 * dofs * reps / el / 1e9,
            "sample": f"{reps} vmults of Q{degree} x cG({r}) on the full {cells}^3-cell mesh "
                      f"({dofs} space-time DoFs), oracle/stfem_cpu_baseline.c: the reference's structure "
                      f"(2 n_blocks cell loops + axpys per vmult) with Cartesian-compressed geometry, SIMD "
                      f"across cells and OpenMP on {threads} cores; deal.II is unavailable, so not the reference binary"}


SWEEP_SOURCES = ["stfem_pencil.hip", "stfem_tile.hip", "stfem_general.hip", "stfem_core.h", "stfem_device.h", "stfem_kernels.h",
                 "stfem_kernels_decl.h", "stfem_capi.hip"]


def sweep_source_hash():
    """sha256 over the sources that decide what a vmult moves (kernels + planner): a committed PMC profile is only
    reported as `roofline.traffic` while these are the files it was taken with (tools/summarize_pmc.py records it)."""
    import hashlib
    h = hashlib.sha256()
    for name in SWEEP_SOURCES:
        path = os.path.join(ROOT, "dealii-stfem_amd", "csrc", name)
        if os.path.exists(path):
            with open(path, "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def measured_traffic(kernel_name, general):
    """HBM bytes per vmult from the committed rocprofv3 PMC passes of THIS command
    (profiles/latest/traffic.json for the Cartesian cfg-1 run, traffic_general.json for the perturbed
    72^3 mesh; written by tools/profile.sh): WRITE_SIZE + 2 x FETCH_SIZE, the factor 2 being the gfx950
    FETCH_SIZE correction of MI355X_MICROARCH.md.  None if the profile does not belong to the kernel
    variant that ran."""
    path = os.path.join(ROOT, "profiles", "latest", "traffic_general.json" if general else "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if not kernel_name.startswith(t.get("kernel", "?")) or "f32" in kernel_name:
            return None, None
        if t.get("sources_sha256") != sweep_source_hash():
            return None, os.path.relpath(path, ROOT) + " is STALE (kernel sources changed since the PMC passes): not reported"
        return 1024.0 * (2.0 * t["fetch_kb_per_vmult"] + t["write_kb_per_vmult"]), os.path.relpath(path, ROOT)
    except (OSError, KeyError, ValueError):
        return None, None


def bench_stokes(args, stfem, torch, dev):
    """BASELINE configs[4] on one GPU: SystemMatrixStokes::vmult (operators.h:825-867), FE_Q(2)^3 x FE_Q(1) x cG(time_degree), unit cube,
    homogeneous Dirichlet velocity.  A step = one space-time vmult; inputs resident in HBM; HIP events on the launch stream."""
    n = 64 if args.cells == 72 else args.cells  # (72 is the default of the heat line)
    r = 1 if args.time_degree == 2 else args.time_degree
    op = stfem.StokesMatrixFreeOperator((n, n, n), viscosity=1.0, device=dev.index or 0)
    Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(stfem.CGP, r, 1.0 / 64, 1)
    nt = r
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    sizes = {0: 3 * op.n_velocity, 1: op.n_pressure}
    keep, src, dst = [], [None] * (2 * nt), [None] * (2 * nt)
    for d in range(nt):
        for v in range(2):
            j = stfem.stokes_block_index(nt, 0, v, d)
            a = torch.rand(sizes[v], dtype=torch.float64, device=dev, generator=gen) * 2 - 1
            b = torch.zeros(sizes[v], dtype=torch.float64, device=dev)
            keep += [a, b]
            src[j], dst[j] = a.data_ptr(), b.data_ptr()
    for _ in range(args.warmup):
        op.st_vmult(Alpha, Beta, 1, nt, dst, src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        op.st_vmult(Alpha, Beta, 1, nt, dst, src)
    e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kms = e0.elapsed_time(e1) / args.steps
    dofs = nt * (3 * op.n_velocity + op.n_pressure)
    alg_bytes = 16.0 * dofs
    achieved = alg_bytes / (kms * 1e-3) / 1e9
    out = {
        "metric": "space-time DoF/s per vmult (Stokes block operator, Q2/Q1 x cG(%d)); achieved HBM GB/s" % r,
        "value": dofs * args.steps / elapsed, "unit": "space-time DoF/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"Stokes space-time block operator, FE_Q(2)^3 x FE_Q(1) x cG({r}), {n}x{n}x{n} cells, {dofs} space-time DoFs "
                               "= BASELINE configs[4] on one GPU",
                   "n_blocks": 2 * nt, "cells_per_gpu": n ** 3,
                   "kernel": "st_sweep_pencil (velocity components) + stokes_div_kernel beside it + stokes_grad_kernel"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "traffic_source": "profiles/r3/stokes (counter passes of tools/stokes_bench.py, per kernel)",
                     "kernel_ms": kms, "algorithmic_bytes_per_launch": alg_bytes},
        "cpu_baseline": None,
    }
    print(json.dumps(out), flush=True)
    del keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cells", type=int, default=72, help="cells per direction per GPU")
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--time-degree", type=int, default=2)
    ap.add_argument("--timesteps-at-once", type=int, default=1,
                    help="time steps per slab (the reference's n_timesteps_at_once, fe_time.h:373-402): cG(r) x s steps = r s blocks")
    ap.add_argument("--distort", type=float, default=None,
                    help="interior-vertex jitter in units of h (0.15 = BASELINE configs[2] mesh); 0 = Cartesian. "
                         "Default: 0 at --gpus 1 (configs[1]), 0.15 at --gpus N > 1 (the configs[2] mesh)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="N > 1 default: strong = the fixed --strong-cells^3 mesh (144 = configs[2]) cut into N z-slabs; "
                         "weak = 72^3 cells per GPU.  N = 1 default: the configs[1] mesh (reported as weak: its own point)")
    ap.add_argument("--strong-cells", type=int, default=144)
    ap.add_argument("--number", choices=["double", "float"], default="double",
                    help="operator Number type: double (headline) or float (the reference's multigrid-level precision)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: rehearsal of the N > 1 path on ONE GPU (all ranks share cuda:0, the interface planes "
                         "are staged through host memory); timings of such a run mean nothing")
    ap.add_argument("--exchange", choices=["abi", "torch"], default=None,
                    help="N > 1: who moves the interface planes: abi = RCCL behind the C-ABI (stfem_halo_begin/end, "
                         "default with --backend nccl), torch = torch.distributed P2P around stfem_plane_pack/unpack "
                         "(default with --backend gloo; also taken if the RCCL communicator cannot be created)")
    ap.add_argument("--check", action="store_true",
                    help="N > 1: compare every rank's slab of the sharded vmult with a single-domain vmult of the "
                         "whole mesh computed on the same GPU (small meshes only)")
    ap.add_argument("--config", type=int, default=1, choices=[1, 3, 4],
                    help="1: BASELINE configs[1] / [2] (heat, the contract line); 3: configs[3] = wave equation, Q3 x dG(2), "
                         "80^3 cells on [-1,1]^3 per GPU, Coefficient(1,9,16) per cell; 4: configs[4] = Stokes block operator, Q2/Q1 x cG(1), "
                         "64^3 cells (--cells) on one GPU (extra lines, not the contract metric's workload)")
    ap.add_argument("--overlap", action="store_true",
                    help="sweep the slab's two interface cell layers first and its interior while their planes travel "
                         "(dealii-stfem_amd/distributed.py: OverlappedSlabOperator; with --exchange abi for N > 1; at N = 1 it measures what "
                         "the three-part sweep costs over one sweep)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-one-gpu-point", action="store_true",
                    help="N > 1, strong scaling: do not run the whole mesh on rank 0's GPU after the timed region")
    ap.add_argument("--cpu-cells", type=int, default=0, help="cells per direction of the CPU baseline mesh (0: as the GPU run)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    stfem = importlib.import_module("dealii-stfem_amd")
    stfem.lib()
    from importlib import import_module
    dmod = import_module("dealii-stfem_amd.distributed")
    if args.config == 4:
        if world != 1:
            raise SystemExit("--config 4 is a one-GPU line (the Stokes operator has no partitioned form yet)")
        return bench_stokes(args, stfem, torch, dev)

    if args.distort is None:
        args.distort = 0.0 if world == 1 else 0.15
    if args.scaling is None:
        args.scaling = "weak" if world == 1 else "strong"
    if args.config == 3:
        if world != 1:
            raise SystemExit("--config 3 is a one-GPU line")
        args.degree, args.time_degree, args.cells, args.distort, args.no_cpu_baseline = 3, 2, 80, 0.0, True
    p, r, n = args.degree, args.time_degree, args.cells
    if args.scaling == "strong":
        n = args.strong_cells
        global_nc = (n, n, n)
    else:
        global_nc = (n, n, n * world)
    zext = global_nc[2] / float(n)  # the domain is [0,1]^2 x [0, zext]: cubic cells
    slab = dmod.make_slab(global_nc, rank, world)
    # tests/tp_01.cc:106-109 with 9 subdivisions, refinement 3: tau = 1/144
    tau = 1.0 / 144
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, r, tau, args.timesteps_at_once)
    nb = Alpha.shape[0]
    if args.distort:
        verts = stfem.mesh_vertices(global_nc, (0, 0, 0), (1, 1, zext), args.distort, 5489,
                                    z_range=(slab.z0, slab.z1))
        ctx = stfem.MatrixFreeOperator(p, slab.ncell, vertices=verts, number=args.number,
                                       dirichlet_mask=slab.dirichlet_mask(63), device=local_rank)
    else:
        ctx = stfem.MatrixFreeOperator(p, slab.ncell, lower=(0, 0, float(slab.z0) / n),
                                       upper=(1, 1, float(slab.z1) / n), number=args.number,
                                       dirichlet_mask=slab.dirichlet_mask(63), device=local_rank)
    if args.config == 3:  # tests/tp_01.cc:141-150 (coefficient on K), fe_time.h:157-305 (wave matrices), SURVEY 8d: tau = 1/64
        lo, up = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)
        ctx = stfem.MatrixFreeOperator(p, slab.ncell, lower=lo, upper=up, number=args.number, device=local_rank)
        coef = stfem.coefficient_per_cell(slab.ncell, stfem.mesh_vertices(slab.ncell, lo, up), 1.0, 9.0, 16.0, 0.0, (5, 5, 5), lo, up)
        ctx.evaluate_coefficient(coef, which=1)
        Alpha, Beta, _, _, _ = stfem.get_fe_time_weights_wave(stfem.DG, r, 1.0 / 64, 1)
        nb = Alpha.shape[0]
    A = stfem.SystemMatrix(ctx, Alpha, Beta)
    ndofs = ctx.n_dofs
    nx = p * n + 1
    plane = nx * nx

    # synthetic data, resident in HBM before the timed region (torch = device-memory plumbing)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    tdt = torch.float64 if args.number == "double" else torch.float32
    esz = 8 if args.number == "double" else 4
    src_t = torch.rand((nb, ndofs), dtype=tdt, device=dev, generator=gen) * 2 - 1
    dst_t = torch.zeros((nb, ndofs), dtype=tdt, device=dev)
    src = stfem.BlockVector(ctx, device_ptrs=[src_t[b].data_ptr() for b in range(nb)])
    dst = stfem.BlockVector(ctx, device_ptrs=[dst_t[b].data_ptr() for b in range(nb)])
    bufs = {k: torch.zeros(nb * plane, dtype=tdt, device=dev) for k in ("ts", "bs", "tr", "br")}
    L = stfem.lib()
    nz_local = p * (slab.z1 - slab.z0) + 1

    def stream():
        return torch.cuda.current_stream().cuda_stream

    def pack(iz, buf):
        rc = L.stfem_plane_pack(ctx._h, dst._h, iz % nz_local, buf.data_ptr(), stream())
        assert rc == 0, rc

    def unpack_add(iz, buf):
        rc = L.stfem_plane_unpack(ctx._h, dst._h, iz % nz_local, buf.data_ptr(), 1, stream())
        assert rc == 0, rc

    xdev = dev if args.backend == "nccl" else torch.device("cpu")  # where exchanged buffers live
    if world > 1:  # make the src ghost plane consistent with its owner once (update_ghost_values)
        if slab.has_lower:
            dist.send(src_t[:, :plane].contiguous().to(xdev), rank - 1)
        if slab.has_upper:
            g = torch.empty((nb, plane), dtype=tdt, device=xdev)
            dist.recv(g, rank + 1)
            src_t[:, -plane:] = g.to(dev)

    class StagedDist:
        """gloo rehearsal: the packed planes go through host memory around the same exchange calls"""
        def __init__(self):
            self.P2POp, self.isend, self.irecv = dist.P2POp, dist.isend, dist.irecv  # instance attributes: not bound
            self.host = {k: torch.zeros(nb * plane, dtype=tdt) for k in bufs}

        def batch_isend_irecv(self, ops):
            torch.cuda.synchronize()
            staged = []
            for op in ops:
                key = next(k for k, v in bufs.items() if v.data_ptr() == op.tensor.data_ptr())
                if op.op == dist.isend:
                    self.host[key].copy_(op.tensor)
                staged.append(dist.P2POp(op.op, self.host[key], op.peer))
            works = dist.batch_isend_irecv(staged)
            for w in works:
                w.wait()
            for op in ops:
                if op.op == dist.irecv:
                    key = next(k for k, v in bufs.items() if v.data_ptr() == op.tensor.data_ptr())
                    op.tensor.copy_(self.host[key])
            return []

    xdist = dist if args.backend == "nccl" else StagedDist()
    if args.exchange is None:
        args.exchange = "abi" if args.backend == "nccl" else "torch"
    comm = None
    if world > 1 and args.exchange == "abi":
        def bcast(raw):
            box = [raw]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        # agree that every rank can bind RCCL BEFORE any rank enters the collective construction (a rank that
        # fails early would leave the others blocked in the broadcast / ncclCommInitRank)
        ok = torch.tensor([1 if dmod.Communicator.available() else 0], device=xdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            try:
                comm = dmod.Communicator(rank, world, local_rank, bcast)
            except Exception as e:  # noqa: BLE001  (reported below; every rank must take the same path)
                print(f"[bench] rank {rank}: RCCL communicator behind the C-ABI not available ({e})", file=sys.stderr, flush=True)
            ok = torch.tensor([1 if comm is not None else 0], device=xdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.close()
            comm, args.exchange = None, "torch"
    lo_rank, up_rank = dmod.neighbours(slab)
    overlapped = None
    if args.overlap:
        if world > 1 and comm is None:
            raise SystemExit("--overlap needs the RCCL exchange behind the C-ABI (--exchange abi)")
        if args.config != 1:
            raise SystemExit("--overlap: heat configurations only")
        smask = slab.dirichlet_mask(63)

        def make_ctx(z0, z1, m):
            if args.distort:
                nvp = (global_nc[0] + 1) * (global_nc[1] + 1) * 3
                import numpy as np
                v = np.asarray(verts).reshape(-1, nvp)[z0:z1 + 1].reshape(-1)
                return stfem.MatrixFreeOperator(p, (slab.ncell[0], slab.ncell[1], z1 - z0), vertices=v, number=args.number, dirichlet_mask=m,
                                                device=local_rank)
            return stfem.MatrixFreeOperator(p, (slab.ncell[0], slab.ncell[1], z1 - z0), lower=(0, 0, float(slab.z0 + z0) / n),
                                            upper=(1, 1, float(slab.z0 + z1) / n), number=args.number, dirichlet_mask=m, device=local_rank)

        overlapped = dmod.OverlappedSlabOperator(stfem, ctx, make_ctx, lambda c: stfem.SystemMatrix(c, Alpha, Beta), src, dst, smask)

    kernel_ms, exchange_ms = [], []

    def step(record=False):
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if overlapped is not None:  # interface layers, exchange in flight behind the interior layers, assembly: one timed piece
            overlapped.vmult(comm if world > 1 else None, lo_rank, up_rank, stream())
            if record:
                e1.record()
                kernel_ms.append((e0, e1))
                if world > 1:
                    exchange_ms.append((e1, e1))
            return
        A.vmult(dst, src, stream=stream())
        if record:
            e1.record()
            kernel_ms.append((e0, e1))
        if world > 1:
            if comm is not None:
                comm.halo_begin(ctx, dst, lo_rank, up_rank, stream())
                comm.halo_end(ctx, dst, stream())
            else:
                dmod.sharded_vmult(slab, lambda: None, pack, unpack_add, bufs, xdist)
            if record:  # pack + RCCL send/recv + unpack-add of the interface planes
                e2 = torch.cuda.Event(enable_timing=True)
                e2.record()
                exchange_ms.append((e1, e2))

    if args.check and world > 1:
        import numpy as np
        gctx = (stfem.MatrixFreeOperator(p, global_nc, vertices=stfem.mesh_vertices(global_nc, (0, 0, 0), (1, 1, zext),
                                                                                      args.distort, 5489), number=args.number)
                if args.distort else
                stfem.MatrixFreeOperator(p, global_nc, lower=(0, 0, 0), upper=(1, 1, zext), number=args.number))
        Xg = np.random.default_rng(99).uniform(-1, 1, (nb, gctx.n_dofs))
        lo, hi = p * slab.z0 * plane, (p * slab.z1 + 1) * plane
        src_t.copy_(torch.from_numpy(Xg[:, lo:hi]).to(tdt))
        step()
        torch.cuda.synchronize()
        gA = stfem.SystemMatrix(gctx, Alpha, Beta)
        gdst = gA.initialize_dof_vector()
        gA.vmult(gdst, stfem.BlockVector(gctx, nb).upload(Xg))
        ref = gdst.download()[:, lo:hi]
        got = dst_t.double().cpu().numpy()
        err = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        print(f"[check] rank {rank}: slab vs single-domain vmult rel-L2 = {err:.3e}", file=sys.stderr, flush=True)
        assert err < (1e-12 if args.number == "double" else 2e-5), err
        del gctx, gA, gdst
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # units: owned space-time DoFs of all ranks (interface planes counted once)
    own = slab.n_owned_planes(p) * plane * nb
    if world > 1:
        t = torch.tensor([own], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        total_dofs = int(t.item())
    else:
        total_dofs = own
    ms_per_step = 1e3 * elapsed / args.steps
    kms = sum(a.elapsed_time(b) for a, b in kernel_ms) / len(kernel_ms)
    xms = sum(a.elapsed_time(b) for a, b in exchange_ms) / len(exchange_ms) if exchange_ms else 0.0

    # strong scaling: the one-GPU point of THIS curve, measured here (the default N = 1 line is configs[1], another mesh).
    # Rank 0 runs the whole mesh on its GPU after the timed region; the other ranks wait at the final barrier.
    one_gpu_ms = None
    if world > 1 and args.scaling == "strong" and rank == 0 and not args.no_one_gpu_point and args.backend == "nccl":
        try:
            del dst, src
            del dst_t, src_t
            torch.cuda.empty_cache()
            if args.distort:
                gctx = stfem.MatrixFreeOperator(p, global_nc, vertices=stfem.mesh_vertices(global_nc, (0, 0, 0), (1, 1, zext), args.distort, 5489),
                                                number=args.number, device=local_rank)
            else:
                gctx = stfem.MatrixFreeOperator(p, global_nc, lower=(0, 0, 0), upper=(1, 1, zext), number=args.number, device=local_rank)
            gA = stfem.SystemMatrix(gctx, Alpha, Beta)
            gs = torch.rand((nb, gctx.n_dofs), dtype=tdt, device=dev, generator=gen) * 2 - 1
            gd = torch.zeros((nb, gctx.n_dofs), dtype=tdt, device=dev)
            gsrc = stfem.BlockVector(gctx, device_ptrs=[gs[b].data_ptr() for b in range(nb)])
            gdst = stfem.BlockVector(gctx, device_ptrs=[gd[b].data_ptr() for b in range(nb)])
            for _ in range(3):
                gA.vmult(gdst, gsrc, stream=stream())
            torch.cuda.synchronize()
            reps = max(5, min(args.steps, 20))
            t1 = time.perf_counter()
            for _ in range(reps):
                gA.vmult(gdst, gsrc, stream=stream())
            torch.cuda.synchronize()
            one_gpu_ms = 1e3 * (time.perf_counter() - t1) / reps
            del gsrc, gdst, gs, gd, gA, gctx
        except Exception as e:  # noqa: BLE001
            print(f"[bench] one-GPU point of the strong-scaling curve not measured: {e}", file=sys.stderr, flush=True)

    if rank == 0:
        # 16 B per space-time DoF per vmult (8 B src read + 8 B dst write)
        alg_bytes = 2.0 * esz * nb * ndofs  # fp64: 16 B per DoF; fp32: 8 B
        achieved = alg_bytes / (kms * 1e-3) / 1e9
        traffic, traffic_file = (measured_traffic(ctx.last_kernel_name, bool(args.distort))
                                 if (world, n, p, r, args.number, args.timesteps_at_once) == (1, 72, 4, 2, "double", 1) and args.distort in (0.0, 0.15)
                                 else (None, None))
        out = {
            "metric": "space-time DoF/s per vmult (3D heat, Q4 space x cG(2) time); achieved HBM GB/s" if args.config == 1 else
                      "space-time DoF/s per vmult (3D wave, Q3 space x dG(2) time); achieved HBM GB/s",
            "value": total_dofs * args.steps / elapsed,
            "unit": "space-time DoF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64" if args.number == "double" else "f32", "data": "synthetic",
            "config": {"workload": (f"3D wave, Q{p} x dG({r}), 80x80x80 cells on [-1,1]^3, Coefficient(1,9,16) per cell, {total_dofs} space-time DoFs "
                                    "= BASELINE configs[3] on one GPU") if args.config == 3 else
                                   (f"3D heat, Q{p} x cG({r})" + (f" x {args.timesteps_at_once} time steps at once" if args.timesteps_at_once > 1 else "")
                                    + f", {global_nc[0]}x{global_nc[1]}x{global_nc[2]} cells "
                                    + (f"perturbed ({args.distort} h vertex jitter)" if args.distort else "Cartesian")
                                    + f" slab mesh, {total_dofs} space-time DoFs"
                                    + (" = BASELINE configs[1]" if (world, n, p, r, args.distort, args.timesteps_at_once) == (1, 72, 4, 2, 0.0, 1) else "")
                                    + (" = the BASELINE configs[2] mesh" if (global_nc, p, r, args.distort) == ((144, 144, 144), 4, 2, 0.15) else "")),
                       "n_blocks": nb, "cells_per_gpu": ctx.n_cells,
                       "partition": f"z-slabs x{world}", "kernel": ctx.last_kernel_name,
                       # rank 0, per step: the local cell sweep and the packed interface-plane exchange
                       # (weak scaling loses exactly the second; no other communication on the path)
                       "local_sweep_ms": kms, "exchange_ms": xms,
                       "overlap": ("interface cell layers first, exchange behind the interior sweep (OverlappedSlabOperator): local_sweep_ms is the "
                                   "whole step") if args.overlap else None,
                       "exchange": None if world == 1 else
                       ("RCCL behind the C-ABI (stfem_halo_begin/end)" if comm is not None
                        else f"torch.distributed ({args.backend}) P2P around stfem_plane_pack/unpack"),
                       # the default N = 1 line is configs[1] (Cartesian mesh, fast-diagonalisation kernel); N > 1 defaults to the
                       # configs[2] mesh type (general-geometry kernel, ~4x the time per cell): the one-GPU point of THIS curve is
                       # `bench.py --gpus 1 --distort 0.15` (weak) or `--gpus 1 --scaling strong --distort 0.15` (strong)
                       "one_gpu_point_of_this_curve": None if world == 1 else
                       ("bench.py --gpus 1 --distort %g" % args.distort + (" --scaling strong" if args.scaling == "strong" else "")),
                       # strong scaling: the same mesh on ONE GPU (rank 0, same process, wall clock over a few vmults after the
                       # timed region) and the speed-up of this N-GPU line over it
                       "one_gpu_ms_per_step_same_mesh": one_gpu_ms,
                       "speedup_vs_one_gpu_same_mesh": (one_gpu_ms / ms_per_step) if one_gpu_ms else None,
                       # what RCCL itself reports for the communicator behind the C-ABI (ncclCommCount)
                       "rccl_nranks": comm.rccl_nranks if comm is not None else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_source": (f"{traffic_file}: rocprofv3 --pmc passes of this command with these kernel sources (sha256 checked) "
                                            "(FETCH_SIZE x 2 + WRITE_SIZE), committed, not measured in this run")
                         if traffic is not None else traffic_file,
                         "kernel_ms": kms, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(stfem, p, r, args.cpu_cells or args.cells,
                                               max(1, min(16, len(os.sched_getaffinity(0)))))  # a 1-GPU box grants 16 cores
        print(json.dumps(out))
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
