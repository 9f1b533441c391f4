"""z-slab sharding of the structured mesh over ranks + interface-plane exchange.

The reference shards the spatial mesh over MPI ranks (parallel::distributed::Triangulation,
tests/tp_01.cc:80); every spatial cell loop is bracketed by a ghost update of src and a
compress(add) of dst inside MatrixFree::cell_loop (include/operators.h:1016-1017).  Here:

  * rank g owns cell layers [z0, z1) of the global mesh; its local DoF box has
    nz_local = p*(z1-z0)+1 planes; its TOP plane is the ghost copy of the upper
    neighbour's BOTTOM plane (owner = upper rank), exactly one shared plane per interface.
  * the fused kernel needs ONE exchange per space-time vmult (all temporal blocks packed
    together) instead of the reference's 2*n_blocks: after the local cell sweep both copies of
    an interface plane hold partial sums; `exchange_add` sends each partial to the other side
    and adds, which leaves owner and ghost consistent (compress(add) + update_ghost_values in
    one step).  Ring neighbours only -> each message rides one xGMI link.

Only torch.distributed point-to-point is used (backend "nccl" = RCCL on the GPUs, "gloo" in
the CPU tests).  The plane buffers are torch tensors; packing/unpacking on the GPU goes through
the C-ABI (stfem_plane_pack / stfem_plane_unpack) via the callables handed in.
"""
from dataclasses import dataclass


@dataclass
class Slab:
    rank: int
    world: int
    z0: int  # first owned cell layer
    z1: int  # one past the last
    global_ncell: tuple

    @property
    def ncell(self):
        return (self.global_ncell[0], self.global_ncell[1], self.z1 - self.z0)

    @property
    def has_lower(self):
        return self.rank > 0

    @property
    def has_upper(self):
        return self.rank < self.world - 1

    def dirichlet_mask(self, global_mask=63):
        """Partition interfaces carry no constraints (include/stfem.h: stfem_mesh_desc)."""
        m = global_mask
        if self.has_lower:
            m &= ~16
        if self.has_upper:
            m &= ~32
        return m

    def n_owned_planes(self, degree):
        """Planes this rank owns: all but the top ghost plane (the last rank owns its top)."""
        nz = degree * (self.z1 - self.z0) + 1
        return nz - 1 if self.has_upper else nz


def make_slab(global_ncell, rank, world):
    nzc = global_ncell[2]
    if world > nzc:
        raise ValueError("more ranks than cell layers")
    base, extra = divmod(nzc, world)
    z0 = rank * base + min(rank, extra)
    z1 = z0 + base + (1 if rank < extra else 0)
    return Slab(rank, world, z0, z1, tuple(global_ncell))


def exchange_add(slab, top_send, bottom_send, top_recv, bottom_recv, dist):
    """Ring neighbour exchange of the packed interface planes (all temporal blocks in one message).

    top_send/bottom_send: this rank's partial sums of its top / bottom plane (torch tensors);
    top_recv/bottom_recv: receive buffers for the neighbour's partials of the same planes.
    Returns the list of work handles; caller waits, then adds top_recv into its top plane and
    bottom_recv into its bottom plane."""
    ops = []
    if slab.has_upper:
        ops.append(dist.P2POp(dist.isend, top_send, slab.rank + 1))
        ops.append(dist.P2POp(dist.irecv, top_recv, slab.rank + 1))
    if slab.has_lower:
        ops.append(dist.P2POp(dist.isend, bottom_send, slab.rank - 1))
        ops.append(dist.P2POp(dist.irecv, bottom_recv, slab.rank - 1))
    return dist.batch_isend_irecv(ops) if ops else []


def sharded_vmult(slab, local_vmult, pack_plane, unpack_add_plane, bufs, dist):
    """One space-time vmult on a z-slab decomposition.

    local_vmult():            runs the cell sweep on this rank's slab (dst overwritten)
    pack_plane(iz, buf):      buf <- dst planes iz of all blocks
    unpack_add_plane(iz, buf): dst planes iz of all blocks += buf
    bufs: dict with tensors 'ts','bs','tr','br' (top/bottom send/recv), each n_blocks*nx*ny."""
    local_vmult()
    if slab.world == 1:
        return
    if slab.has_upper:
        pack_plane(-1, bufs["ts"])
    if slab.has_lower:
        pack_plane(0, bufs["bs"])
    for w in exchange_add(slab, bufs["ts"], bufs["bs"], bufs["tr"], bufs["br"], dist):
        w.wait()
    if slab.has_upper:
        unpack_add_plane(-1, bufs["tr"])
    if slab.has_lower:
        unpack_add_plane(0, bufs["br"])


class Communicator:
    """The RCCL communicator behind the C-ABI (include/stfem.h: stfem_comm_*): what a C++ caller uses
    instead of torch.distributed.  `bcast(bytes_or_None) -> bytes` carries the 128-byte id from rank 0
    to the others (MPI_Bcast on the deal.II side; torch.distributed / gloo in bench.py and the tests)."""

    @staticmethod
    def available():
        """RCCL can be bound in this process.  Ask on every rank and agree (all-reduce MIN) BEFORE any rank constructs a
        Communicator: the constructor is collective (broadcast of the id, ncclCommInitRank)."""
        from . import lib
        return bool(lib().stfem_comm_available())

    @property
    def rccl_nranks(self):
        return int(self._L.stfem_comm_rccl_count(self._h)) if getattr(self, "_h", None) else 0

    def __init__(self, rank, world, device, bcast):
        import ctypes as C
        from . import lib, _check
        self._L = lib()
        uid = C.create_string_buffer(128)
        if rank == 0:
            _check(self._L.stfem_comm_get_unique_id(uid), "stfem_comm_get_unique_id")
        raw = bcast(uid.raw if rank == 0 else None)
        assert len(raw) == 128
        h = C.c_void_p()
        rc = self._L.stfem_comm_create(C.create_string_buffer(raw, 128), rank, world, device, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"stfem_comm_create: status {rc}: {self._L.stfem_comm_last_error().decode()}")
        self._h, self.rank, self.world = h, rank, world

    def close(self):
        if getattr(self, "_h", None):
            self._L.stfem_comm_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: status {rc}: {self._L.stfem_comm_last_error().decode()}")

    def ghost_update(self, ctx, vec, lower, upper, stream=None):
        self._chk(self._L.stfem_ghost_update(ctx._h, self._h, vec._h, lower, upper, stream), "stfem_ghost_update")

    def halo_begin(self, ctx, vec, lower, upper, stream=None):
        self._chk(self._L.stfem_halo_begin(ctx._h, self._h, vec._h, lower, upper, stream), "stfem_halo_begin")

    def halo_begin_split(self, ctx_lo, v_lo, ctx_hi, v_hi, lower, upper, stream=None):
        self._chk(self._L.stfem_halo_begin_split(self._h, ctx_lo._h, v_lo._h, ctx_hi._h, v_hi._h, lower, upper, stream), "stfem_halo_begin_split")

    def halo_end(self, ctx, vec, stream=None):
        self._chk(self._L.stfem_halo_end(ctx._h, self._h, vec._h, stream), "stfem_halo_end")

    def dot(self, ctx, a, b, n_own):
        import ctypes as C
        out = C.c_double(0.0)
        self._chk(self._L.stfem_dot_global(ctx._h, self._h, a._h, b._h, n_own, C.byref(out), None), "stfem_dot_global")
        return out.value


def neighbours(slab):
    """(lower_rank, upper_rank) of a z-slab, -1 at the ends of the mesh."""
    return (slab.rank - 1 if slab.has_lower else -1, slab.rank + 1 if slab.has_upper else -1)


def sharded_vmult_abi(slab, comm, ctx, local_vmult, dst, stream=None):
    """sharded_vmult with the exchange inside the C-ABI (RCCL): sweep, then one packed exchange."""
    local_vmult()
    if slab.world == 1:
        return
    lo, up = neighbours(slab)
    comm.halo_begin(ctx, dst, lo, up, stream)
    comm.halo_end(ctx, dst, stream)


class OverlappedSlabOperator:
    """The space-time vmult of one z-slab with the interface-plane exchange hidden behind the interior cells (deal.II's cell_loop
    overlaps the ghost exchange with the cells that do not need it, include/operators.h:1016-1017): the bottom and the top CELL LAYER of
    the slab are swept first, each on a one-layer context into a small vector of its own; their outer planes - the slab's two interface
    planes, complete - start travelling (stfem_halo_begin_split, RCCL on the communicator's stream); the interior layers are swept
    meanwhile straight into the destination vector; the two thin results are moved in (stfem_planes_move: the plane each shares with
    the interior is added); stfem_halo_end adds what arrived.  Needs at least three cell layers.

    make_ctx(z0, z1, dirichlet_mask) -> MatrixFreeOperator of the slab's local cell layers [z0, z1) (same x / y extent, degree and
    Number as `ctx`, the context of the whole slab that `src` / `dst` belong to); make_system(ctx) -> the SystemMatrix on it."""

    def __init__(self, stfem, ctx, make_ctx, make_system, src, dst, slab_mask):
        self.stfem, self.ctx, self.src, self.dst = stfem, ctx, src, dst
        nz = ctx.ncell[2]
        if nz < 3:
            raise ValueError("OverlappedSlabOperator: at least three cell layers per slab")
        p = ctx.degree
        self.p, self.nz = p, nz
        plane_bytes = (p * ctx.ncell[0] + 1) * (p * ctx.ncell[1] + 1) * (8 if ctx.number == "double" else 4)
        nb_in, nb_out = src.n_blocks, dst.n_blocks
        lo_open, hi_open = slab_mask & ~32, slab_mask & ~16  # the bottom layer's upper face / the top layer's lower face are interior
        self.bot = make_ctx(0, 1, lo_open)
        self.top = make_ctx(nz - 1, nz, hi_open)
        self.mid = make_ctx(1, nz - 1, slab_mask & ~48)
        self.A_bot, self.A_top, self.A_mid = make_system(self.bot), make_system(self.top), make_system(self.mid)
        view = lambda vec, c, z0, n: stfem.BlockVector(c, device_ptrs=[vec.block_ptr(b) + plane_bytes * p * z0 for b in range(n)])  # noqa: E731
        self.src_bot, self.src_top, self.src_mid = view(src, self.bot, 0, nb_in), view(src, self.top, nz - 1, nb_in), view(src, self.mid, 1, nb_in)
        self.dst_mid = view(dst, self.mid, 1, nb_out)
        self.dst_bot, self.dst_top = stfem.BlockVector(self.bot, nb_out), stfem.BlockVector(self.top, nb_out)

    def vmult(self, comm=None, lower=-1, upper=-1, stream=None, transpose=False):
        L = self.stfem.lib()
        p, nz = self.p, self.nz
        ap = (lambda A, d, s_: A.Tvmult(d, s_, stream)) if transpose else (lambda A, d, s_: A.vmult(d, s_, stream))
        ap(self.A_bot, self.dst_bot, self.src_bot)
        ap(self.A_top, self.dst_top, self.src_top)
        if comm is not None:
            comm.halo_begin_split(self.bot, self.dst_bot, self.top, self.dst_top, lower, upper, stream)
        ap(self.A_mid, self.dst_mid, self.src_mid)
        rc = L.stfem_planes_move(self.bot._h, self.dst_bot._h, 0, self.ctx._h, self.dst._h, 0, p + 1, 2, stream)
        assert rc == 0, rc
        rc = L.stfem_planes_move(self.top._h, self.dst_top._h, 0, self.ctx._h, self.dst._h, p * (nz - 1), p + 1, 1, stream)
        assert rc == 0, rc
        if comm is not None:
            comm.halo_end(self.ctx, self.dst, stream)
