"""Run-to-run reproducibility (collected last: these are not oracle-parity tests and must not sit in front of any).

No kernel on the operator path sums in an order that depends on timing: the pencil sweep accumulates in registers and
wave-private LDS, the tile sweep (general geometry; round 2 also the 4-8 block systems) gives every slab DoF a fixed
order of contributions since round 3 (owner store, same-wave x neighbour, then the wave below: csrc/stfem_tile.hip),
the Vanka apply and the transfers use colour launches / gathers.  So repeated applications agree BITWISE, in fp64 and
fp32, and the V-cycle recorded into a hipGraph replays to the bits of the plain launches.
(Round 2's driver run failed exactly here: the tile kernel summed shared DoFs with LDS atomics from several waves.)"""
import importlib
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dealii-stfem_amd", "host")


@pytest.mark.parametrize("number", ["double", "float"])
@pytest.mark.parametrize("p,nc,ttype,k,nsteps,distort", [
    (1, (24, 24, 24), 1, 1, 2, 0.0),   # dG(1) x 2 steps = 4 blocks: the failing level operator of round 2
    (2, (20, 16, 12), 0, 1, 4, 0.0),   # configs[0]: Q2 x cG(1), four steps at once
    (4, (12, 12, 12), 0, 2, 2, 0.0),   # Q4 x cG(2), two steps: 4 blocks
    (4, (12, 10, 8), 0, 2, 1, 0.15),   # configs[2] type: general geometry
    (3, (12, 12, 12), 1, 2, 2, 0.1),   # general geometry, 6 blocks
    (4, (18, 16, 16), 0, 2, 1, 0.0),   # configs[1] type: pencil sweep
])
def test_vmult_bitwise_reproducible(p, nc, ttype, k, nsteps, distort, number):
    stfem = importlib.import_module("dealii-stfem_amd")
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(ttype, k, 1.0 / 64, nsteps)
    nb = Alpha.shape[0]
    verts = stfem.mesh_vertices(nc, distort=distort, seed=5) if distort else stfem.mesh_vertices(nc)
    ctx = stfem.MatrixFreeOperator(p, nc, vertices=verts, number=number)
    A = stfem.SystemMatrix(ctx, Alpha, Beta)
    X = np.random.default_rng(3).uniform(-1, 1, (nb, ctx.n_dofs))
    src = stfem.BlockVector(ctx, nb).upload(X)
    dst = stfem.BlockVector(ctx, nb)
    A.vmult(dst, src)
    first = dst.download()
    assert np.isfinite(first).all() and np.abs(first).max() > 0
    for rep in range(12):
        (A.Tvmult if rep == 5 else A.vmult)(dst, src)  # something else in between now and then
        if rep == 5:
            continue
        assert np.array_equal(dst.download(), first), (ctx.last_kernel_name, rep)


@pytest.mark.parametrize("number", ["double", "float"])
@pytest.mark.parametrize("ttype,k,n,nsteps,p,ctype,pmg,distort", [
    (1, 1, 4, 2, 1, "space_or_time", False, 0.0),     # the round-2 failure: dG(1), 2 steps, h h k t
    (0, 2, 2, 2, 3, "space_and_time", True, 0.0),
    (1, 0, 4, 4, 2, "space_or_time", False, 0.1),     # perturbed mesh
])
def test_vcycle_graph_replay_bitwise(ttype, k, n, nsteps, p, ctype, pmg, distort, number, tmp_path):
    """plain launches, the application that records the cycle into a hipGraph, and its replay (host/test_host_stmg.cpp)"""
    exe = os.path.join(HOST, "test_host_stmg")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    res = subprocess.run([exe, str(ttype), str(k), str(n), str(nsteps), str(p), "1" if ctype == "space_and_time" else "0", "1" if pmg else "0", number,
                          str(distort), str(tmp_path / "stmg.bin")], capture_output=True, text=True, timeout=600, env=dict(os.environ, STFEM_MG_GRAPH="1"))
    assert res.returncode == 0, res.stdout + res.stderr
    recorded, replayed, size = (float(x) for x in res.stdout.split("graph: recorded")[1].replace("replayed", "").replace("of", "").split())
    assert size > 0 and recorded == 0.0 and replayed == 0.0, res.stdout
