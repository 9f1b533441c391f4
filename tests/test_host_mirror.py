"""The C++ host-side mirror (dealii-stfem_amd/host/stfem/operators.h: MatrixFreeOperator,
SystemMatrix with the reference's method names) driven by a C++ caller, checked against the oracle."""
import importlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dealii-stfem_amd", "host")


def test_host_mirror_compiles():
    """build() compiles the header-only mirror against include/stfem.h (no GPU needed)."""
    stfem = importlib.import_module("dealii-stfem_amd")
    if not os.path.exists(stfem.LIB_PATH):
        stfem.build()
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(HOST, "test_host_mirror"))
    assert os.path.exists(os.path.join(HOST, "test_host_stokes"))
    assert os.path.exists(os.path.join(HOST, "test_host_sharded"))


@pytest.mark.gpu
@pytest.mark.parametrize("number", ["double", "float"])
@pytest.mark.parametrize("case", [(2, (4, 3, 5), 0, 2, 2), (3, (3, 3, 2), 1, 1, 1), (4, (7, 5, 3), 0, 2, 1)])
def test_cpp_caller_matches_oracle(case, number, tmp_path, oracle_mod):
    p, nc, ttype, r, ns = case
    tol = 1e-12 if number == "double" else 1e-5
    stfem = importlib.import_module("dealii-stfem_amd")
    exe = os.path.join(HOST, "test_host_mirror")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    out = tmp_path / "out.bin"
    res = subprocess.run([exe, str(p), *map(str, nc), str(ttype), str(r), str(ns), str(out), number],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "exceptions=3" in res.stdout
    raw = np.fromfile(out, dtype=np.uint64, count=2)
    nb, n = int(raw[0]), int(raw[1])
    flat = np.fromfile(out, dtype=np.float64, offset=16)
    data = flat[:9 * nb * n].reshape(9, nb, n)
    X, Y, YT, RHS, RES0, RES1, YJ, DIAG, DINV = data
    DK, DKI = flat[9 * nb * n:9 * nb * n + 2 * n].reshape(2, n)
    SM = flat[9 * nb * n + 2 * n:].reshape(nb, n)
    Alpha, Beta, Gamma, Zeta = stfem.get_fe_time_weights(ttype, r, 1.0 / 32, ns)
    verts = stfem.mesh_vertices(nc)
    orc = oracle_mod.Oracle(p, nc, verts, 63)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(Y, orc.st_vmult(Alpha, Beta, X)) < tol
    assert rel(YT, orc.st_vmult(Alpha, Beta, X, transpose=True)) < tol
    g, z = (Gamma, Zeta) if ttype == 0 else (np.zeros_like(Gamma), Gamma)  # tests/tp_01.cc:160-166
    ref = orc.st_vmult(g, z, X[:1])
    assert rel(RHS, 2 * ref) < tol  # vmult_slice followed by vmult_slice_add
    # PDE<> (operators.h:1953-2050): residual = rhs - form, vmult = Jacobian
    assert np.linalg.norm(RES0) < 10 * tol * np.linalg.norm(Y)
    assert rel(RES1, 0.5 * Y) < 10 * tol
    assert rel(YJ, Y) < tol
    # diagonals: spatial (1092-1110) and space-time (613-637, combined exactly as the reference does)
    dK, dM = orc.diagonal(0.0, 1.0), orc.diagonal(1.0, 0.0)
    guard = np.sqrt(np.finfo(np.float64 if number == "double" else np.float32).eps)
    inv = lambda d: np.where(np.abs(d) > guard, 1.0 / np.where(d == 0, 1.0, d), 1.0)  # noqa: E731
    assert rel(DK, dK) < tol and rel(DKI, inv(dK)) < 10 * tol
    for i in range(nb):
        assert rel(DIAG[i], Alpha[i, i] * dK + Beta[i, i] * dM) < tol
        assert rel(DINV[i], inv(dK) / Alpha[i, i] + inv(dM) / Beta[i, i]) < 10 * tol
    # PreconditionVanka (stmg.h:619-907) from the C++ mirror against the numpy restatement
    from oracle import vanka_oracle
    want = vanka_oracle.VankaOracle(p, nc, verts, 63, Alpha, Beta).vmult(X)
    assert rel(SM, want) < (1e-10 if number == "double" else 2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [((3, 2, 4), 0, 2, 1, 0.5, 0), ((2, 3, 2), 1, 1, 2, 2.0, 0), ((3, 3, 2), 0, 2, 1, 0.4, 0b100011)])
def test_cpp_stokes_caller_matches_oracle(case, tmp_path):
    """SystemMatrixStokes / StokesMatrixFreeOperator / StokesNitscheMatrixFreeOperator mirror (host/stfem/stokes.h) vs the CPU oracle;
    the last case with weak (Nitsche) boundary ids 0, 1, 5."""
    from oracle import oracle
    nc, ttype, r, ns, nu, weak = case
    stfem = importlib.import_module("dealii-stfem_amd")
    exe = os.path.join(HOST, "test_host_stokes")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    out = tmp_path / "stokes.bin"
    res = subprocess.run([exe, *map(str, nc), str(ttype), str(r), str(ns), str(nu), str(out)] + ([str(weak)] if weak else []),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    nb = int(raw[:8].view(np.uint64)[0])
    off, X = 8, []
    for _ in range(nb):
        n = int(raw[off:off + 8].view(np.uint64)[0]); off += 8
        X.append(raw[off:off + 8 * n].view(np.float64).copy()); off += 8 * n
    Y = []
    for b in range(nb):
        n = X[b].size
        Y.append(raw[off:off + 8 * n].view(np.float64).copy()); off += 8 * n
    verts = stfem.mesh_vertices(nc, distort=0.1, seed=99)
    orc = oracle.StokesOracle(nc, verts, 63 & ~weak, nu, weak_mask=weak)
    Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(ttype, r, 1.0 / 32, ns)
    nt = r if ttype == 0 else r + 1
    ref = orc.st_vmult(Alpha, Beta, ns, nt, X)
    for b in range(nb):
        assert np.linalg.norm(Y[b] - ref[b]) <= 1e-12 * np.linalg.norm(ref[b]), b
    if weak:
        FU = raw[off:off + 8 * 3 * orc.n_u].view(np.float64); off += 8 * 3 * orc.n_u
        FP = raw[off:off + 8 * orc.n_p].view(np.float64)
        pts = orc.face_points()
        G = np.stack([np.sin(pts[:, 0] + 2 * pts[:, 1]), pts[:, 2] ** 2 - pts[:, 0], np.cos(pts[:, 1] * pts[:, 2])], axis=1)
        ru, rp = orc.nitsche_rhs(G)
        assert np.linalg.norm(FU - ru.reshape(-1)) <= 1e-12 * np.linalg.norm(ru) and np.linalg.norm(FP - rp) <= 1e-12 * np.linalg.norm(rp)
