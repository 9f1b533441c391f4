"""Temporal matrices: oracle vs (a) the reference's own golden tests/tp_02.output (committed
as a data fixture) and (b) the exact mpmath derivation in tests/golden/time_weights_exact.json.

Reference: include/fe_time.h:351-409, 485-514, 643-744, 157-305; tests/tp_02.cc:12-27 (format:
%7.2f, entries with |x| < 0.01 printed as 7 blanks, matrices separated by empty lines)."""
import json
import os
import re

import numpy as np
import pytest


def parse_tp02(path):
    """-> list of (header, [matrices]) in file order."""
    sections = []
    cur = None
    rows = None
    with open(path) as f:
        for raw in f:
            line = raw.rstrip("\n")
            if re.match(r"^(CG|DG|Waves|Stokes|Evolutionary|Extrapolation)", line):
                if rows:
                    cur[1].append(rows)
                rows = None
                cur = (line.strip(), [])
                sections.append(cur)
                continue
            if line == "":
                if rows:
                    cur[1].append(rows)
                rows = None
                continue
            assert len(line) % 7 == 0, repr(line)
            vals = []
            for k in range(len(line) // 7):
                tok = line[7 * k:7 * k + 7]
                vals.append(None if tok.strip() == "" else float(tok))
            rows = (rows or []) + [vals]
    if rows:
        cur[1].append(rows)
    return sections


def assert_matches_print(mat, golden_rows, what):
    mat = np.atleast_2d(mat)
    assert mat.shape == (len(golden_rows), len(golden_rows[0])), what
    for i, row in enumerate(golden_rows):
        for j, g in enumerate(row):
            v = mat[i, j]
            if g is None:
                assert abs(v) < 0.01 + 1e-9, (what, i, j, v)
            else:
                assert abs(v - g) <= 0.005 + 1e-6, (what, i, j, v, g)


@pytest.fixture(scope="module")
def tp02(golden_dir):
    secs = parse_tp02(os.path.join(golden_dir, "tp_02.output"))
    return secs


def test_tp02_single_step(tp02, oracle_mod):
    o = oracle_mod
    it = iter(tp02)
    checked = 0
    secs = list(tp02)
    for k, (hdr, mats) in enumerate(secs):
        m = re.match(r"^(CG|DG)\((\d)\)$", hdr)
        if not m:
            continue
        kind, r = m.group(1), int(m.group(2))
        waves_hdr, waves = secs[k + 1]
        assert waves_hdr == "Waves"
        if kind == "CG":
            M, D = o.cg_weights(r)
            assert_matches_print(M, mats[0], hdr + " M")
            assert_matches_print(D, mats[1], hdr + " D")
        else:
            M, D, j = o.dg_weights(r)
            assert_matches_print(j, mats[0], hdr + " jump")
            assert_matches_print(M, mats[1], hdr + " M")
            assert_matches_print(D, mats[2], hdr + " D")
        W = o.time_weights_wave(o.CGP if kind == "CG" else o.DG, r, 1.0, 1)
        assert len(waves) == 5
        for a, (w, g) in enumerate(zip(W, waves)):
            assert_matches_print(w, g, f"{hdr} wave[{a}]")
        checked += 1
    assert checked == 10


def test_tp02_multi_step(tp02, oracle_mod):
    o = oracle_mod
    checked = 0
    for hdr, mats in tp02:
        m = re.match(r"^(Waves )?(CG|DG)\((\d)\) - (\d) timesteps in one system$", hdr)
        if not m:
            continue
        wave, kind, r, ns = bool(m.group(1)), m.group(2), int(m.group(3)), int(m.group(4))
        t = o.CGP if kind == "CG" else o.DG
        got = o.time_weights_wave(t, r, 1.0, ns) if wave else o.time_weights(t, r, 1.0, ns)
        assert len(mats) == len(got), hdr
        for a, (w, g) in enumerate(zip(got, mats)):
            assert_matches_print(w, g, f"{hdr} [{a}]")
        checked += 1
    assert checked == 24


def test_exact_single_step(golden_dir, oracle_mod):
    o = oracle_mod
    with open(os.path.join(golden_dir, "time_weights_exact.json")) as f:
        ex = json.load(f)
    for r in range(1, 6):
        M, D = o.cg_weights(r)
        np.testing.assert_allclose(M, np.array(ex[f"cg{r}"]["M"]), rtol=0, atol=2e-14)
        np.testing.assert_allclose(D, np.array(ex[f"cg{r}"]["D"]), rtol=0, atol=2e-13)
    for r in range(0, 6):
        M, D, j = o.dg_weights(r)
        np.testing.assert_allclose(M, np.array(ex[f"dg{r}"]["M"]), rtol=0, atol=2e-14)
        np.testing.assert_allclose(D, np.array(ex[f"dg{r}"]["D"]), rtol=0, atol=5e-13)
        np.testing.assert_allclose(j[:, 0], np.array(ex[f"dg{r}"]["jump"]), rtol=0, atol=5e-13)


def test_known_values(oracle_mod):
    """SURVEY 8a-12 closed forms."""
    o = oracle_mod
    A, B, G, Z = o.time_weights(o.CGP, 2, 1.0, 1)
    np.testing.assert_allclose(A, [[2 / 3, 0], [0, 1 / 6]], atol=1e-14)
    np.testing.assert_allclose(B, [[4 / 3, 1 / 3], [-4 / 3, 2 / 3]], atol=1e-14)
    np.testing.assert_allclose(G[:, 0], [-1 / 3, 1 / 6], atol=1e-14)
    np.testing.assert_allclose(Z[:, 0], [5 / 3, -2 / 3], atol=1e-14)
    A, B, G, Z = o.time_weights(o.CGP, 1, 0.5, 1)
    np.testing.assert_allclose([A[0, 0], B[0, 0], G[0, 0], Z[0, 0]], [0.25, 1, -0.25, 1],
                               atol=1e-15)
    A, B, G, Z = o.time_weights(o.DG, 1, 1.0, 1)
    np.testing.assert_allclose(A, [[3 / 4, 0], [0, 1 / 4]], atol=1e-14)
    np.testing.assert_allclose(B, [[9 / 8, 3 / 8], [-9 / 8, 5 / 8]], atol=1e-14)
    np.testing.assert_allclose(G[:, 0], [3 / 2, -1 / 2], atol=1e-14)
    assert np.all(Z == 0)
