"""Time-multigrid transfer matrices (reference include/fe_time.h:749-898) pinned by the reference's
own tests/transfer_02.output (committed as the data golden tests/golden/transfer_02.output): every "Prolongation",
"Restriction" and "Projection" section, at the file's print rule (%7.2f, blank below 0.01; tests/transfer_02.cc:9-26).
Checked: the numpy restatement (oracle/oracle.py) and the product's host code through the C-ABI."""
import importlib
import os
import re

import numpy as np
import pytest


def sections(golden_dir):
    text = open(os.path.join(golden_dir, "transfer_02.output")).read()
    text = text[:text.index("Test MG in time operators")]
    out = []
    for block in re.split(r"\n(?=- (?:Prolongation|Restriction|Projection))", text):
        lines = block.split("\n")
        m = re.match(r"- (Prolongation|Restriction|Projection)", lines[0])
        if not m:
            continue
        kind = m.group(1)
        if kind == "Projection":
            h = re.match(r"(CG|DG) From (\d+) to (\d+)", lines[1])
            n = int(re.match(r"Timesteps at once: (\d+)", lines[2]).group(1))
            key = (kind, h.group(1), int(h.group(2)), int(h.group(3)), n)
            rows = lines[3:]
        else:
            h = re.match(r"(CG|DG)\((\d+)\)", lines[1])
            key = (kind, h.group(1), int(h.group(2)))
            rows = lines[2:]
        mat = []
        for r in rows:
            if not r.strip() and not r.startswith("       "):
                break
            mat.append([r[7 * j:7 * j + 7] for j in range(len(r) // 7)])
        out.append((key, mat))
    return out


def matches(mat, A):
    """the printed cells of a section against a matrix: %7.2f, seven blanks where |a| < 0.01"""
    A = np.asarray(A)
    if len(mat) != A.shape[0]:
        return False
    for i, row in enumerate(mat):
        if len(row) > A.shape[1]:
            return False
        for j in range(A.shape[1]):
            want = row[j] if j < len(row) else "       "
            got = "       " if abs(A[i, j]) < 0.01 else "%7.2f" % A[i, j]
            if got != want:
                # a value within 1e-9 of a rounding boundary may print either way
                if want.strip() and abs(abs(A[i, j] - float(want)) - 0.005) < 1e-9:
                    continue
                return False
    return True


def n_steps_of(key, occurrence):
    """transfer_02.cc:123-153: the prolongation / restriction sections come first with 2 steps at once, later with 4"""
    return 2 if occurrence == 0 else 4


def test_sections_found(golden_dir):
    secs = sections(golden_dir)
    kinds = [k[0] for k, _ in secs]
    assert kinds.count("Prolongation") == 17 and kinds.count("Restriction") == 17 and kinds.count("Projection") == 26


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_time_transfer_matrices_match_reference_output(impl, golden_dir, oracle_mod):
    if impl == "oracle":
        prol, rest, proj = oracle_mod.time_prolongation, oracle_mod.time_restriction, oracle_mod.time_projection
    else:
        stfem = importlib.import_module("dealii-stfem_amd")
        prol, rest, proj = stfem.get_time_prolongation_matrix, stfem.get_time_restriction_matrix, stfem.get_time_projection_matrix
    seen = {}
    for key, mat in sections(golden_dir):
        ttype = 0 if key[1] == "CG" else 1
        if key[0] == "Projection":
            A = proj(ttype, key[2], key[3], key[4])
        else:
            occ = seen.get(key, 0)
            seen[key] = occ + 1
            A = (prol if key[0] == "Prolongation" else rest)(ttype, key[2], n_steps_of(key, occ))
        assert matches(mat, A), (key, np.round(A, 2))
