#!/usr/bin/env python3
"""LDS bank-conflict model of the four slab access patterns of PencilCore (csrc/stfem_core.h) for
8-byte elements on gfx950 (MI355X_MICROARCH.md, LDS table: ds_write_b64 = 4 groups of 16 lanes over
32 banks, ds_read_b64 = 2 groups of 32 lanes over 64 banks).  Prints LDS cycles per cell group for the
dense layout of round 1 and for the padded layout, and searches the strides.
  usage: tools/lds_conflicts.py [P] [NBM]"""
import sys
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NBM = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N = P + 1
CPW = (64 // N) // NBM
lanes = [(l % N, (l // N) % CPW, l // (N * CPW)) for l in range(N * CPW * NBM)]


def cycles(addrs, group, nbanks):
    tot = 0
    for g in range(0, 64, group):
        banks = {}
        for a in addrs[g:g + group]:
            for d in (a, a + 1):
                banks.setdefault(d % nbanks, set()).add(d)
        if banks:
            tot += max(len(v) for v in banks.values())
    return tot


def model(CBS, PS, cb):
    w1 = sum(cycles([2 * (cb(c, b) * CBS + i * PS + y * N + x) for (i, c, b) in lanes], 16, 32) for y in range(N) for x in range(N))
    r2 = sum(cycles([2 * (cb(c, ib) * CBS + x * PS + y * N + k) for (k, c, b) in lanes], 32, 64) for ib in range(NBM) for y in range(N) for x in range(N))
    w3 = sum(cycles([2 * (cb(c, b) * CBS + x * PS + y * N + k) for (k, c, b) in lanes], 16, 32) for y in range(N) for x in range(N))
    r4 = sum(cycles([2 * (cb(c, b) * CBS + i * PS + y * N + x) for (i, c, b) in lanes], 32, 64) for y in range(N) for x in range(N))
    return dict(forward_write=w1, middle_read=r2, middle_write=w3, backward_read=r4, total=w1 + r2 + w3 + r4)


ideal = N * N * (4 + 2 * NBM + 4 + 2)
print(f"P = {P}, NBM = {NBM}: {CPW} cells per wave; conflict-free = {ideal} LDS cycles per cell group")
print("round-1 layout (cell-major blocks, dense):", model(N ** 3, N * N, lambda c, b: c * NBM + b))
PS = 16 * ((N * N - 1 + 15) // 16) + 1
print(f"padded layout (block-major, PS = {PS}, CBS = {N * PS}):", model(N * PS, PS, lambda c, b: b * CPW + c))
