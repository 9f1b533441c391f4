"""Space transfers of the multigrid (stfem_transfer_*) on the cfg-1 mesh: time per call and HBM GB/s against the
algorithmic bytes (prolongate_and_add: read coarse, read + write fine; restrict_and_add: read fine, read + write coarse).
Run on the GPU box: python tools/transfer_bench.py > gpurun_out/transfer_bench.txt"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stfem = importlib.import_module("dealii-stfem_amd")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


torch.zeros(1, device="cuda")
for number, es in (("double", 8), ("float", 4)):
    for name, pf, ncf, pc, ncc in (("h  72^3 -> 36^3, Q4", 4, (72,) * 3, 4, (36,) * 3), ("p  Q4 -> Q2, 72^3", 4, (72,) * 3, 2, (72,) * 3),
                                   ("h  36^3 -> 18^3, Q4", 4, (36,) * 3, 4, (18,) * 3)):
        fine = stfem.MatrixFreeOperator(pf, ncf, number=number)
        coarse = stfem.MatrixFreeOperator(pc, ncc, number=number)
        T = stfem.MGTwoLevelTransfer(fine, coarse)
        nb = 2
        uf, uc = stfem.BlockVector(fine, nb), stfem.BlockVector(coarse, nb)
        tp = timed(lambda: T.prolongate_and_add(uf, uc))
        tr = timed(lambda: T.restrict_and_add(uc, uf))
        bp = nb * es * (coarse.n_dofs + 2 * fine.n_dofs)
        br = nb * es * (fine.n_dofs + 2 * coarse.n_dofs)
        print(f"{number:6s} {name:22s} {nb} blocks: prolongate_and_add {tp:7.3f} ms = {bp / tp / 1e6:7.0f} GB/s ({bp / tp / 8e9 * 100:4.1f} % of 8 TB/s)   "
              f"restrict_and_add {tr:7.3f} ms = {br / tr / 1e6:7.0f} GB/s ({br / tr / 8e9 * 100:4.1f} %)")
        del T, uf, uc, fine, coarse
