#!/usr/bin/env python3
"""Phase timeline of the pencil kernel (diagnostic build: tools/build_pencil_exp.sh tl
-DSTFEM_PENCIL_TIMELINE, run with STFEM_LIB=.../libstfem_tl.so STFEM_TIMELINE=<file>)."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.int64)
nblk, nw, ng, ns = raw[:4]
t = raw[4:4 + nblk * nw * ng * ns].reshape(nblk, nw, ng, ns).astype(np.float64)
ok = t[..., 0] > 0
print(f"{int(ok.any(axis=(1, 2)).sum())} recorded workgroups x {nw} waves x {ng} groups; tick = 10 ns")
names = ["0-1 mask, xin issue, forward (waits for src)", "1-2 prefetch issue", "2-3 middle", "3-4 xout, backward",
         "4-5 carries + stores"]
for k, n in enumerate(names):
    d = (t[..., k + 1] - t[..., k])[ok] / 100.0
    print(f"  {n:46s} mean {d.mean():6.2f} us  median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}")
g = (t[..., 5] - t[..., 0])[ok] / 100.0
print(f"  group (0->5) mean {g.mean():.2f} us median {np.median(g):.2f}")
# layer end stamps live in the cyl = 0 slot of each layer: 6 = before the barrier, 7 = after
b = t[..., 6] > 0
bw = (t[..., 7] - t[..., 6])[b] / 100.0
print(f"  layer barrier wait mean {bw.mean():.2f} us median {np.median(bw):.2f} p90 {np.percentile(bw, 90):.2f}")
start = np.where(ok, t[..., 0], np.inf).min(axis=(1, 2))
end = np.where(b, t[..., 7], 0).max(axis=(1, 2))
use = np.isfinite(start) & (end > 0)
print(f"  workgroup lifetime mean {((end - start)[use]).mean() / 100:.1f} us; kernel span {(end[use].max() - start[use].min()) / 100:.1f} us; "
      f"started within 5 us: {int((start[use] < start[use].min() + 500).sum())} of {int(use.sum())}")
