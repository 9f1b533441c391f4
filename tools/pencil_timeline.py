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
names = ["0-1 mask, forward", "1-2 next-group addresses", "2-3 middle (+ src prefetch)", "3-4 backward (+ dst stores)",
         "4-5 carry rotation, wait for the prefetch"]
for k, n in enumerate(names):
    d = (t[..., k + 1] - t[..., k])[ok] / 100.0
    print(f"  {n:46s} mean {d.mean():6.2f} us  median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}")
g = (t[..., 5] - t[..., 0])[ok] / 100.0
print(f"  group (0->5) mean {g.mean():.2f} us median {np.median(g):.2f}")
start = np.where(ok, t[..., 0], np.inf).min(axis=(1, 2))
end = np.where(ok, t[..., 5], 0).max(axis=(1, 2))
use = np.isfinite(start) & (end > 0)
life = (end - start)[use] / 100.0
span = (end[use].max() - start[use].min()) / 100.0
print(f"  tile lifetime (first stamp -> last stamp) mean {life.mean():.1f} us, p90 {np.percentile(life, 90):.1f}; "
      f"kernel span {span:.1f} us; tiles {int(use.sum())}")
# how many tiles are in flight over time (sampled every microsecond)
ticks = np.arange(start[use].min(), end[use].max(), 100)
alive = ((start[use][None, :] <= ticks[:, None]) & (end[use][None, :] > ticks[:, None])).sum(1)
print(f"  tiles in flight: mean {alive.mean():.0f}, max {alive.max()}; first / last 10 % of the span: "
      f"{alive[:max(1, len(alive) // 10)].mean():.0f} / {alive[-max(1, len(alive) // 10):].mean():.0f}")
# per wave: time between the end of one group and the start of the next (tile changes excluded)
gap = (t[:, :, 1:, 0] - t[:, :, :-1, 5])[ok[:, :, 1:] & ok[:, :, :-1]] / 100.0
print(f"  gap between consecutive groups of a wave: mean {gap.mean():.2f} us median {np.median(gap):.2f} p90 {np.percentile(gap, 90):.2f}")
