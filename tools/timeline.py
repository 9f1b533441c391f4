#!/usr/bin/env python3
"""Phase timeline of the tile kernel (diagnostic build: tools/build_abl.sh -DSTFEM_TIMELINE, run with
STFEM_LIB=.../libstfem_abl.so STFEM_TIMELINE=<file>).  Prints mean phase durations per layer and
how the workgroups sharing a CU overlap."""
import sys
import numpy as np

NAMES = ["0-1 mask/coef/xslab-load issue", "1-2 cell core", "2-3 barrier A", "3-4 owner init",
         "4-5 barrier B", "5-6 adds + src prefetch issue", "6-7 barrier C", "7-8 store phase",
         "8-9 carry read", "9-10 barrier D", "10-11 wait src prefetch"]

raw = np.fromfile(sys.argv[1], dtype=np.int64)
nblk, nw, lz, ns = raw[:4]
t = raw[4:].reshape(nblk, nw, lz, ns)
used = t[:, 0, 0, 0] > 0
t = t[used]
print(f"{t.shape[0]} recorded workgroups x {nw} waves x {lz} layers; tick = 10 ns")
lay_ok = t[..., 0] > 0
d = np.diff(t[..., :12], axis=-1).astype(float)[lay_ok] / 100.0  # us
tot = ((t[..., 11] - t[..., 0])[lay_ok]) / 100.0
print(f"layer time (0->11): mean {tot.mean():.2f} us, median {np.median(tot):.2f}, p90 {np.percentile(tot, 90):.2f}")
for i, n in enumerate(NAMES):
    print(f"  {n:36s} mean {d[..., i].mean():6.2f} us  median {np.median(d[..., i]):6.2f}  p90 {np.percentile(d[..., i], 90):6.2f}")
wg_start, wg_end = t[:, :, 0, 0].min(1), t[:, :, :, 11].max(axis=(1, 2))
print(f"workgroup lifetime mean {(wg_end - wg_start).mean() / 100:.1f} us; kernel span {(wg_end.max() - wg_start.min()) / 100:.1f} us")
# which CU: HW_ID cu_id[11:8] sh_id[12] se_id[15:13] (gfx9 layout), XCC_ID low bits
hw = t[:, 0, 0, 15]
cu = ((hw >> 32) & 0xF) * 4096 + (hw & 0xFF00)
ids, counts = np.unique(cu, return_counts=True)
print(f"distinct (xcc, se, sh, cu) keys: {len(ids)}; workgroups per key: min {counts.min()} max {counts.max()}")
# concurrency: at the start of each workgroup, how many others on the same key are alive
conc = []
for key in ids[:64]:
    m = cu == key
    s, e = wg_start[m], wg_end[m]
    for i in range(len(s)):
        conc.append(int(((s <= s[i]) & (e > s[i])).sum()))
print("workgroups alive on the same CU at a workgroup's start (incl. itself): mean %.2f" % np.mean(conc))
# phase overlap of co-resident pairs: fraction of time both are inside the cell core
key = ids[0]
m = np.where(cu == key)[0]
print("first CU: workgroup start/end (us rel.):", [(round((wg_start[i] - wg_start[m].min()) / 100, 1), round((wg_end[i] - wg_start[m].min()) / 100, 1)) for i in m])

# do the workgroups march in lockstep?  fraction of live workgroups (wave 0) inside the cell core /
# inside a memory-issuing phase, sampled every microsecond
t0 = t[:, 0, :, :12]
valid = t0[..., 0] > 0  # chunks shorter than the longest leave unused layer slots at zero
lo, hi = t0[..., 0][valid].min(), t0[..., 11][valid].max()
assert hi - lo < 10 ** 8, (lo, hi)
ticks = np.arange(lo, hi, 100)
core = np.zeros(len(ticks)); mem = np.zeros(len(ticks)); alive = np.zeros(len(ticks))
for w in range(t0.shape[0]):
    for l in range(t0.shape[1]):
        s = t0[w, l]
        if s[0] == 0:
            continue
        a = np.searchsorted(ticks, [s[0], s[11], s[1], s[2], s[5], s[6], s[7], s[8]])
        alive[a[0]:a[1]] += 1
        core[a[2]:a[3]] += 1
        mem[a[4]:a[5]] += 1
        mem[a[6]:a[7]] += 1
ok = alive > 0.5 * alive.max()
fc, fm = core[ok] / alive[ok], mem[ok] / alive[ok]
print(f"fraction of live workgroups in the core: mean {fc.mean():.2f} min {fc.min():.2f} max {fc.max():.2f} std {fc.std():.3f}")
print(f"fraction in gather-issue / store phases: mean {fm.mean():.2f} min {fm.min():.2f} max {fm.max():.2f} std {fm.std():.3f}")
print("core fraction per us (first 60):", " ".join(f"{x:.2f}" for x in fc[:60]))

# average number of workgroups resident on the chip = sum of lifetimes / kernel span
print(f"mean resident workgroups: {(wg_end - wg_start).sum() / (wg_end.max() - wg_start.min()):.1f}")
first = wg_start.min()
print("workgroups started within 5 / 20 us of the first:", int((wg_start < first + 500).sum()), int((wg_start < first + 2000).sum()))
ev = np.concatenate([np.stack([wg_start, np.ones_like(wg_start)], 1), np.stack([wg_end, -np.ones_like(wg_end)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
print("maximum number of concurrently resident workgroups:", int(np.cumsum(ev[:, 1]).max()))
