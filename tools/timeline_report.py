"""Reads the per-wave time stamps of the timeline experiment (profiles/EXPERIMENTS.md, round 3: a scratch build of the emit
kernel that marks s_memrealtime at seven points of every wave of 512 tiles) and prints where a tile's life goes."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8, 8).astype(np.int64)  # tile, wave, slot (10 ns ticks)
ok = (a[:, :, 0] > 0).all(axis=1) & (a[:, :, 7] > 0).all(axis=1)  # tiles with hits in every wave's life
a = a[ok]
us = lambda x: x / 100.0
t0 = a[:, :, 0].min(axis=1, keepdims=True)
print("tiles with complete records: %d" % len(a))
names = ["entry -> loads arrived", "loads -> after scan barrier", "scan -> tables requested", "tables -> list built (peel)",
         "list built -> list barrier passed", "list barrier -> wave done (scoring)"]
pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 7)]
for n, (i, j) in zip(names, pairs):
    d = us(a[:, :, j] - a[:, :, i])
    print("%-40s mean %6.2f us   p10 %6.2f  p50 %6.2f  p90 %6.2f   (max over a tile's waves: mean %6.2f)" %
          (n, d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), d.max(axis=1).mean()))
life = us(a[:, :, 7].max(axis=1) - a[:, :, 0].min(axis=1))
print("workgroup life (first entry -> last wave done): mean %.2f us  p10 %.2f  p50 %.2f  p90 %.2f" %
      (life.mean(), *np.percentile(life, [10, 50, 90])))
lb = us(a[:, 0, 6] - a[:, 0, 5])
print("wave 0: list barrier -> look-back resolved: mean %.2f us  p50 %.2f  p90 %.2f  max %.2f" % (lb.mean(), *np.percentile(lb, [50, 90, 100])))
skew = us(a[:, :, 7].max(axis=1) - a[:, :, 7].min(axis=1))
print("end skew between a tile's waves: mean %.2f us  p90 %.2f" % (skew.mean(), np.percentile(skew, 90)))
peel_skew = us(a[:, :, 4].max(axis=1) - a[:, :, 4].min(axis=1))
print("skew of 'list built' between a tile's waves: mean %.2f us  p90 %.2f" % (peel_skew.mean(), np.percentile(peel_skew, 90)))
span = us(a[:, :, 7].max() - a[:, :, 0].min())
print("all %d tiles ran within %.1f us -> %.2f tiles in flight on average (machine: 768 slots)" % (len(a), span, life.sum() / span))
