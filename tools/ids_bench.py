"""crp_legacy_ids alone: crispr ids per second (the draws are one sequential MT19937 stream: a serial floor of the CSV stage)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cropsr_amd import rows  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
np.random.seed(1)
rows.draw_ids(1000)
for rep in range(3):
    t0 = time.perf_counter()
    a = rows.draw_ids(n, reverse=True)
    dt = time.perf_counter() - t0
    print("%d ids in %.3f s = %.1f M ids/s (%.2f ns per character)" % (n, dt, n / dt / 1e6, dt / n / 7 * 1e9), flush=True)
