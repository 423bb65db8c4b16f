"""The native CSV formatter (crp_write_rows_ex) alone: rows/s and GB/s against the thread count, into /dev/null (formatting
only) and into a file under $TMPDIR (formatting + write(2)).  No GPU involved."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cropsr_amd import _native as nat  # noqa: E402

L = nat.lib()
rng = np.random.default_rng(1)
n_chars, n_rows = 60_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
text = rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), n_chars)
pos = np.sort(rng.integers(40, n_chars - 40, n_rows).astype(np.uint32))
minus = (rng.integers(0, 2, n_rows)).astype(np.uint8)
score = rng.random(n_rows)
ids = rng.choice(np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789", dtype=np.uint8), (n_rows, 7))
chrom = b"[('Chr01K',"
tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), "format_bench.csv")
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # rows per call (0: one call; the CLI writes 1 000 000 per call)
for target in ("/dev/null", tmp):
    for nt in (1, 4, 8, 16, 32):
        fd = os.open(target, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        written, total = ctypes.c_uint64(), 0
        t0 = time.perf_counter()
        for a in range(0, n_rows, chunk or n_rows):
            b = min(n_rows, a + (chunk or n_rows))
            st = L.crp_write_rows_ex(fd, text.ctypes.data_as(nat.u8p), text.size, ctypes.cast(ctypes.c_char_p(chrom), nat.u8p),
                                     len(chrom), 20, pos[a:b].ctypes.data_as(nat.u32p), minus[a:b].ctypes.data_as(nat.u8p),
                                     score[a:b].ctypes.data_as(nat.f64p), ids[a:b].ctypes.data_as(nat.u8p), b - a, None, None,
                                     None, None, ctypes.byref(written), nt)
            assert st == 0
            total += written.value
        dt = time.perf_counter() - t0
        os.close(fd)
        print("%-10s threads %2d: %.3f s  %.1f M rows/s  %.2f GB/s  (%.0f ns per row and thread)" %
              ("null" if target == "/dev/null" else "file", nt, dt, n_rows / dt / 1e6, total / dt / 1e9, dt * nt / n_rows * 1e9),
              flush=True)
if os.path.exists(tmp):
    os.unlink(tmp)
