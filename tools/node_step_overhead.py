"""Host-side cost of one crp_node_scan_score as the number of devices grows: a genome so small that the kernels are all
launch latency (one SMALL tile per device), N logical devices on GPU 0.  usage: python tools/node_step_overhead.py [N ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cropsr_amd import node as nd  # noqa: E402

rng = np.random.default_rng(1)
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    contigs = [b"'" + rng.choice(alpha, 20000 * world).tobytes() + b"')]"]
    with nd.Node([0] * world) as node:
        node.load(contigs)
        for _ in range(50):
            node.scan_score_device(20)
        t0 = time.perf_counter()
        for _ in range(500):
            node.scan_score_device(20)
        dt = (time.perf_counter() - t0) / 500
    print(json.dumps({"logical_devices": world, "us_per_node_scan": round(dt * 1e6, 1)}))
