#!/usr/bin/env python3
"""bench.py -- gRNAs scored / s of the MI355X PAM-scan + score path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (crp_scan_score: ONE kernel launch that scans, compacts and
scores, taking its table offsets from a chained scan across workgroups; --two-pass selects the
count -> tile scan -> emit+score launch sequence instead) over the rank's arena, with the packed
genome already resident in HBM.  Contigs are
independent, so at N > 1 the steps run with NO collective on the data path; the
path's one exchange -- the FINAL RCCL gatherv of the per-rank hit tables to rank 0 --
runs once after the timed steps and is reported on its own (`gatherv`), together with
the rate that includes it (`value_with_final_gatherv`).  --gather-every-step puts it
inside every step instead.
Workload at N = 1: the >= 1 Gb crop genome BASELINE.json's target is quoted on
("switchgrass-like", SURVEY.md 8d cfg 5, seeded synthetic).  Weak scaling: N ranks
process N such genomes (seeds 0..N-1), contigs dealt to ranks by LPT.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline`
(HIP-event timing of the emit+score kernel on the library's stream) and, at N = 1,
`cpu_baseline` (the reference-faithful numpy port on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="switchgrass", choices=["switchgrass", "tair10", "ecoli"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the switchgrass-like genome (debug)")
    ap.add_argument("--two-pass", action="store_true",
                    help="count / tile-scan / emit launch sequence (CRP_OPT_TWO_PASS=1) instead of the default single launch")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the final RCCL gatherv altogether")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="N > 1: run the gatherv inside every timed step instead of once at the end")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo (+ --share-gpu0) rehearses the N > 1 control flow on one GPU")
    ap.add_argument("--share-gpu0", action="store_true", help="rehearsal only: every rank uses device 0")
    ap.add_argument("--cpu-sample-bases", type=int, default=40000000,
                    help="upper bound on the bases of the same workload timed on the CPU port; the actual "
                         "sample is sized for about 12 s of CPU work (0 = skip)")
    return ap.parse_args()


def make_workload(name, genome, scale):
    import bench_workload as bw
    if name == "switchgrass":
        return bw.switchgrass_like(genome, scale)
    if name == "tair10":
        return bw.tair10_like()
    return bw.ecoli_like()


def cpu_baseline(sample_string, n_bases):
    """Reference-faithful numpy port (oracle/faithful_port.py) on one core."""
    from oracle import faithful_port as fp
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        # size the sample for about 12 s of CPU work: time a 200 kb probe first
        probe = min(200000, n_bases)
        t0 = time.perf_counter()
        fp.scan_score(sample_string[:probe + 1])
        rate = probe / (time.perf_counter() - t0)
        n_bases = int(min(n_bases, max(probe, 12.0 * rate)))
        t0 = time.perf_counter()
        rows, scores = fp.scan_score(sample_string[:n_bases + 1])
        dt = time.perf_counter() - t0
    scored = int((scores != -1.0).sum())
    return {"value": scored / dt, "unit": "gRNAs/s", "cores": 1, "kind": "port",
            "sample": "first %d bases of contig 0 of the same workload, scan+score only "
                      "(%d gRNAs in %.1f s; %.1f kb/s)" % (n_bases, scored, dt, n_bases / dt / 1e3)}


def main():
    args = parse_args()
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.share_gpu0 else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            # a collective that cannot complete must end in an exception, not in a hang or an abort:
            # the bench line is printed either way (the final gatherv is wrapped in try/except)
            import datetime
            os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group("gloo")

    from cropsr_amd import Engine
    from cropsr_amd import parallel

    eng = Engine(local_rank)  # raises without libcropsr_hip.so / GPU: no fallback
    if args.two_pass:
        eng.configure(two_pass=True)

    # ---- workload: `world` genomes, contigs dealt to ranks by LPT (weak scaling)
    genomes = [make_workload(args.workload, g, args.scale) for g in range(world)]
    all_specs = [(g, k) for g in range(world) for k in range(len(genomes[g].specs))]
    lengths = [genomes[g].specs[k].length + 4 for g, k in all_specs]  # + decoration
    owner = parallel.partition_contigs(lengths, world)
    mine = [i for i, o in enumerate(owner) if o == rank]
    t_gen = time.perf_counter()
    builder = eng.arena_builder([lengths[i] for i in mine])
    sample = None
    my_bases = 0
    t_upload = 0.0
    for i in mine:
        g, k = all_specs[i]
        s = genomes[g].contig_string(k)
        if sample is None and rank == 0:
            sample = s[:args.cpu_sample_bases + 1].tobytes().decode()  # keeps the leading quote
        t_up = time.perf_counter()
        builder.add(s)  # characters over PCIe + the ballot pack kernel, synchronous
        t_upload += time.perf_counter() - t_up
        my_bases += genomes[g].specs[k].length
        del s
    arena = builder.seal()
    t_gen = time.perf_counter() - t_gen

    gather = None
    if world > 1 and not args.no_gather:
        gather = parallel.TableGather(dst=0)

    def step():
        n_plus, n_minus = arena.scan_score_device(20, want_pre=False)
        if gather is not None and args.gather_every_step:
            gather(parallel.device_tables_as_tensors(arena, n_plus, n_minus))
        return n_plus, n_minus

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        n_plus, n_minus = step()
    if args.warmup == 0:
        n_plus, n_minus = arena.scan_score_device(20, want_pre=False)
    # units: kept hits that got a real score (a complete 30-window)
    t = parallel.device_tables_as_tensors(arena, n_plus, n_minus)
    scored = int((t["score_plus"] != -1.0).sum().item() + (t["score_minus"] != -1.0).sum().item())
    del t
    t_fetch = time.perf_counter()
    if rank == 0:
        arena.fetch(n_plus, n_minus)  # D2H of the tables into pageable numpy arrays (outside the timed region)
    t_fetch = time.perf_counter() - t_fetch

    # count / scan kernel times from a few extra steps outside the timed region; inside it only the
    # emit+score kernel (the one the roofline is quoted on) is bracketed by HIP events
    eng.profile(2)
    eng.profile_read(reset=True)
    for _ in range(3):
        arena.scan_score_device(20, want_pre=False)
    side = eng.profile_read(reset=True)
    eng.profile(1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_read(reset=True)
    prof["count"], prof["tile_scan"] = side["count"], side["tile_scan"]
    eng.profile(0)

    tot = torch.tensor([dt, float(scored), float(my_bases), float(n_plus + n_minus)], dtype=torch.float64,
                       device="cuda")
    if dist is not None:
        mx = tot[:1].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot[1:].clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx.item())
        scored_all, bases_all, hits_all = [float(x) for x in sm.tolist()]
    else:
        scored_all, bases_all, hits_all = float(scored), float(my_bases), float(n_plus + n_minus)

    # the final exchange, once, timed on its own (barrier + sync on both sides, max over ranks)
    gather_info = None
    if gather is not None and not args.gather_every_step:
        try:
            gather(parallel.device_tables_as_tensors(arena, n_plus, n_minus))  # warm-up: RCCL sets up its P2P channels
            fence()
            tg = time.perf_counter()
            gather(parallel.device_tables_as_tensors(arena, n_plus, n_minus))
            fence()
            tg = time.perf_counter() - tg
            gather_info = {"s": tg}  # rank 0 finishes last: it waits for every receive
        except Exception as e:  # keep the bench line even if the exchange fails on this node
            gather_info = {"error": repr(e)[:300]}

    if rank == 0:
        info = eng.device_info()
        n_chars = arena.stats()["n_chars"]
        hits = n_plus + n_minus
        algo_bytes = (n_chars + 3) // 4 + 2 * ((n_chars + 7) // 8) + 12 * hits  # SURVEY.md 8d, rank 0's launch
        emit = prof["emit_score"]
        emit_ms = emit["ms"] / max(1, emit["launches"])
        achieved = algo_bytes / (emit_ms * 1e-3) / 1e9
        path_ms = sum(p["ms"] / max(1, p["launches"]) for p in prof.values())
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("workload") == genomes[0].name and tj.get("kernel") == "emit_kernel":
                traffic = tj.get("hbm_bytes_per_launch")
        line = {
            "metric": "gRNAs scored/sec", "value": scored_all * args.steps / dt, "unit": "gRNAs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": genomes[0].name, "genomes": world, "contigs_per_genome": len(genomes[0].specs),
                       "bases_total": int(bases_all), "kept_hits_total": int(hits_all),
                       "guide_len": 20,
                       # a look-back time-out would switch the context to the three-launch sequence for good
                       "launches_per_step": 3 if (args.two_pass or side["count"]["launches"] > 0) else 1,
                       "parallelism": ("contigs by LPT over %d ranks" % world) +
                       ("" if gather is None else (" + %s gatherv to rank 0 " % ("RCCL" if args.backend == "nccl" else "gloo (host-staged)") +
                                                   ("every step" if args.gather_every_step else "once, after the steps"))),
                       "device": info["name"].strip()},
            "bases_per_s": bases_all * args.steps / dt,
            "roofline": {"bound": "hbm",
                         "kernel": "emit_kernel (scan+compact+score)" if args.two_pass else
                                   "emit_kernel, single launch (masks + chained tile offsets + compact + score)",
                         "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": int(algo_bytes),
                         "kernel_ms": emit_ms, "all_kernels_ms": path_ms,
                         "count_kernel_ms": prof["count"]["ms"] / max(1, prof["count"]["launches"]),
                         "tile_scan_ms": prof["tile_scan"]["ms"] / max(1, prof["tile_scan"]["launches"])},
            "setup_s": {"generate_pack_upload": t_gen},
            # host-buffer boundary: characters H2D + pack, one scan, tables D2H (never `value`)
            "pcie_inclusive": {"upload_pack_s": t_upload, "fetch_tables_s": t_fetch,
                               "gRNAs_per_s": scored / (t_upload + dt / args.steps + t_fetch)},
        }
        if gather_info is not None:
            if "s" in gather_info:
                moved = 12.0 * (hits_all - hits)  # bytes that crossed xGMI to rank 0
                gather_info.update({"bytes_to_root": int(moved), "GB_per_s_into_root": moved / gather_info["s"] / 1e9})
                line["value_with_final_gatherv"] = scored_all * args.steps / (dt + gather_info["s"])
            line["gatherv"] = gather_info
        if world == 1 and args.cpu_sample_bases > 0:
            from oracle import oracle as _o
            _o.lib()
            line["cpu_baseline"] = cpu_baseline(sample, args.cpu_sample_bases)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)

    arena.close()
    eng.close()
    if dist is not None:
        if gather_info is not None and "error" in gather_info:
            sys.stdout.flush()
            os._exit(0)  # the process group is in an unknown state after a failed exchange: no collective teardown
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
