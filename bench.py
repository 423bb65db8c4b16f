#!/usr/bin/env python3
"""bench.py -- gRNAs scored / s of the MI355X PAM-scan + score path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its N ranks itself, cropsr_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

(torch.distributed.run is only the launcher: this program imports no PyTorch.  The ranks find
each other through cropsr_amd.rendezvous and talk over RCCL inside libcropsr_hip.so.)

A step = one pass of the hot path (crp_scan_score: ONE kernel launch that scans, compacts and
scores, taking its table offsets from a chained scan across workgroups; --two-pass selects the
count -> tile scan -> emit+score launch sequence instead) over the rank's arena, with the packed
genome already resident in HBM.  Contigs are independent, so at N > 1 the steps run with NO
collective on the data path; the path's one exchange -- the FINAL RCCL gatherv of the per-rank hit
tables to rank 0 (crp_gather_hits) -- runs once after the timed steps and is reported on its own
(`gatherv`), together with the whole-job rate that includes it (`value_with_final_gatherv` = gRNAs of all
ranks / (one step + the one exchange): the number a scaling curve should be built from, since `value`'s
steps hold no collective; `per_rank` lists every rank's kernel time and share).
--gather-every-step puts it inside every step instead.  A failed exchange still prints the line
(`gatherv_ok: false`) and then exits non-zero; one that never RETURNS is ended by rank 0's watchdog after
--collective-timeout seconds, the line printed with the scan's numbers first.  If the RCCL communicator cannot be CREATED
(or its bootstrap does not return within CROPSR_COMM_INIT_TIMEOUT_S; every rank
learns that together, before any collective), the measurement still runs -- the scan needs no
collective -- with the control sockets as fence and the host transport for the final exchange; the
line then carries `rccl_error` and names the transport in `config.parallelism`, and stderr says so.
The W warm-up steps run immediately before the timed region and are followed by further untimed steps
until --preheat-ms (60) have passed: the GPU's clocks drop during the side measurements that precede
them (a 0.6 GB copy of the tables to the host), and K = 20 steps of 0.5 ms are over before they are
back up.  `untimed_steps_before` reports how many steps ran untimed in all.
Workload at N = 1: the >= 1 Gb crop genome BASELINE.json's target is quoted on
("switchgrass-like", SURVEY.md 8d cfg 5, seeded synthetic).  Weak scaling: N ranks
process N such genomes (seeds 0..N-1), contigs dealt to ranks by LPT.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (HIP-event timing of
the emit+score kernel on the library's stream), `offtarget` (the opt-in genome-wide seed scan of
cfg 5 on the same resident genome, its own steps and roofline) and, at N = 1, `cpu_baseline` (the
reference-faithful numpy port on a bounded sample: one core, and all cores for BLAS).
"""
import argparse
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
N_SIMD = 1024          # 256 CUs x 4


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preheat-ms", type=float, default=60.0,
                    help="untimed steps go on after the W warm-up steps until this much time has passed since the first "
                         "of them (GPU clocks back at steady state); 0 = exactly W")
    ap.add_argument("--workload", default="switchgrass", choices=["switchgrass", "tair10", "ecoli", "sorghum"])
    ap.add_argument("--fasta", default=None, metavar="PATH",
                    help="a REAL genome instead of the synthetic stand-in (SURVEY.md 8d: \"real FASTA may be substituted on the GPU box "
                         "if present\"): read through cropsr_amd.fasta exactly as the CLI reads it, same arena builder; the line then "
                         "says data: real and names the file.  At N > 1 every rank's genome is this file")
    ap.add_argument("--gff", default=None, metavar="PATH",
                    help="with --fasta: the GFF3 the annotate block joins (default: the block is skipped for a real genome)")
    ap.add_argument("--phytozome", default=None, metavar="PATH", help="with --gff: the Phytozome annotation_info file")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N through the library's node handle (crp_node_*): ONE process, no launcher, no sockets -- the cut, "
                         "the fan-out over the N devices and the gatherv happen inside libcropsr_hip.so")
    ap.add_argument("--strong-only", action="store_true",
                    help="--single-process: only the strong-scaling block (ONE genome over the N devices, digest-checked); the line "
                         "is {\"strong\": ...}.  What rank 0 of a process-per-GPU run starts as its closing node block")
    ap.add_argument("--devices", default=None, metavar="LIST",
                    help="--single-process: the HIP devices to use, e.g. 0,1,2,3 (default 0..N-1; with --share-gpu0: device 0, N times)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the switchgrass-like genome (debug)")
    ap.add_argument("--geometry", default="auto", choices=["auto", "large", "small"],
                    help="tile shape of the scan (CRP_OPT_TILE_GEOMETRY); auto = by the arena's size: large for the >= 1 Gb genome")
    ap.add_argument("--two-pass", action="store_true",
                    help="count / tile-scan / emit launch sequence (CRP_OPT_TWO_PASS=1) instead of the default single launch")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the final RCCL gatherv altogether")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="N > 1: run the gatherv inside every timed step instead of once at the end")
    ap.add_argument("--share-gpu0", action="store_true",
                    help="rehearsal only: every rank uses device 0; fences and sums go over the control sockets and "
                         "the gatherv over the host transport (RCCL cannot put two ranks on one GPU)")
    ap.add_argument("--collective-timeout", type=float, default=300.0,
                    help="N > 1: rank 0 prints the line with the timed steps' numbers and aborts the run if the final gatherv "
                         "or the off-target block does not return within this many seconds (0 = wait for ever)")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="--gpus N without a launcher: stop the self-started ranks after this many seconds (kept well below "
                         "a 1500 s step time-out of whoever runs the bench, so that the launcher's own 124 path comes first)")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="N = 1: skip the host-to-host block that sends the same genome through crp_scan_stream (upload | scan | "
                         "fetch as a pipeline over slices) beside the serial upload / scan / fetch numbers of `pcie_inclusive`")
    ap.add_argument("--no-strong", action="store_true",
                    help="N > 1: skip the strong-scaling block (ONE genome cut over the N ranks, BASELINE.json configs[3], [4])")
    ap.add_argument("--no-node-block", action="store_true",
                    help="N > 1, one process per GPU: skip rank 0's closing block that drives the SAME N devices through the "
                         "library's single-process node handle (crp_node_*: ncclCommInitAll + one grouped send/recv, and device-to-"
                         "device copies) on the strong-scaling genome, while the other ranks wait")
    ap.add_argument("--strong-steps", type=int, default=0, help="timed steps of the strong-scaling block (0 = --steps)")
    ap.add_argument("--no-strong-check", action="store_true",
                    help="strong-scaling block: skip rank 0's N = 1 scan of the whole genome (the efficiency's denominator and "
                         "the digest the stitched tables are compared with)")
    ap.add_argument("--offtarget-steps", type=int, default=5,
                    help="timed steps of the off-target seed scan (0 = skip that block)")
    ap.add_argument("--annotate-steps", type=int, default=5,
                    help="N = 1: timed look-ups of the opt-in annotation join (crp_annotate_lookup) over the same resident hit "
                         "tables, with a seeded synthetic Phytozome-style GFF for the workload (0 = skip that block)")
    ap.add_argument("--annotate-genes", type=int, default=50000, help="gene models of that GFF at --scale 1")
    ap.add_argument("--offtarget-seeds-from-planes", action="store_true",
                    help="off-target block: the seed stage reads the planes itself (ot_seed_kernel) instead of taking the "
                         "seed words from the scan")
    ap.add_argument("--cpu-sample-bases", type=int, default=40000000,
                    help="upper bound on the bases of the same workload timed on the CPU port; the actual "
                         "sample is sized for about 12 s of CPU work per leg (0 = skip)")
    return ap.parse_args()


class FastaWorkload:
    """A real genome as a workload (--fasta): the contig table the reference builds from the file (cropsr_amd.fasta, the
    CLI's own loader: CROPSR.py:54-74 with cropsr_functions.py:190-229), contig strings as they are scanned."""

    class Spec:
        def __init__(self, name, length):
            self.name, self.length = name, length

    def __init__(self, path):
        import numpy as np
        from cropsr_amd import annotate, fasta
        table = fasta.load_bytes(path)
        self.path = path
        self._strings = [np.frombuffer(v, dtype=np.uint8) for _, v in table]
        self.keys = [k for k, _ in table]
        self.dec = 0 if (self.keys and self.keys[0].startswith(">")) else 1  # re-formatted path: one decoration character in front
        self.decoration = 4 * self.dec
        self.specs = [self.Spec(annotate.contig_name(k), max(0, int(v.size) - self.decoration)) for k, v in zip(self.keys, self._strings)]
        self.name = "%s (%d contigs, %d bases)" % (os.path.basename(path), len(self.specs), sum(s.length for s in self.specs))

    def contig_string(self, k):
        return self._strings[k]

    def string_length(self, k):
        return int(self._strings[k].size)


_FASTA_CACHE = {}


def make_workload(args, genome):
    """Genome number `genome` of the run's workload: the seeded stand-in (bench_workload.py), or --fasta's file."""
    import bench_workload as bw
    if args.fasta:
        if args.fasta not in _FASTA_CACHE:
            _FASTA_CACHE[args.fasta] = FastaWorkload(args.fasta)
        return _FASTA_CACHE[args.fasta]
    if args.workload == "switchgrass":
        wl = bw.switchgrass_like(genome, args.scale)
    elif args.workload == "tair10":
        wl = bw.tair10_like()
    elif args.workload == "sorghum":
        wl = bw.sorghum_like()
    else:
        wl = bw.ecoli_like()
    wl.string_length = lambda k, wl=wl: wl.specs[k].length + 4  # + decoration
    wl.decoration = 4
    return wl


def algorithmic_bytes(n_chars, hits, composition, n_contigs):
    """B_in = ceil(N / 4) for the 2-bit codes + 2 * ceil(N / 8) for the upper-case and acgt bit-planes --
    the planes "omit[ted] for inputs that are entirely uppercase ACGT, i.e. cfg 2" -- and B_out = 12 * H.  "Entirely
    upper-case ACGT" is measured, not assumed: crp_arena_composition counts the characters of the resident arena, and the
    only others allowed are the <= 4 decoration characters cropsr_functions.py:221-229 leaves around every contig.
    Returns (bytes, planes_counted)."""
    planes = composition is None or composition["n_other"] > 4 * n_contigs
    return (n_chars + 3) // 4 + (2 * ((n_chars + 7) // 8) if planes else 0) + 12 * hits, planes


def _cpu_leg(fp, sample_string, n_bases, threads, seconds):
    """One timed run of the reference-faithful port with `threads` BLAS threads."""
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=threads)
    except Exception:  # pragma: no cover
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        # size the sample for about `seconds` of CPU work: time a 200 kb probe first
        probe = min(200000, n_bases)
        t0 = time.perf_counter()
        fp.scan_score(sample_string[:probe + 1])
        rate = probe / (time.perf_counter() - t0)
        n_bases = int(min(n_bases, max(probe, seconds * rate)))
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        rows, scores = fp.scan_score(sample_string[:n_bases + 1])
        dt = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
    scored = int((scores != -1.0).sum())
    return {"value": scored / dt, "cores": threads, "bases": n_bases, "gRNAs": scored, "wall_s": dt,
            "user_s": ru1.ru_utime - ru0.ru_utime, "sys_s": ru1.ru_stime - ru0.ru_stime,
            "minor_faults": ru1.ru_minflt - ru0.ru_minflt}


def _host_cpu(n_cores):
    """The CPU the baseline ran on: model name and core count from /proc/cpuinfo, the threads this process may use."""
    model, logical = "unknown", 0
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    logical += 1
                    if model == "unknown":
                        model = line.split(":", 1)[1].strip()
    except OSError:
        pass
    return {"model": model, "logical_cpus": logical or (os.cpu_count() or 0), "usable_by_this_process": n_cores}


def cpu_baseline(sample_string, n_bases):
    """Reference-faithful numpy port (oracle/faithful_port.py) on one core, then with every host core
    available to BLAS -- the reference's only multi-threaded calls are its two np.matmul
    (CROPSR.py:305,311).  user/sys seconds and page faults are reported because the reference's
    scorer is page-fault bound (12 KB of fresh temporaries per gRNA): its speed depends on the host's
    memory system far more than on its cores."""
    from oracle import faithful_port as fp
    try:
        n_cores = len(os.sched_getaffinity(0))
    except AttributeError:
        n_cores = os.cpu_count() or 1
    one = _cpu_leg(fp, sample_string, n_bases, 1, 12.0)
    out = {"value": one["value"], "unit": "gRNAs/s", "cores": 1, "kind": "port",
           "sample": "first %d bases of contig 0 of the same workload, scan+score only (%d gRNAs in %.1f s; %.1f kb/s; "
                     "user %.1f s, sys %.1f s, %d minor page faults)"
                     % (one["bases"], one["gRNAs"], one["wall_s"], one["bases"] / one["wall_s"] / 1e3,
                        one["user_s"], one["sys_s"], one["minor_faults"]),
           "calibration": "profiles/cpu_calibration.json (real reference vs this port, development container)",
           # the baseline moved 14 x between the development container and the GPU host: say what it ran on
           "host": _host_cpu(n_cores)}
    if n_cores > 1:
        allc = _cpu_leg(fp, sample_string, n_bases, n_cores, 8.0)
        out["all_cores"] = {"value": allc["value"], "unit": "gRNAs/s", "cores": n_cores,
                            "sample": "first %d bases, %d gRNAs in %.1f s (user %.1f s, sys %.1f s)"
                                      % (allc["bases"], allc["gRNAs"], allc["wall_s"], allc["user_s"], allc["sys_s"])}
    return out


def load_profile_facts(build_id, workload):
    """HBM traffic and VALU instruction counts per launch of the emit kernel, from the committed
    rocprofv3 summary -- used only if it was measured on THIS build of the library and this workload."""
    import glob
    # profiles/traffic.json is the headline workload's; profiles/traffic_<config>.json the smaller configs' (collect.sh rNN <config>)
    paths = [os.path.join(ROOT, "profiles", "traffic.json")] + sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json")))
    why = "no profiles/traffic.json"
    for path in paths:
        if not os.path.exists(path):
            continue
        with open(path) as f:
            tj = json.load(f)
        name = "profiles/" + os.path.basename(path)
        if tj.get("workload") != workload or tj.get("kernel") != "emit_kernel":
            why = "no profiles/traffic*.json for this workload"
            continue
        if tj.get("build_id") != build_id:
            return None, "%s was measured on build %s, this is %s: stale, not used" % (name, tj.get("build_id"), build_id)
        return tj, tj.get("source", name)
    return None, why


def table_digests(per_contig):
    """sha256 over (positions, score bits) per strand of every contig's tables: one hex string per contig."""
    import hashlib
    import numpy as np
    out = []
    for h in per_contig:
        d = hashlib.sha256()
        for key in ("pos_plus", "score_plus", "pos_minus", "score_minus"):
            d.update(np.ascontiguousarray(h[key]).tobytes())
        out.append(d.hexdigest())
    return out


def strong_scaling_block(args, eng, group, use_rccl, fence, reduce):
    """BASELINE.json configs[3], [4] as written: ONE genome over the N GPUs of the node ("per-chromosome shard across
    8 x MI355X + RCCL gatherv").  The reference's contig loop (CROPSR.py:409) is what shards: parallel.strong_plan cuts
    every contig longer than a rank's fair share into pieces (128 characters of halo either side, hits owned by match
    position) and deals the pieces by LPT; every rank scans ITS share (no collective), then the one exchange of the path,
    crp_gather_hits, brings the tables to rank 0, which stitches them.  Timed: the scan (max over ranks) and the exchange,
    each fenced on both sides.  Rank 0 also scans the WHOLE genome alone (the N = 1 time the efficiency is quoted against)
    and compares the digest of its stitched tables with the digest of those N = 1 tables, contig by contig."""
    import numpy as np
    from cropsr_amd import parallel
    from cropsr_amd.engine import Hits
    rank, world = group.rank, group.world
    steps = args.strong_steps or args.steps
    check = not args.no_strong_check
    wl = make_workload(args, 0)
    lengths = [wl.string_length(k) for k in range(len(wl.specs))]
    plan = parallel.strong_plan(lengths, world)
    pieces, mine = plan["pieces"], plan["by_rank"][rank]
    wanted = {}
    for q in mine:
        wanted.setdefault(pieces[q][0], []).append(q)
    ref_builder = eng.arena_builder(lengths) if (rank == 0 and check) else None
    t_gen = time.perf_counter()
    views = {}
    for k in (range(len(lengths)) if ref_builder is not None else sorted(wanted)):
        s = wl.contig_string(k)
        if ref_builder is not None:
            ref_builder.add(s)
        for q in wanted.get(k, ()):
            v, shift = parallel.piece_view(s, pieces[q][1], pieces[q][2])
            views[q] = (np.array(v, dtype=np.uint8, copy=True), shift)  # (a copy: the contig itself is released)
        del s
    builder = eng.arena_builder([views[q][0].size for q in mine])
    for q in mine:
        builder.add(views[q][0])
    arena = builder.seal()
    layout = [(q, 0, int(arena.offsets[j]), int(arena.lengths[j])) for j, q in enumerate(mine)]
    del views
    t_gen = time.perf_counter() - t_gen

    def gatherv(n_plus, n_minus):
        """The exchange proper: on RCCL the tables land in rank 0's HBM (what is timed); their copy to rank 0's host for the
        stitch-and-digest check is fetch_gathered(), outside the timed region.  The host transport IS a copy to the host."""
        if use_rccl:
            return eng.gather_hits(arena, 0)
        cols = arena.fetch(n_plus, n_minus)
        return parallel.gather_host(group, [dict(zip(parallel.COLUMNS, (cols[0], cols[2], cols[3], cols[5])))], 0)

    def fetch_gathered(result):
        if not use_rccl or rank != 0:
            return result
        return [[eng.gathered_fetch(r, result)] for r in range(world)]

    n_plus, n_minus = arena.scan_score_device(20)
    scored = arena.count_scored()
    t_pre = time.perf_counter()
    n_warm = max(1, args.warmup)
    for _ in range(n_warm):
        arena.scan_score_device(20)
    spent = time.perf_counter() - t_pre
    extra = int(reduce([max(0.0, args.preheat_ms * 1e-3 - spent) / max(spent / n_warm, 1e-6)], "max")[0] + 0.999)
    for _ in range(extra):
        arena.scan_score_device(20)
    eng.profile(1)
    eng.profile_read(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        arena.scan_score_device(20)
    fence()
    dt_scan = reduce([time.perf_counter() - t0], "max")[0] / steps
    emit = eng.profile_read(reset=True)["emit_score"]
    eng.profile(0)
    mine_report = {"rank": rank, "pieces": len(mine), "bases": int(sum(pieces[q][2] - pieces[q][1] for q in mine)),
                   "characters_with_halos": int(arena.stats()["n_chars"]), "tiles": arena.tiles()["n_tiles"],
                   "tile_geometry": arena.tiles()["geometry"],
                   "kept_hits": int(n_plus + n_minus), "kernel_ms": emit["ms"] / max(1, emit["launches"])}
    per_rank = group.all_gather(mine_report)
    layouts = group.all_gather(layout)
    gatherv(n_plus, n_minus)  # warm-up: RCCL sets up its point-to-point channels
    fence()
    tg = time.perf_counter()
    gathered = gatherv(n_plus, n_minus)
    fence()
    dt_gather = reduce([time.perf_counter() - tg], "max")[0]
    bytes_to_root = (eng.gather_bytes() if use_rccl else int(getattr(group, "bytes_gathered", 0))) if rank == 0 else 0
    gathered = fetch_gathered(gathered)
    hits_all, scored_all = reduce([n_plus + n_minus, scored], "sum")
    out = {"workload": wl.name, "scaling": "strong", "genomes": 1, "steps": steps, "pieces": len(pieces),
           "contigs_cut": int(sum(1 for k in range(len(lengths)) if sum(1 for p in pieces if p[0] == k) > 1)),
           "halo": parallel.HALO, "ms_scan_max_rank": dt_scan * 1e3, "ms_gatherv": dt_gather * 1e3,
           "gatherv_transport": "RCCL (in-library)" if use_rccl else "host-socket",
           "bytes_to_root": bytes_to_root, "positions": "16 bits per hit + one word per 65 536 arena positions (CRP_GATHER_POS16)",
           "value": None, "unit": "gRNAs/s", "per_rank": per_rank, "setup_s": t_gen}
    # an owned hit is a hit of the whole contig; halo hits are counted by their owner only -- the unit is taken from the
    # stitched tables when the check runs, else from the sum over ranks minus nothing (halo hits are < 0.001 % of it)
    out["kept_hits_all_ranks_incl_halos"] = int(hits_all)
    out["value_scan_only"] = scored_all / dt_scan
    out["value"] = scored_all / (dt_scan + dt_gather)
    if rank == 0:
        per_piece = parallel.merge_gathered(gathered, layouts)
        stitched, q = [], 0
        for k in range(len(lengths)):
            parts = []
            while q < len(pieces) and pieces[q][0] == k:
                _, start, end = pieces[q]
                parts.append((start, end, start - max(0, start - parallel.HALO), per_piece[q]))
                q += 1
            stitched.append(parallel.stitch_pieces(parts))
        out["kept_hits"] = int(sum(h["pos_plus"].size + h["pos_minus"].size for h in stitched))
        n_scored = int(sum((h["score_plus"] != -1).sum() + (h["score_minus"] != -1).sum() for h in stitched))
        out["gRNAs_scored"] = n_scored
        out["value_scan_only"] = n_scored / dt_scan
        out["value"] = n_scored / (dt_scan + dt_gather)
        del gathered, per_piece
    if check:
        # the N = 1 scan of the same genome on rank 0's GPU (the other ranks wait at the fence)
        if rank == 0:
            ref = ref_builder.seal()
            rp, rm = ref.scan_score_device(20)
            for _ in range(n_warm + extra):
                ref.scan_score_device(20)
            eng.profile(1)
            eng.profile_read(reset=True)
            t0 = time.perf_counter()
            for _ in range(steps):
                ref.scan_score_device(20)
            t1 = (time.perf_counter() - t0) / steps
            e1 = eng.profile_read(reset=True)["emit_score"]
            eng.profile(0)
            whole = Hits(ref.offsets, ref.lengths, 20, ref.fetch(rp, rm))
            want = table_digests([whole.contig(k) for k in range(len(lengths))])
            got = table_digests(stitched)
            bad = [k for k in range(len(lengths)) if want[k] != got[k]]
            out["n1"] = {"ms_scan": t1 * 1e3, "kernel_ms": e1["ms"] / max(1, e1["launches"]), "kept_hits": int(rp + rm)}
            out["digest_ok"] = not bad
            if bad:
                out["digest_mismatch_contigs"] = bad[:10]
            out["speedup_vs_n1"] = t1 / (dt_scan + dt_gather)
            out["efficiency_vs_n1"] = t1 / (dt_scan + dt_gather) / world
            out["efficiency_vs_n1_scan_only"] = t1 / dt_scan / world
            ref.close()
        fence()
    arena.close()
    return out


def load_offtarget_traffic(build_id, workload):
    """Counter traffic of the off-target block per step and stage (tools/pmc_calibrate.sh -> profiles/offtarget_traffic.json),
    used only if it was measured on THIS build and workload."""
    path = os.path.join(ROOT, "profiles", "offtarget_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        tj = json.load(f)
    if tj.get("workload") != workload or tj.get("build_id") != build_id:
        return None
    return tj


def node_strong_block(args, node, devices, wl_name, one, n_warm, want=None, n1=None):
    """ONE genome (`one`: its contig strings) over the devices of `node` (crp_node_*): load (the library cuts it into
    contiguous equal shares with halos), timed scans, the gatherv with and without the 16-bit position packing -- and as
    device-to-device copies as well when the default transport was RCCL -- then the stitched tables' SHA-256 per contig
    against `want` (the digests of an N = 1 scan; computed here on devices[0] if not given)."""
    from cropsr_amd import Engine
    from cropsr_amd import _native as nat
    from cropsr_amd.engine import Hits
    world = node.size
    steps = args.strong_steps or args.steps

    def time_gather(**kw):
        node.gather(0, **kw)  # warm-up: buffers sized, RCCL's point-to-point channels set up
        reps = [node.gather(0, **kw) for _ in range(3)]
        best = min(reps, key=lambda r: r["ms_total"])
        return {"ms": best["ms_total"], "ms_exchange": best["ms_exchange"], "bytes_to_root": best["bytes_to_root"],
                "transport": best["transport"], "ms_all": [round(r["ms_total"], 4) for r in reps], "note": best["note"]}

    t_up = time.perf_counter()
    node.load(one)
    t_up = time.perf_counter() - t_up
    node.scan_score_device(20)
    for _ in range(n_warm):
        node.scan_score_device(20)
    node.profile(1)
    for k in range(world):
        node.profile_read(k, reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        node.scan_score_device(20)
    dt_scan = (time.perf_counter() - t0) / steps
    sprof = [node.profile_read(k, reset=True)["emit_score"] for k in range(world)]
    node.profile(0)
    g = time_gather(pos16=True)
    g_raw = time_gather(pos16=False)
    g_peer = time_gather(pos16=True, peer_copy=True) if g["transport"].startswith("RCCL") else None
    # To the HOST (where the reference's consumer, the CSV writer, lives): the gatherv into device 0 + one fetch over device 0's
    # PCIe link, against CRP_NODE_HOST_GATHER -- no exchange at all, every device's rows over that device's own link, side by
    # side, into the same host arrays.  Second pass each (the arrays' pages are touched, the staging buffers exist).
    to_host = {}
    hits = None
    for label, kw in (("gatherv_to_device_0_then_one_link", {}), ("every_device_over_its_own_link", {"to_host": True})):
        for rep in range(3):
            t0 = time.perf_counter()
            node.gather(0, **kw)
            t1 = time.perf_counter()
            hits = node.fetch(out=hits)  # (the same host arrays every time: their pages are touched after the first pass)
            t2 = time.perf_counter()
        to_host[label] = {"ms_gather": (t1 - t0) * 1e3, "ms_fetch": (t2 - t1) * 1e3, "ms": (t2 - t0) * 1e3}
    node.gather(0)
    dt_gather = g["ms"] * 1e-3
    n_scored = node.count_scored()
    hits = node.fetch()
    plan = node.plan()
    strong = {"workload": wl_name, "scaling": "strong", "genomes": 1, "steps": steps, "pieces": len(plan),
              "contigs_cut": len(plan) - len(one), "halo": nat.HALO, "devices": list(devices),
              "ms_scan_max_rank": dt_scan * 1e3, "ms_gatherv": g["ms"], "gatherv_transport": g["transport"],
              "bytes_to_root": g["bytes_to_root"], "gatherv": g, "gatherv_raw_u32_positions": g_raw, "tables_to_the_host": to_host,
              "kept_hits": hits.n_plus + hits.n_minus, "gRNAs_scored": int(n_scored), "unit": "gRNAs/s",
              "value_scan_only": n_scored / dt_scan, "value": n_scored / (dt_scan + dt_gather),
              "per_rank": [{"rank": k, "kernel_ms": sprof[k]["ms"] / max(1, sprof[k]["launches"]),
                            "characters_with_halos": (node.arena_stats(k) or {}).get("n_chars", 0),
                            "tiles": (node.arena_stats(k) or {}).get("n_tiles", 0)} for k in range(world)],
              "setup_s": t_up}
    if g_peer is not None:
        strong["gatherv_device_to_device_copies"] = g_peer
    if not args.no_strong_check:
        got = table_digests([hits.contig(k) for k in range(len(one))])
        if want is None:
            with Engine(devices[0]) as eng:
                builder = eng.arena_builder([s_.size for s_ in one])
                for s_ in one:
                    builder.add(s_)
                ref = builder.seal()
                rp, rm = ref.scan_score_device(20)
                for _ in range(n_warm):
                    ref.scan_score_device(20)
                eng.profile(1)
                eng.profile_read(reset=True)
                t0 = time.perf_counter()
                for _ in range(steps):
                    ref.scan_score_device(20)
                t1 = (time.perf_counter() - t0) / steps
                e1 = eng.profile_read(reset=True)["emit_score"]
                whole = Hits(ref.offsets, ref.lengths, 20, ref.fetch(rp, rm))
                want = table_digests([whole.contig(k) for k in range(len(one))])
                ref.close()
            n1 = {"ms_scan": t1 * 1e3, "kernel_ms": e1["ms"] / max(1, e1["launches"]), "kept_hits": int(rp + rm)}
        bad = [k for k in range(len(one)) if want[k] != got[k]]
        strong["n1"] = n1
        strong["digest_ok"] = not bad
        if bad:
            strong["digest_mismatch_contigs"] = bad[:10]
        if n1:
            t1 = n1["ms_scan"] * 1e-3
            strong["speedup_vs_n1"] = t1 / (dt_scan + dt_gather)
            strong["efficiency_vs_n1"] = t1 / (dt_scan + dt_gather) / world
            strong["efficiency_vs_n1_scan_only"] = t1 / dt_scan / world
    return strong


def main_single_process(args):
    """--gpus N --single-process: the same measurement through the library's node handle (crp_node_*) -- ONE
    process, no launcher, no sockets, no bootstrap.  Weak headline: N genomes (seeds 0..N-1) as ONE contig list, cut by the
    library into N contiguous equal shares; a step = crp_node_scan_score (the scan queued on every device, then
    collected).  The path's one exchange, crp_node_gather, runs after the timed steps and is reported on its own, with and
    without the 16-bit position packing, on RCCL and as device-to-device copies.  `strong`: ONE genome over the N devices,
    stitched tables compared contig by contig (SHA-256) with an N = 1 scan of the same genome on device 0."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from cropsr_amd import Engine, node as nd
    from cropsr_amd import _native as nat
    world = args.gpus
    if args.devices:
        devices = [int(d) for d in args.devices.split(",")]
        if len(devices) != world:
            raise SystemExit("--devices lists %d devices, --gpus says %d" % (len(devices), world))
    else:
        devices = [0] * world if args.share_gpu0 else list(range(world))
    node = nd.Node(devices)  # raises without libcropsr_hip.so / GPUs: no fallback
    node.configure(two_pass=True if args.two_pass else None, geometry=None if args.geometry == "auto" else args.geometry)
    build_id = nat.lib().crp_build_id().decode()
    info = node.device_info(0)
    is_real = bool(args.fasta)

    def strings_of(wl):
        return [wl.contig_string(k) for k in range(len(wl.specs))]

    if args.strong_only:
        wl = make_workload(args, 0)
        strong = node_strong_block(args, node, devices, wl.name, strings_of(wl), max(1, args.warmup))
        print(json.dumps({"strong": strong, "n_gpus": world, "config": {"library_build": build_id, "device": info["name"].strip()}}), flush=True)
        node.close()
        if strong.get("digest_ok") is False:
            sys.exit(1)
        return

    # ---- weak workload: `world` genomes, generated side by side (numpy releases the GIL), ONE contig list
    t_gen = time.perf_counter()
    genomes = [make_workload(args, g) for g in range(world)]
    if is_real:
        per_genome = [strings_of(genomes[0])] * world
    else:
        with ThreadPoolExecutor(max(1, min(world, (os.cpu_count() or 2) // 2))) as pool:
            per_genome = list(pool.map(strings_of, genomes))
    strings = [s for g in per_genome for s in g]
    bases_all = sum(spec.length for wl in genomes for spec in wl.specs)
    t_gen = time.perf_counter() - t_gen
    sample = bytes(strings[0][:args.cpu_sample_bases + 1]).decode("ascii", "replace") if strings else ""
    t_upload = time.perf_counter()
    node.load(strings)  # cut + one host thread per device: H2D + pack
    t_upload = time.perf_counter() - t_upload
    n_strings = len(strings)
    del strings, per_genome

    watch = {"timer": None, "fired": False, "lock": threading.Lock()}
    partial = {}

    def guard(stage, line_fn):
        """RCCL has no time-out: if the stage does not come back within --collective-timeout seconds, print the line with what
        has been measured so far and leave non-zero."""
        if world == 1 or args.collective_timeout <= 0:
            return

        def fire():
            with watch["lock"]:
                if watch["timer"] is None:
                    return
                watch["fired"] = True
            try:
                print(json.dumps(line_fn("%s did not return within %.0f s" % (stage, args.collective_timeout))), flush=True)
            finally:
                os._exit(1)
        watch["timer"] = threading.Timer(args.collective_timeout, fire)
        watch["timer"].daemon = True
        watch["timer"].start()

    def unguard():
        with watch["lock"]:
            if watch["timer"] is not None:
                watch["timer"].cancel()
                watch["timer"] = None
        if watch["fired"]:
            threading.Event().wait()

    # ---- the timed steps: no exchange inside
    node.scan_score_device(20)  # first scan: sizes the tables
    dev_counts = [node.device_counts(k) for k in range(world)]
    stats = [node.arena_stats(k) for k in range(world)]
    comp0 = node.arena_composition(0)
    t_pre = time.perf_counter()
    n_warm = max(1, args.warmup)
    for _ in range(n_warm):
        node.scan_score_device(20)
    spent = time.perf_counter() - t_pre
    extra = int(max(0.0, args.preheat_ms * 1e-3 - spent) / max(spent / n_warm, 1e-6) + 0.999)
    for _ in range(extra):
        node.scan_score_device(20)
    n_warm += extra
    node.profile(1)
    for k in range(world):
        node.profile_read(k, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        node.scan_score_device(20)  # returns when every device's stream has drained
    dt = time.perf_counter() - t0
    prof = [node.profile_read(k, reset=True)["emit_score"] for k in range(world)]
    node.profile(0)
    kernel_ms = [p["ms"] / max(1, p["launches"]) for p in prof]
    # units until the gather has counted them exactly: rows with a real score in every device's own tables (a hit inside a
    # halo is in two of them: < 0.001 %)
    scored_all = sum(c["n_scored"] for c in dev_counts if c)
    hits_all = sum(c["n_plus"] + c["n_minus"] for c in dev_counts if c)

    def build_line(gather_info=None, strong=None, error=None):
        c0, s0 = dev_counts[0], stats[0]
        hits0 = (c0["n_plus"] + c0["n_minus"]) if c0 else 0
        n_chars0 = s0["n_chars"] if s0 else 0
        algo_bytes, planes_counted = algorithmic_bytes(n_chars0, hits0, comp0, s0["n_texts"] if s0 else 0)
        achieved = algo_bytes / (kernel_ms[0] * 1e-3) / 1e9 if kernel_ms[0] else None
        facts, facts_src = load_profile_facts(build_id, genomes[0].name)
        line = {
            "metric": "gRNAs scored/sec", "value": partial.get("scored_all", scored_all) * args.steps / dt, "unit": "gRNAs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "untimed_steps_before": n_warm,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "real" if is_real else "synthetic",
            "config": {"workload": genomes[0].name, "genomes": world, "contigs_per_genome": len(genomes[0].specs),
                       "bases_total": int(bases_all), "kept_hits_total": int(partial.get("hits_all", hits_all)), "guide_len": 20,
                       "tile_geometry": s0["geometry"] if s0 else None, "tiles_per_launch": s0["n_tiles"] if s0 else 0,
                       "launches_per_step": 3 if args.two_pass else 1,
                       "parallelism": "ONE process, node handle (crp_node_*): %d contig strings cut by the library into %d contiguous "
                                      "equal shares (halo %d) over devices %s; gatherv to device %d once, after the steps"
                                      % (n_strings, world, nat.HALO, devices, devices[0]),
                       "device": info["name"].strip(), "library_build": build_id},
            "bases_per_s": bases_all * args.steps / dt,
            "per_rank": [{"rank": k, "device": devices[k], "kernel_ms": kernel_ms[k],
                          "characters_with_halos": stats[k]["n_chars"] if stats[k] else 0,
                          "tiles": stats[k]["n_tiles"] if stats[k] else 0,
                          "kept_hits_incl_halos": (dev_counts[k]["n_plus"] + dev_counts[k]["n_minus"]) if dev_counts[k] else 0}
                         for k in range(world)],
            "roofline": {"bound": "hbm", "kernel": "emit_kernel, single launch (masks + chained tile offsets + compact + score), device 0's",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved else None,
                         "traffic": facts.get("hbm_bytes_per_launch") if (facts and world == 1) else None,
                         "traffic_source": facts_src if world == 1 else "profiles/traffic.json is an N = 1 measurement",
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "algorithmic_bytes": "ceil(N/4) + 2*ceil(N/8) + 12*H" if planes_counted else "ceil(N/4) + 12*H",
                         "kernel_ms": kernel_ms[0]},
            "setup_s": {"generate": t_gen, "cut_upload_pack_all_devices": t_upload},
        }
        if gather_info is not None:
            line["gatherv_ok"] = "s" in gather_info
            if "s" in gather_info:
                line["value_with_final_gatherv"] = partial.get("scored_all", scored_all) / (dt / args.steps + gather_info["s"])
                line["ms_scan_plus_gatherv"] = (dt / args.steps + gather_info["s"]) * 1e3
            line["gatherv"] = gather_info
        if strong is not None:
            line["strong"] = strong
        if error:
            line["error"] = error
        return line

    def time_gather(**kw):
        node.gather(0, **kw)  # warm-up: buffers sized, RCCL's point-to-point channels set up
        reps = []
        for _ in range(3):
            st = node.gather(0, **kw)
            reps.append(st)
        best = min(reps, key=lambda r: r["ms_total"])
        return {"ms": best["ms_total"], "ms_exchange": best["ms_exchange"], "bytes_to_root": best["bytes_to_root"],
                "transport": best["transport"], "ms_all": [round(r["ms_total"], 4) for r in reps], "note": best["note"]}

    # ---- the path's one exchange
    gather_info = None
    if world > 1 and not args.no_gather:
        guard("crp_node_gather", lambda why: build_line({"error": why}))
        try:
            packed = time_gather(pos16=True)
            raw = time_gather(pos16=False)
            gather_info = {"s": packed["ms"] * 1e-3, "transport": packed["transport"], "bytes_to_root": packed["bytes_to_root"],
                           "GB_per_s_into_root": packed["bytes_to_root"] / (packed["ms_exchange"] * 1e-3) / 1e9 if packed["ms_exchange"] else None,
                           "positions": "16 bits per hit + one word per 65 536 arena positions (CRP_GATHER_POS16)",
                           "pos16": packed, "raw_u32_positions": raw}
            if packed["transport"].startswith("RCCL"):
                gather_info["device_to_device_copies"] = time_gather(pos16=True, peer_copy=True)
            node.gather(0)
            _, gp, gm = node.counts()
            partial["hits_all"] = gp + gm
            partial["scored_all"] = node.count_scored()
        except Exception as e:
            gather_info = {"error": repr(e)[:300]}
        unguard()
    elif world == 1:
        node.gather(0)
        _, gp, gm = node.counts()
        partial["hits_all"], partial["scored_all"] = gp + gm, node.count_scored()

    # ---- strong scaling: ONE genome over the N devices, checked against device 0 alone
    strong = None
    if world > 1 and not args.no_strong and not (gather_info and "error" in gather_info):
        guard("strong-scaling block", lambda why: build_line(gather_info, {"error": why}))
        try:
            strong = node_strong_block(args, node, devices, genomes[0].name, strings_of(genomes[0]), n_warm)
        except Exception as e:
            import traceback
            traceback.print_exc()
            strong = {"error": repr(e)[:300]}
        unguard()

    line = build_line(gather_info, strong)
    if world == 1 and args.cpu_sample_bases > 0:
        from oracle import oracle as _o
        _o.lib()
        line["cpu_baseline"] = cpu_baseline(sample, args.cpu_sample_bases)
        line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
    print(json.dumps(line), flush=True)
    failed = (bool(gather_info and "error" in gather_info) or
              bool(strong and ("error" in strong or strong.get("digest_ok") is False)))
    if failed:
        sys.stdout.flush()
        os._exit(1)
    node.close()


def main():
    args = parse_args()
    if args.single_process:
        return main_single_process(args)
    from cropsr_amd import launch
    if launch.wanted(args.gpus):
        # `python bench.py --gpus N` with no launcher in the environment: this process -- which has not
        # touched HIP or RCCL and never will -- starts the N ranks as fresh child processes of the same
        # command line, lets rank 0's JSON line through (inherited stdout) and leaves with a status that is 0
        # only if every rank's was (cropsr_amd/launch.py).  Under torch.distributed.run WORLD_SIZE is set
        # and this branch is not taken.
        sys.stdout.flush()
        sys.exit(launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                    timeout_s=args.launch_timeout))
    from cropsr_amd import Engine, parallel, rendezvous
    from cropsr_amd import _native as nat

    group = rendezvous.Group.from_env()
    rank = group.rank if group else 0
    world = group.world if group else 1
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    local_rank = 0 if args.share_gpu0 else (group.local_rank if group else 0)
    # (CROPSR_BENCH_FORCE_RCCL=1 with --share-gpu0 asks RCCL for a communicator of ranks that share a GPU, which it
    # refuses: a way to see, on a one-GPU box, what happens when the communicator cannot be created -- below)
    use_rccl = world > 1 and (not args.share_gpu0 or os.environ.get("CROPSR_BENCH_FORCE_RCCL") == "1")

    eng = Engine(local_rank)  # raises without libcropsr_hip.so / GPU: no fallback
    if args.two_pass:
        eng.configure(two_pass=True)
    if args.geometry != "auto":
        eng.configure(geometry=args.geometry)
    rccl_error = None
    if use_rccl:
        from cropsr_amd import rendezvous
        try:
            eng.comm_init(group)  # RCCL communicator inside the library; the id travels over the control sockets
        except rendezvous.RankError as e:
            # Every rank has the same error (comm_init agrees on it before anyone proceeds).  The scan itself needs no
            # collective, so the measurement goes on with the control sockets as fence and host transport for the
            # final exchange; the line says so (`rccl_error`, `parallelism`) and stderr says it loudly.
            rccl_error = str(e)[:500]
            if not eng.comm_stuck:  # (a bootstrap that never returned still holds the context on its helper thread)
                nat.lib().crp_comm_destroy(eng._ctx)
            use_rccl = False
            sys.stderr.write("[bench rank %d] RCCL communicator unavailable, falling back to the host transport "
                             "for fences and the final gatherv: %s\n" % (rank, rccl_error))

    # ---- workload: `world` genomes, contigs dealt to ranks by LPT (weak scaling)
    genomes = [make_workload(args, g) for g in range(world)]
    all_specs = [(g, k) for g in range(world) for k in range(len(genomes[g].specs))]
    lengths = [genomes[g].string_length(k) for g, k in all_specs]  # + decoration
    owner = parallel.partition_contigs(lengths, world)
    mine = [i for i, o in enumerate(owner) if o == rank]
    t_gen = time.perf_counter()
    builder = eng.arena_builder([lengths[i] for i in mine])
    sample = None
    my_bases = 0
    t_upload = 0.0
    keep_strings = world == 1 and not args.no_pipelined  # (the pipelined host-to-host block below sends them through again)
    kept_strings = []
    for i in mine:
        g, k = all_specs[i]
        s = genomes[g].contig_string(k)
        if sample is None and rank == 0:
            sample = bytes(s[:args.cpu_sample_bases + 1]).decode("ascii", "replace")  # keeps the leading quote
        t_up = time.perf_counter()
        builder.add(s)  # characters over PCIe + the ballot pack kernel, synchronous
        t_upload += time.perf_counter() - t_up
        my_bases += genomes[g].specs[k].length
        if keep_strings:
            kept_strings.append(s)
        del s
    t_up = time.perf_counter()
    arena = builder.seal()  # (flushes the small contigs the builder still holds, waits for the uploads)
    t_upload += time.perf_counter() - t_up
    t_gen = time.perf_counter() - t_gen

    def gatherv(n_plus, n_minus):
        if use_rccl:
            return eng.gather_hits(arena, 0)
        cols = arena.fetch(n_plus, n_minus)  # host transport (rehearsal): D2H, then the control sockets
        return parallel.gather_host(group, [dict(zip(("pos_plus", "score_plus", "pos_minus", "score_minus"),
                                                      (cols[0], cols[2], cols[3], cols[5])))], 0)

    want_gather = world > 1 and not args.no_gather

    def step():
        n_plus, n_minus = arena.scan_score_device(20, want_pre=False)
        if want_gather and args.gather_every_step:
            gatherv(n_plus, n_minus)
        return n_plus, n_minus

    def fence():
        nat.check(nat.lib().crp_synchronize(eng._ctx), "crp_synchronize", eng._ctx)
        if use_rccl:
            eng.comm_barrier()  # all-reduce on the library's stream + stream sync
        elif group:
            group.barrier()

    def reduce(values, op):
        if use_rccl:
            return eng.comm_allreduce(values, op)
        return group.allreduce(values, op) if group else [float(v) for v in values]

    n_plus, n_minus = step()  # first scan: sizes the tables (the W warm-up steps proper follow below)
    # units: kept hits that got a real score (a complete 30-window), counted on the GPU
    scored = arena.count_scored()
    t_fetch = time.perf_counter()
    t_fetch_again = None
    if rank == 0:
        host_cols = arena.fetch(n_plus, n_minus)  # D2H of the tables into FRESH pageable numpy arrays (outside the timed region)
        t_fetch = time.perf_counter() - t_fetch
        t_fetch_again = time.perf_counter()
        arena.fetch(n_plus, n_minus, out=host_cols)  # and again into the same arrays: what a caller that keeps its buffers pays
        t_fetch_again = time.perf_counter() - t_fetch_again
    else:
        t_fetch = time.perf_counter() - t_fetch
        host_cols = None
    # ---- the same boundary as a PIPELINE (crp_scan_stream, never `value`): the genome goes up in slices while the slice
    # before is scanned and the tables of the one before that come down -- host strings in, host tables out, one call
    pipelined = None
    if keep_strings and rank == 0:
        import numpy as np
        try:
            eng.stream_prepare()

            def run(out=None):
                h = eng.scan_stream(kept_strings, 20, out=out)
                return h, h.stream_stats
            fresh, pinned = [], []
            for rep in range(8):  # (the first call checks the tables; the clocks and the host's pages settle over the next few)
                h, st = run()
                if rep == 0:  # the same rows as the one-arena scan: counts, and the f64 column bit for bit (same order)
                    same = (h.n_plus == n_plus and h.n_minus == n_minus and
                            np.array_equal(h.score_plus.view(np.uint64), host_cols[2].view(np.uint64)) and
                            np.array_equal(h.score_minus.view(np.uint64), host_cols[5].view(np.uint64)))
                else:
                    fresh.append(st["wall_s"])
                stats_fresh = st
                del h
            t_pin = time.perf_counter()
            out = eng.empty_tables(n_plus, n_minus)
            t_pin = time.perf_counter() - t_pin
            # (the serial fetch into the same pinned tables: crp_fetch_hits by DMA, no staging copy)
            t_fetch_pinned = time.perf_counter()
            arena.fetch(n_plus, n_minus, out=[out[0], None, out[1], out[2], None, out[3]])
            t_fetch_pinned = time.perf_counter() - t_fetch_pinned
            for rep in range(8):
                h, st = run(out)
                if rep:
                    pinned.append(st["wall_s"])
                del h
            del out
            q_after = eng.query()
            up_bytes, down_bytes = int(sum(int(x.size) for x in kept_strings)), 12 * int(n_plus + n_minus)
            med = lambda v: sorted(v)[len(v) // 2]
            pipelined = {"pipelined_s": med(fresh), "pipelined_into_pinned_tables_s": med(pinned), "pipelined_best_s": min(fresh),
                         "pipelined_into_pinned_tables_best_s": min(pinned), "tables_equal_the_serial_scan": bool(same),
                         "runs_s": {"fresh_pageable_tables": fresh, "pinned_tables": pinned}, "pinning_the_tables_once_s": t_pin,
                         "fetch_tables_into_pinned_arrays_s": t_fetch_pinned,
                         "slices": int(stats_fresh["slices"]), "lanes": int(stats_fresh["lanes"]),
                         # (scans of the pipeline whose look-back timed out and were repeated with three launches: none expected)
                         "chain_timeouts_after": q_after["chain_timeouts"], "three_launch_mode_after": bool(q_after["two_pass_active"]),
                         "first_slice_on_host_s": stats_fresh["first_slice_on_host_s"],
                         "uploader_busy_s": stats_fresh["uploader_busy_s"], "copier_busy_s": stats_fresh["copier_busy_s"],
                         "bytes_up": up_bytes, "bytes_down": down_bytes,
                         # the link: 56 GB/s one way alone, 48 + 48 GB/s with both directions busy (profiles/microbench/duplex_copy.hip)
                         "link_floor_s": max(up_bytes, down_bytes) / 48e9, "link_floor_if_one_direction_at_a_time_s": (up_bytes + down_bytes) / 56e9}
            if not same:
                raise SystemExit("bench.py: crp_scan_stream's tables differ from the one-arena scan's")
        except SystemExit:
            raise
        except Exception as e:  # (reported, not fatal: the headline does not depend on it)
            pipelined = {"error": "%s: %s" % (type(e).__name__, e)}
    del host_cols
    kept_strings = []

    # count / scan kernel times from a few extra steps outside the timed region; inside it only the
    # emit+score kernel (the one the roofline is quoted on) is bracketed by HIP events
    eng.profile(2)
    eng.profile_read(reset=True)
    for _ in range(3):
        arena.scan_score_device(20, want_pre=False)
    side = eng.profile_read(reset=True)
    # the box's practical HBM ceiling, measured live: the count kernel of the three-launch mode is a pure
    # streaming read of the four planes (0.5 B per character, nothing written but 8 B per tile)
    stream_GBs = None
    three_launch_side = None  # the count / tile-scan / emit kernels of the three-launch mode, measured apart from the timed steps
    if not args.two_pass:
        eng.configure(two_pass=True)
        for _ in range(5):
            arena.scan_score_device(20, want_pre=False)
        three = eng.profile_read(reset=True)
        eng.configure(two_pass=False)  # (back to what the run was started with: this branch is the not --two-pass one)
        three_launch_side = {k: three[k]["ms"] / max(1, three[k]["launches"]) for k in ("count", "tile_scan", "emit_score")}
        cnt = three["count"]
        if cnt["launches"]:
            stream_GBs = (arena.stats()["n_chars"] / 2.0) / (cnt["ms"] / cnt["launches"] * 1e-3) / 1e9
    else:
        three_launch_side = {k: side[k]["ms"] / max(1, side[k]["launches"]) for k in ("count", "tile_scan", "emit_score")}
    eng.profile(0)
    # The W warm-up steps run HERE, right before the timed region: the side measurements above (a 0.6 GB
    # device-to-host copy among them) leave the GPU idle long enough for its clocks to drop, and a timed region
    # of 20 x 0.5 ms is over before they are back up (measured: 0.547 ms per kernel with 20 timed steps against
    # 0.489 with 200 on one box, whatever W had been before the copy).  --preheat-ms adds untimed steps until about
    # that much time has passed since the first warm-up step, so that a small W still ends at steady clocks
    # (`untimed_steps_before` in the line says how many ran in all).
    t_pre = time.perf_counter()
    n_warm = max(1, args.warmup)
    for _ in range(n_warm):
        step()
    spent = time.perf_counter() - t_pre
    extra = max(0.0, args.preheat_ms * 1e-3 - spent) / max(spent / n_warm, 1e-6)
    extra = int(reduce([extra], "max")[0] + 0.999)  # every rank runs the same number of steps (a step may hold a collective)
    for _ in range(extra):
        step()
    n_warm += extra
    eng.profile(1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_read(reset=True)
    eng.profile(0)
    state = eng.query()

    dt = reduce([dt], "max")[0]
    scored_all, bases_all, hits_all = reduce([scored, my_bases, n_plus + n_minus], "sum")
    # every rank's own view of the timed region, for the line (control sockets: a few numbers per rank)
    my_emit = prof["emit_score"]
    hbm_free, hbm_total = eng.hbm()
    mine_report = {"rank": rank, "device": local_rank, "kernel_ms": my_emit["ms"] / max(1, my_emit["launches"]),
                   "bases": int(my_bases), "kept_hits": int(n_plus + n_minus), "gRNAs_scored": int(scored),
                   # start-up: how long this rank waited for the group to form, and what the DEVICE holds now (all processes
                   # on it: with --share-gpu0 that is every rank's context, arena and tables together)
                   "rendezvous_s": round(group.connect_s, 3) if group else 0.0,
                   "device_hbm_in_use_GiB": round((hbm_total - hbm_free) / 2.0 ** 30, 3)}
    per_rank = group.all_gather(mine_report) if group else [mine_report]

    # ---- everything the line needs that the collectives below cannot change, taken NOW: if the exchange (or the
    # off-target block's all-reduce) never returns on this node, rank 0's watchdog still prints the scan's numbers
    info = eng.device_info() if rank == 0 else None
    build_id = nat.lib().crp_build_id().decode()
    n_chars = arena.stats()["n_chars"]
    tiles = arena.tiles()
    composition = arena.composition()  # upper-case ACGT vs everything else, counted on the GPU (which planes count)
    is_real = bool(args.fasta)

    def build_line(gather_info, ot, strong=None):
        hits = n_plus + n_minus
        algo_bytes, planes_counted = algorithmic_bytes(n_chars, hits, composition, len(mine))  # SURVEY.md 8d, rank 0's launch
        emit = prof["emit_score"]
        emit_ms = emit["ms"] / max(1, emit["launches"])
        achieved = algo_bytes / (emit_ms * 1e-3) / 1e9
        facts, facts_src = load_profile_facts(build_id, genomes[0].name)
        three_launches = bool(args.two_pass or state["two_pass_active"])
        # the kernels that ran inside the timed steps: one in the single-launch mode, three otherwise -- the
        # count and tile-scan times of the three-launch mode come from the side run and are reported under `three_launch`
        path_ms = emit_ms + ((three_launch_side["count"] + three_launch_side["tile_scan"]) if three_launches else 0.0)
        roof = {"bound": "hbm",
                "kernel": "emit_kernel (scan+compact+score)" if three_launches else
                          "emit_kernel, single launch (masks + chained tile offsets + compact + score)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": facts.get("hbm_bytes_per_launch") if facts else None, "traffic_source": facts_src,
                "algorithmic_bytes_per_launch": int(algo_bytes),
                "algorithmic_bytes": "ceil(N/4) + 2*ceil(N/8) + 12*H" if planes_counted else
                                     "ceil(N/4) + 12*H (entirely upper-case ACGT input: SURVEY.md 8d leaves the two bit-planes out)",
                "characters": {"N": int(n_chars), "upper_case_acgt": composition["n_plain"], "other": composition["n_other"]},
                # a pure streaming read of the same planes on this box (the count kernel): the practical ceiling beside the 8 TB/s spec
                "measured_stream_read_GBs": stream_GBs,
                "kernel_ms": emit_ms, "all_kernels_ms": path_ms,
                # CRP_OPT_TWO_PASS = 1's kernels on the same arena, from a side run outside the timed region
                "three_launch": {"count_kernel_ms": three_launch_side["count"], "tile_scan_ms": three_launch_side["tile_scan"],
                                 "emit_kernel_ms": three_launch_side["emit_score"]}}
        if facts and facts.get("valu_insts_per_launch") and facts.get("kernel_cycles_per_launch"):
            # the second, honest ceiling: the kernel is VALU-issue bound (one wave64 VALU instruction
            # holds its SIMD for 4 cycles on average here): issue slots used / issue slots there were
            roof["valu_issue_frac"] = facts["valu_insts_per_launch"] * 4.0 / N_SIMD / facts["kernel_cycles_per_launch"]
            roof["valu_insts_per_launch"] = facts["valu_insts_per_launch"]
            # the time those instructions alone take at the part's 2.4 GHz peak shader clock: no schedule of THIS
            # instruction stream is faster
            roof["valu_floor_ms"] = facts["valu_insts_per_launch"] * 4.0 / N_SIMD / 2.4e6
            roof["valu_floor_clock_GHz"] = 2.4
        line = {
            "metric": "gRNAs scored/sec", "value": scored_all * args.steps / dt, "unit": "gRNAs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "untimed_steps_before": n_warm,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "real" if is_real else "synthetic",
            "config": {"workload": genomes[0].name, "genomes": world, "contigs_per_genome": len(genomes[0].specs),
                       "bases_total": int(bases_all), "kept_hits_total": int(hits_all),
                       "guide_len": 20,
                       # rank 0's arena: tile shape (picked by the arena's size unless --geometry) and workgroups per launch
                       "tile_geometry": tiles["geometry"], "tiles_per_launch": tiles["n_tiles"], "tile_words": tiles["tile_words"],
                       "launches_per_step": 3 if three_launches else 1,
                       # single-launch scans that timed out in a look-back and were repeated as three launches
                       "chain_timeouts": state["chain_timeouts"],
                       "parallelism": ("contigs by LPT over %d ranks" % world) +
                       ("" if not want_gather else (" + %s gatherv to rank 0 " % ("RCCL (in-library)" if use_rccl else "host-socket") +
                                                    ("every step" if args.gather_every_step else "once, after the steps"))),
                       "device": info["name"].strip(), "library_build": build_id},
            "bases_per_s": bases_all * args.steps / dt,
            "per_rank": per_rank,
            "roofline": roof,
            "setup_s": {"generate_pack_upload": t_gen},
            # host-buffer boundary: characters H2D + pack, one scan, tables D2H (never `value`)
            "pcie_inclusive": {"upload_pack_s": t_upload, "fetch_tables_s": t_fetch,
                               # the same fetch into the SAME host arrays (pages already touched): the link's rate
                               "fetch_tables_into_reused_arrays_s": t_fetch_again,
                               "gRNAs_per_s": scored / (t_upload + dt / args.steps + t_fetch)},
        }
        if pipelined is not None:
            line["pcie_inclusive"].update(pipelined)
            if "pipelined_s" in pipelined:
                line["pcie_inclusive"]["gRNAs_per_s_pipelined"] = scored / pipelined["pipelined_s"]
        if rccl_error:
            line["rccl_error"] = rccl_error
        if strong is not None:
            line["strong"] = strong
        if ot is not None:
            tj = load_offtarget_traffic(build_id, genomes[0].name) if "roofline" in ot else None
            if tj:
                # counter traffic per step and stage; the FETCH_SIZE correction is calibrated for this block's access
                # patterns (streams, random 16-byte gathers: every line that reaches the fabric is tallied at half its size)
                r = ot["roofline"]
                r["traffic"] = tj.get("traffic_bytes_per_step")
                r["stage_traffic"] = tj.get("stage_traffic_bytes_per_step")
                r["stage_traffic_over_algorithmic"] = tj.get("stage_traffic_over_algorithmic")
                r["traffic_source"] = "profiles/offtarget_traffic.json: " + tj.get("correction", "")
                if r.get("traffic") and ot.get("kernels_ms"):
                    k = ot["kernels_ms"]
                    ms = k["ot_seed"] + k["ot_ball"] + k["ot_lookup"]
                    # what the memory system actually moves per second (whole 128-byte lines for every 16-byte gather)
                    r["traffic_GBs"] = r["traffic"] / (ms * 1e-3) / 1e9
                    if r["stage_traffic"].get("lookup_gather") and k["ot_lookup"]:
                        r["lookup_gather_traffic_GBs"] = r["stage_traffic"]["lookup_gather"] / (k["ot_lookup"] * 1e-3) / 1e9
            line["offtarget"] = ot
        if gather_info is not None:
            line["gatherv_ok"] = "s" in gather_info
            if "s" in gather_info:
                # bytes that crossed to rank 0, as the transport counted them: 10 B per hit with the 16-bit position packing
                # (CRP_GATHER_POS16; 12 B raw) + one word per 65 536 arena positions and table
                moved = float(gather_info.get("bytes_to_root", 12.0 * (hits_all - hits)))
                gather_info.update({"bytes_to_root": int(moved), "bytes_if_raw_u32_positions": int(12.0 * (hits_all - hits)),
                                    "GB_per_s_into_root": moved / gather_info["s"] / 1e9})
                # the WHOLE job of the path at N ranks = one scan on every rank + the one exchange: this, not
                # `value` (whose timed steps hold no collective and therefore grow ~N-fold by construction),
                # is the number to build a scaling curve from
                line["value_with_final_gatherv"] = scored_all / (dt / args.steps + gather_info["s"])
                line["ms_scan_plus_gatherv"] = (dt / args.steps + gather_info["s"]) * 1e3
            line["gatherv"] = gather_info
        return line

    # Rank 0's watchdog over the collectives that follow (RCCL has no time-out of its own): if one of them does not return
    # within --collective-timeout seconds, the line is printed with what the timed steps measured and the run is taken
    # down through the abort channel (non-zero exit on every rank) -- an unattended first multi-GPU run leaves data behind
    # even if the node's RCCL cannot complete a point-to-point exchange.
    import threading
    watch = {"stage": None, "timer": None, "lock": threading.Lock(), "fired": False}
    gather_info = strong = None  # (what the watchdog prints for the stages that have not run yet)

    def _timed_out():
        with watch["lock"]:  # the collective may return at the very moment the timer fires: one of the two prints
            if watch["timer"] is None:
                return
            watch["fired"] = True
        stage = watch["stage"]
        why = "%s did not return within %.0f s" % (stage, args.collective_timeout)
        gi = {"error": why} if stage == "final gatherv" else gather_info
        si = {"error": why} if stage == "strong-scaling block" else strong
        oi = {"error": why} if stage.startswith("off-target") else None
        try:
            print(json.dumps(build_line(gi, oi, si)), flush=True)
        finally:
            group.abort("bench: " + why)

    def guard(stage):
        if rank == 0 and world > 1 and args.collective_timeout > 0:
            watch["stage"] = stage
            watch["timer"] = threading.Timer(args.collective_timeout, _timed_out)
            watch["timer"].daemon = True
            watch["timer"].start()

    def unguard():
        with watch["lock"]:
            if watch["timer"] is not None:
                watch["timer"].cancel()
                watch["timer"] = None
        if watch["fired"]:  # the watchdog is printing the line and taking the run down: stay out of its way
            threading.Event().wait()

    # the final exchange, once, timed on its own (barrier + sync on both sides, max over ranks)
    gather_info = None
    if want_gather and not args.gather_every_step:
        guard("final gatherv")
        try:
            if os.environ.get("CROPSR_BENCH_TEST_STALL") == "%d:gatherv" % rank:  # tests: this rank never reaches the exchange
                time.sleep(3600)
            gatherv(n_plus, n_minus)  # warm-up: RCCL sets up its point-to-point channels
            fence()
            tg = time.perf_counter()
            gatherv(n_plus, n_minus)
            fence()
            tg = time.perf_counter() - tg
            gather_info = {"s": tg}  # rank 0 finishes last: it waits for every receive
            if rank == 0:
                gather_info["bytes_to_root"] = eng.gather_bytes() if use_rccl else int(getattr(group, "bytes_gathered", 0))
                gather_info["positions"] = "16 bits per hit + one word per 65 536 arena positions (CRP_GATHER_POS16)"
        except Exception as e:  # the bench line is printed even if the exchange fails on this node
            gather_info = {"error": repr(e)[:300]}
        unguard()

    # ---- strong scaling: ONE genome over the N ranks (the headline above stays weak scaling)
    if world > 1 and not args.no_strong and not (gather_info and "error" in gather_info):
        guard("strong-scaling block")
        try:
            strong = strong_scaling_block(args, eng, group, use_rccl, fence, reduce)
        except rendezvous.RankError:
            raise
        except Exception as e:
            import traceback
            traceback.print_exc()
            strong = {"error": repr(e)[:300]}
        unguard()

    # ---- the same N devices through the library's single-process node handle (crp_node_*), on the strong
    # block's genome.  Rank 0 starts a CHILD PROCESS -- `bench.py --gpus N --single-process --strong-only`: one fresh process
    # that opens a node over all N devices (its peers here hold their own contexts on them and wait at the barrier below),
    # cuts, scans, and runs the gatherv on RCCL-in-one-process (ncclCommInitAll, one grouped send/recv) and as device-to-
    # device copies, digest-checked against its own N = 1 scan -- and puts the child's `strong` block into the line.  A child,
    # not a thread: whatever goes wrong in there (a crash inside RCCL, a bootstrap that never returns: the child is killed
    # after --collective-timeout) cannot take this rank or its line down.
    node_block = None
    if world > 1 and not args.no_node_block and not args.no_strong:
        if rank == 0:
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--single-process", "--strong-only",
                   "--steps", str(args.strong_steps or args.steps), "--warmup", str(args.warmup), "--workload", args.workload,
                   "--scale", str(args.scale), "--geometry", args.geometry, "--cpu-sample-bases", "0",
                   "--collective-timeout", str(args.collective_timeout)]
            cmd += ["--share-gpu0"] if args.share_gpu0 else []
            cmd += ["--two-pass"] if args.two_pass else []
            cmd += ["--no-strong-check"] if args.no_strong_check else []
            cmd += ["--fasta", args.fasta] if args.fasta else []
            env = {k: v for k, v in os.environ.items()
                   if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR",
                                "MASTER_PORT", "CROPSR_LAUNCHED", "CROPSR_RDZV_ENDPOINT", "TORCHELASTIC_RUN_ID")}
            if args.collective_timeout > 0:
                # the library bounds its own RCCL waits (crp_node.cpp): a bootstrap or a group that never comes back is given up
                # inside the child, which then finishes on device-to-device copies and SAYS so (`note`) -- well inside the limit
                # after which the child would be killed with nothing to show
                env.setdefault("CRP_NODE_COMM_INIT_TIMEOUT_S", "%.0f" % max(10.0, 0.4 * args.collective_timeout))
                env.setdefault("CRP_NODE_COLLECTIVE_TIMEOUT_S", "%.0f" % max(10.0, 0.3 * args.collective_timeout))
            t_child = time.perf_counter()
            try:
                limit = (args.collective_timeout + 120.0) if args.collective_timeout > 0 else None
                p = subprocess.run(cmd, capture_output=True, text=True, timeout=limit, env=env, cwd=ROOT)
                lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                if lines:
                    node_block = json.loads(lines[-1]).get("strong") or {"error": "the child printed no strong block"}
                else:
                    node_block = {"error": "the child printed no line (status %d): %s" % (p.returncode, p.stderr[-300:])}
                if p.returncode != 0 and "error" not in node_block and node_block.get("digest_ok") is not False:
                    node_block["child_status"] = p.returncode
            except subprocess.TimeoutExpired:
                node_block = {"error": "the child did not finish within %.0f s and was killed" % limit}
            except Exception as e:
                node_block = {"error": repr(e)[:300]}
            node_block["wall_s_incl_process_start"] = round(time.perf_counter() - t_child, 3)
        group.barrier()

    # ---- the opt-in off-target seed scan of cfg 5 on the same resident genome and hit tables
    ot = None
    if args.offtarget_steps > 0 and rccl_error and world > 1:
        ot = {"skipped": "the site histogram is summed over the ranks by an RCCL all-reduce, and RCCL is unavailable here"}
    elif args.offtarget_steps > 0 and not (gather_info and "error" in gather_info):
        guard("off-target block (site-histogram all-reduce)")
        try:
            # The scan hands the seed words over (CRP_SCAN_SEEDS: the emit kernel writes them from the windows it
            # holds anyway); what that costs the scan is measured here and reported beside the step.
            eng.profile(1)
            eng.profile_read(reset=True)
            for _ in range(8):
                arena.scan_score_device(20, want_pre=False, want_seeds=not args.offtarget_seeds_from_planes)
            scan_seeds = eng.profile_read(reset=True)["emit_score"]
            eng.profile(0)
            scan_seeds_ms = scan_seeds["ms"] / max(1, scan_seeds["launches"])

            def ot_step():
                eng.offtarget_reset()
                sites = arena.offtarget_add(20)
                if use_rccl:
                    eng.offtarget_reduce()  # 64 MiB all-reduce of the site histogram over xGMI
                eng.offtarget_solve()
                arena.offtarget_counts(n_plus, n_minus, fetch=False)
                return sites
            sites = ot_step()
            for _ in range(4):  # (buffers sized, clocks up)
                ot_step()
            eng.profile(2)
            eng.profile_read(reset=True)
            fence()
            t1 = time.perf_counter()
            for _ in range(args.offtarget_steps):
                ot_step()
            fence()
            dt_ot = reduce([time.perf_counter() - t1], "max")[0]
            pot = eng.profile_read(reset=True)
            eng.profile(0)
            sites_all = reduce([sites], "sum")[0]
            per = {k: pot[k]["ms"] / max(1, pot[k]["launches"]) for k in ("ot_seed", "ot_ball", "ot_lookup", "ot_reduce")}
            hits = n_plus + n_minus
            # Algorithmic bytes of one step on this rank.  Seed stage: the seed words in (4 B per
            # hit; with --offtarget-seeds-from-planes instead the planes once, 0.5 B per character, + 4 B of position) and
            # the Morton codes out (4 B per hit); partition level 1: 4 B in per hit, 4 B out per site; level 2: 4 B in,
            # 2 B out per site; bucket histograms: 2 B in per site + 64 MiB in and out.  Ball passes: 64 MiB in,
            # 3 x 256 MiB out, 2 x 256 MiB back in.  Look-up: 4 B seed + 16 B gather + 16 B out per hit.
            n_chars_r = arena.stats()["n_chars"]
            seed_in = (n_chars_r // 2 + 4 * hits) if args.offtarget_seeds_from_planes else 4 * hits
            seed_bytes = seed_in + 4 * hits + 4 * hits + sites * (4 + 4 + 2 + 2) + 2 * (1 << 26)
            ball_bytes = (1 << 24) * (4 + 16 * 5)
            look_bytes = hits * (4 + 16 + 16)
            step_ms = dt_ot / args.offtarget_steps * 1e3
            kernels_ms = per["ot_seed"] + per["ot_ball"] + per["ot_lookup"]
            frac = lambda nbytes, ms: (nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms else None
            total_bytes = seed_bytes + ball_bytes + look_bytes
            ot = {"metric": "guides off-target-scanned/sec", "value": hits_all * args.offtarget_steps / dt_ot,
                  "unit": "guides/s", "steps": args.offtarget_steps, "ms_per_step": step_ms,
                  "sites_total": int(sites_all), "seed_len": 12, "max_mismatches": 3,
                  "kernels_ms": per,
                  "seeds": "from the planes (ot_seed_kernel)" if args.offtarget_seeds_from_planes else
                           "from the scan (CRP_SCAN_SEEDS): the emit kernel writes a seed word per hit",
                  # what the hand-over costs the scan's own kernel (per scan, not part of ms_per_step)
                  "scan_kernel_ms_with_seed_words": scan_seeds_ms,
                  "scan_kernel_ms_plain": prof["emit_score"]["ms"] / max(1, prof["emit_score"]["launches"]),
                  # the WHOLE step against the HBM roofline, and every stage beside it: the look-up (one random 16-byte
                  # gather per hit out of a 256 MiB table) is the dominant kernel and runs at the memory system's
                  # gather rate, not at its streaming rate
                  "roofline": {"bound": "hbm", "kernel": "whole step: seeds + partition + histogram, 3 ball passes, look-up",
                               "achieved": total_bytes / (kernels_ms * 1e-3) / 1e9 if kernels_ms else None,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac(total_bytes, kernels_ms),
                               "traffic": None, "algorithmic_bytes_per_step": int(total_bytes),
                               "stage_frac": {"seed_partition_histogram": frac(seed_bytes, per["ot_seed"]),
                                              "ball_passes": frac(ball_bytes, per["ot_ball"]),
                                              "lookup_gather": frac(look_bytes, per["ot_lookup"])},
                               "stage_bytes": {"seed_partition_histogram": int(seed_bytes), "ball_passes": int(ball_bytes),
                                               "lookup_gather": int(look_bytes)}},
                  "parity": "unpinned: the reference has no off-target step (oracle: oracle/crp_oracle.c all-pairs)"}
        except Exception as e:
            ot = {"error": repr(e)[:300]}
        unguard()

    # ---- the opt-in annotation join (BASELINE.json configs[2], [3] name a GFF) on the same resident tables
    ann_block = None
    if args.annotate_steps > 0 and world == 1 and (not is_real or args.gff):
        try:
            import shutil
            import tempfile
            import bench_workload as bw
            from cropsr_amd import annotate
            tmp = tempfile.mkdtemp(prefix="cropsr_bench_gff_")
            try:
                if is_real:  # the user's own annotation beside the user's own genome
                    gff, info_path = args.gff, args.phytozome
                    gff_rows = (None, None)
                else:
                    gff, info_path = os.path.join(tmp, "genes.gff3"), os.path.join(tmp, "annotation_info.txt")
                    n_genes = max(50, int(args.annotate_genes * (args.scale if args.workload == "switchgrass" else 1.0)))
                    gff_rows = bw.synthetic_annotation(genomes[0], gff, info_path, n_genes=n_genes)
                t_b = time.perf_counter()
                ann = annotate.Annotation(gff, info_path)
                t_b = time.perf_counter() - t_b
                gff_bytes = os.path.getsize(gff)
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
            names = [genomes[all_specs[i][0]].specs[all_specs[i][1]].name for i in mine]
            req = annotate.Request(ann, names, getattr(genomes[0], "dec", 1))
            t_t = time.perf_counter()
            points, ids = req.track([(j, int(arena.offsets[j]), int(arena.lengths[j])) for j in range(len(mine))])
            arena.annotate_set_track(points, ids)
            t_t = time.perf_counter() - t_t
            arena.scan_score_device(20, want_pre=False)  # (the look-up needs the tables of the arena's LAST scan)
            for _ in range(3):
                arena.annotate_lookup(n_plus, n_minus, fetch=False)
            eng.profile(2)
            eng.profile_read(reset=True)
            t_l = time.perf_counter()
            for _ in range(args.annotate_steps):
                arena.annotate_lookup(n_plus, n_minus, fetch=False)
            t_l = (time.perf_counter() - t_l) / args.annotate_steps
            pa = eng.profile_read(reset=True)["annotate"]
            eng.profile(0)
            feat = arena.annotate_lookup(n_plus, n_minus)
            import numpy as _np
            n_feat = int((feat[0] != annotate.NO_FEATURE).sum() + (feat[1] != annotate.NO_FEATURE).sum())
            k_ms = pa["ms"] / max(1, pa["launches"])
            algo = 16.0 * (n_plus + n_minus)  # 4 B position + 8 B score in, 4 B label-set id out per hit
            ann_block = {"metric": "hits annotated/sec", "value": (n_plus + n_minus) / t_l, "unit": "hits/s",
                         "steps": args.annotate_steps, "ms_per_lookup": t_l * 1e3, "kernel_ms": k_ms,
                         "gff": {"gene_rows": gff_rows[0], "cds_rows": gff_rows[1], "bytes": gff_bytes,
                                 "data": os.path.basename(args.gff) if is_real else "synthetic, seeded"},
                         "label_sets": len(ann.strings), "track_points": int(points.size), "hits_with_a_feature": n_feat,
                         "host_build_s": t_b, "track_layout_upload_s": t_t,
                         "roofline": {"bound": "hbm", "kernel": "annot_lookup_kernel (both tables, one launch)",
                                      "achieved": algo / (k_ms * 1e-3) / 1e9 if k_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": (algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms else None,
                                      "algorithmic_bytes_per_launch": int(algo), "traffic": None, "traffic_source": None},
                         "parity": "unpinned: the reference parses the GFF and never joins it (oracle: oracle/annotate_oracle.py)"}
            ann.close()
            # counter traffic of the look-up kernel per launch (tools/pmc_calibrate.sh -> profiles/offtarget_traffic.json), used
            # only if it was measured on THIS build, this workload and this block's own tables (same algorithmic bytes)
            tj = load_offtarget_traffic(build_id, genomes[0].name)
            if tj and tj.get("annotate_lookup_traffic_bytes_per_launch") and \
                    tj.get("annotate_lookup_algorithmic_bytes_per_launch") == int(algo):
                r = ann_block["roofline"]
                r["traffic"] = tj["annotate_lookup_traffic_bytes_per_launch"]
                r["traffic_over_algorithmic"] = r["traffic"] / algo
                r["traffic_source"] = "profiles/offtarget_traffic.json: " + tj.get("correction", "")
        except Exception as e:
            import traceback
            traceback.print_exc()
            ann_block = {"error": repr(e)[:300]}

    if rank == 0:
        line = build_line(gather_info, ot, strong)
        if ann_block is not None:
            line["annotate"] = ann_block
        if node_block is not None:
            line["single_process_node"] = node_block
        if world == 1 and args.cpu_sample_bases > 0:
            from oracle import oracle as _o
            _o.lib()
            line["cpu_baseline"] = cpu_baseline(sample, args.cpu_sample_bases)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)

    failed = (bool(gather_info and "error" in gather_info) or bool(ot and "error" in ot) or
              bool(strong and ("error" in strong or strong.get("digest_ok") is False)) or
              bool(node_block and node_block.get("digest_ok") is False))  # (a node block that could not RUN is reported, not fatal)
    if failed:
        # the communicator is in an unknown state after a failed exchange: no collective teardown,
        # and the run must not be recorded as a clean success
        sys.stdout.flush()
        os._exit(1)
    arena.close()
    eng.close()
    if group:
        group.close()
    from cropsr_amd import engine as _engine
    _engine.leave_if_comm_stuck(0)  # (a communicator bootstrap that never returned left a thread inside RCCL)


def _leave(status):
    """Every error exit of a process that may have opened the GPU: if a communicator bootstrap never returned
    (Engine.comm_init), a helper thread still sits inside RCCL and a normal tear-down may wait for it -- os._exit then."""
    eng_mod = sys.modules.get("cropsr_amd.engine")
    if eng_mod is not None:
        eng_mod.leave_if_comm_stuck(status)
    sys.exit(status)


if __name__ == "__main__":
    try:
        main()
    except Exception as e:  # a rank that fails alone must not leave its peers in a collective
        import traceback
        traceback.print_exc()
        from cropsr_amd import rendezvous
        g = getattr(rendezvous, "LAST_GROUP", None)
        if g is not None and g.world > 1:
            if isinstance(e, rendezvous.RankError):  # every rank has this error: leave together
                g.close()
                _leave(1)
            g.abort("%s: %s" % (type(e).__name__, e))
        _leave(1)
