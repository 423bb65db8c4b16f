"""`python -m cropsr_amd` -- the CROPSR command line on top of the MI355X engine.

Same flags, defaults, stdout lines, time.txt and CSV bytes as the reference's
CROPSR.py (v1.11b); the PAM scan and the scoring run in libcropsr_hip.so
instead of re/str.replace/numpy.  Reference anchors are given per step.

Differences that are NOT observable in the outputs:
  * all contigs are uploaded and scanned in ONE arena pass before the per-contig
    output loop starts (the reference scans inside the loop, CROPSR.py:413-434);
  * the 5 s sleep per contig (CROPSR.py:478) is kept only with --reference-sleep.
"""
import argparse
import sys
import time
from multiprocessing import cpu_count

import numpy as np

from . import fasta, rows

__version__ = "1.11b"  # the CROPSR version whose behaviour is reproduced

BANNER = r"""
################################################################################
##                                                                            ##
##                                                                            ##
##          .o88b.   d8888b.    .d88b.    d8888b.   .d8888.   d8888b.         ##
##         d8P  Y8   88  `8D   .8P  Y8.   88  `8D   88'  YP   88  `8D         ##
##         8P        88oobY'   88    88   88oodD'   `8bo.     88oobY'         ##
##         8b        88`8b     88    88   88ººº       `Y8b.   88`8b           ##
##         Y8b  d8   88 `88.   `8b  d8'   88        db   8D   88 `88.         ##
##          `Y88P'   88   YD    `Y88P'    88        `8888Y'   88   YD         ##
##                                                                            ##
##                                                                            ##
################################################################################
U.S. Dept. of Energy's Center for Advanced Bioenergy and Bioproducts Innovation
University of Illinois at Urbana-Champaign
"""


def build_parser():
    """The reference's flags (CROPSR.py:24-49) plus engine options that default
    to reference behaviour."""
    p = argparse.ArgumentParser(prog="CROPSR.py")
    p.add_argument("-f", "--fasta", metavar="", required=True, dest="f",
                   help="[required] path to input file in FASTA format")
    p.add_argument("-g", "--gff", metavar="", dest="g", help="path to input file in GFF format")
    p.add_argument("-p", "--phytozome", metavar="", dest="p", default=None,
                   help="path to input annotation info file in TXT format, default = None")
    p.add_argument("-o", "--output", metavar="", dest="o", default="data.csv",
                   help="path to output file, default = data.csv")
    p.add_argument("-l", "--length", metavar="", dest="l", type=int, default=20,
                   help="length of the gRNA se3quence, default = 20")
    p.add_argument("-L", "--flanking", metavar="", dest="L", type=int, default=200,
                   help="length of flanking region for verification, default = 200")
    p.add_argument("--cas9", action="store_true",
                   help="specifies that design will be made for the Cas9 CRISPR system")
    p.add_argument("-v", "--verbose", action="store_true",
                   help="prints visual indicators for each iteration")
    eng = p.add_argument_group("MI355X engine")
    eng.add_argument("--device", type=int, default=None,
                     help="HIP device index, default = 0 (LOCAL_RANK under torch.distributed.run)")
    eng.add_argument("--gpus", type=int, default=1,
                     help="run on this many GPUs of the node, one process per GPU: contigs (cut where longer than a "
                          "GPU's share) are scanned in parallel and the hit tables gathered to the first one over RCCL; "
                          "the ranks are started here unless a launcher (torch.distributed.run) already did")
    eng.add_argument("--devices", default=None, metavar="LIST",
                     help="run on these HIP devices from THIS one process, e.g. 0,1,2,3: the genome is cut into equal "
                          "contiguous shares, every device scans its own and the hit tables are gathered to the first one "
                          "(RCCL) -- all inside the library (its node handle); no launcher, no ranks.  Same CSV bytes")
    eng.add_argument("--seed", type=int, default=None,
                     help="seed numpy's global RNG so crispr_id is reproducible (reference: unseeded)")
    eng.add_argument("--csv-writer", choices=["native", "python"], default="native",
                     help="native: multi-threaded C++ row formatter (same bytes); python: csv module like the reference")
    eng.add_argument("--each-contig-once", action="store_true",
                     help="NOT reference behaviour: write every contig's rows once instead of re-writing all "
                          "earlier contigs on every pass (the reference never clears Complete_dataset, "
                          "CROPSR.py:407, which makes multi-contig genomes quadratic)")
    eng.add_argument("--reference-sleep", action="store_true",
                     help="also reproduce the reference's 5 s pause per contig")
    eng.add_argument("--score-finalize", choices=["gpu", "host"], default="gpu",
                     help="gpu (default): on_site_score comes from the GPU, whose exp is glibc's -- the bytes the "
                          "reference prints on a host without AVX-512; host: the GPU delivers the pre-sigmoid sum "
                          "(CROPSR.py:312) and THIS host's numpy applies 1/(1+np.exp(.)) (CROPSR.py:313), so the CSV "
                          "equals what the reference prints on this very host (numpy's exp differs in the last bit "
                          "between CPU families)")
    eng.add_argument("--offtarget", action="store_true",
                     help="NOT in the reference: genome-wide off-target seed scan; appends four columns "
                          "offtarget_seed_mm0..3 (other PAM-adjacent sites whose 12-nt PAM-proximal seed differs in "
                          "0..3 places; -1 where the guide has no 12-base seed)")
    eng.add_argument("--annotate", action="store_true",
                     help="NOT in the reference (which parses the GFF and drops it, CROPSR.py:375): fill the "
                          "`features` column with the gene/CDS rows of the GFF (-g) that contain the cut site, "
                          "named through the Phytozome annotation_info file (-p) when given")
    eng.add_argument("--bench-json", metavar="PATH", default=None,
                     help="write stage timings of this run (read, upload+scan, fetch, format+write) as one JSON object")
    return p


def import_gff_file(gff, verbose, out=sys.stdout):
    """Side effects of CROPSR.py:77-95.  The table is parsed exactly as the
    reference parses it and, exactly as there, never used."""
    import pandas as pd
    start_index = 0
    with open(gff, "r") as raw:
        if verbose:
            print(f"Annotation file {gff} successfully imported", file=out)
        lines = raw.readlines()
        for index, line in enumerate(lines):
            if "##" not in line:
                start_index = index
                break
    cols = ["chromosome", "source", "feature", "start", "end", "score", "strand", "phase", "attributes"]
    table = pd.read_csv(gff, sep="\t", skiprows=start_index, header=None, names=cols)
    if verbose:
        print("Annotation database successfully generated", file=out)
    return table


class EngineResident:
    """Hit tables of this rank's texts, still in HBM (parallel.sharded_scan's Resident)."""

    def __init__(self, backend, genome, counts, guide_len, want_pre):
        self.backend, self.genome, self.counts = backend, genome, counts
        self.guide_len, self.want_pre = guide_len, want_pre
        self.layout = []
        for k in range(genome.n_contigs):
            a, j = genome._where[k]
            self.layout.append((a, int(genome.arenas[a].offsets[j]), int(genome.arenas[a].lengths[j])))
        self._ot = False
        self._annotated = False

    def annotate(self, request):
        """The annotation join over this rank's resident tables (annotate.Request for ITS texts); the ids stay in HBM
        and travel with the tables in gather()."""
        self.genome.annotate(request, self.counts, fetch=False)
        self.backend.last_annotate_s = self.genome.annotate_s
        self._annotated = True

    def offtarget(self, group, own_by_arena):
        eng = self.backend.engine
        eng.offtarget_reset()
        for a, arena in enumerate(self.genome.arenas):
            own = own_by_arena[a] if a < len(own_by_arena) else []
            arena.offtarget_add(self.guide_len, np.asarray(own, dtype=np.uint64).reshape(-1, 2))
        if self.backend.transport == "rccl":
            eng.offtarget_reduce()  # the histogram is summed over the ranks on xGMI
        else:  # ranks sharing a GPU: sum the (sparse) histograms over the control sockets
            h = eng.offtarget_hist()
            idx = np.flatnonzero(h)
            total = np.zeros_like(h)
            for i, v in group.all_gather((idx, h[idx])):
                np.add.at(total, i, v)
            eng.offtarget_hist(total)
        eng.offtarget_solve()
        for arena, (n_plus, n_minus) in zip(self.genome.arenas, self.counts):
            arena.offtarget_counts(n_plus, n_minus, fetch=False)
        self._ot = True

    def _host_cols(self, arena, n_plus, n_minus, offtarget, features):
        c = arena.fetch(n_plus, n_minus, self.want_pre)
        cols = {"pos_plus": c[0], "score_plus": c[1] if self.want_pre else c[2],
                "pos_minus": c[3], "score_minus": c[4] if self.want_pre else c[5]}
        if offtarget:
            cols["ot_plus"], cols["ot_minus"] = arena.offtarget_counts(n_plus, n_minus)
        if features:
            cols["feat_plus"], cols["feat_minus"] = arena.annotate_lookup(n_plus, n_minus)
        return cols

    def gather(self, group, dst, offtarget, features=False):
        """[rank][arena] -> column dict on dst.  With want_pre the f64 column carries the pre-sigmoid
        sum (the root finalises the score on its host)."""
        from . import _native as nat
        from . import parallel
        eng = self.backend.engine
        if self.backend.transport != "rccl":
            mine = [self._host_cols(a, n[0], n[1], offtarget, features) for a, n in zip(self.genome.arenas, self.counts)]
            return parallel.gather_host(group, mine, dst, offtarget, features)
        n_arenas = group.all_gather(len(self.genome.arenas))
        out = [[] for _ in range(group.world)]
        for rnd in range(max(n_arenas)):
            arena = self.genome.arenas[rnd] if rnd < len(self.genome.arenas) else None
            err, counts = None, None
            try:
                counts = eng.gather_hits(arena, dst, offtarget, pre=self.want_pre, features=features)
            except nat.CropsrHipError as e:
                # crp_gather_hits agrees on what can fail on one rank BEFORE the tables move (a rank without tables, a
                # root that cannot size its receive buffers): every rank is back from the same call, so all of them
                # reach check() below and raise the same RankError.  Anything else (an RCCL or HIP error inside the
                # exchange) is this rank's alone, its peers may be blocked in ncclSend/Recv: take the run down NOW
                # (abort channel; exits with status 3) instead of waiting for them in check().
                if e.status not in nat.AGREED_GATHER_ERRORS:
                    group.abort("%s: %s" % (type(e).__name__, e))
                err = "%s: %s" % (type(e).__name__, e)
            group.check(err)
            if group.rank == dst:
                for r in range(group.world):
                    if rnd < n_arenas[r]:
                        out[r].append(eng.gathered_fetch(r, counts, offtarget, features))
        return out if group.rank == dst else None

    def release(self):
        self.genome.close()


class EngineBackend:
    """The product's hit provider: libcropsr_hip.so on one MI355X."""

    def __init__(self, device=0, group=None, finalize="gpu"):
        import os
        from .engine import Engine
        self.engine = Engine(device)
        self.finalize = finalize
        self.group = group
        # rccl: the tables cross xGMI inside the library; host: every rank copies its tables over its
        # own PCIe link and the control sockets carry them (ranks sharing one GPU, where RCCL cannot run)
        self.transport = os.environ.get("CROPSR_GATHER", "rccl")
        self.last_annotate_s = None  # --bench-json: seconds the last annotation join took on this rank
        self.last_stream = None      # --bench-json: crp_scan_stream's own numbers for the last plain scan

    def connect(self):
        """Collective over the group: create the RCCL communicator (transport "rccl")."""
        if self.group is not None and self.transport == "rccl":
            from .rendezvous import RankError
            try:
                self.engine.comm_init(self.group)
            except RankError as e:
                # every rank has this error (comm_init agrees on it): the job of the run is the CSV, and the tables can
                # travel over the control sockets instead
                import sys
                if self.group.rank == 0:
                    sys.stderr.write("cropsr_amd: no RCCL communicator (%s); the final exchange uses the host transport\n" % e)
                self.transport = "host"

    def _finalize(self, hits):
        """--score-finalize=host: CROPSR.py:313 on this host's numpy, from the GPU's pre-sigmoid sum."""
        for strand in ("plus", "minus"):
            pre = hits.pop("pre_" + strand, None)
            if pre is not None and self.finalize == "host":
                hits["score_" + strand] = host_sigmoid(pre)
        return hits

    def warm_up(self):
        """On the start-up helper thread, while the FASTA is still being read: the lanes of the pipelined scan (two further
        streams, four slice arenas with their tables, the landing buffers)."""
        import os
        if self.group is None and os.environ.get("CROPSR_STREAM", "1") != "0":
            self.engine.stream_prepare()
            # and one scan of a few hundred characters through them: a process's first launch of the scan kernels loads their
            # code object (17-25 ms, otherwise paid by the genome's first slice)
            self.engine.scan_stream([np.frombuffer(b"ACGTTGCAAGGCCTTA" * 40, dtype=np.uint8)], 20)

    def scan(self, contig_strings, guide_len, offtarget=False, annotation=None):
        """One pass on the GPU for all contig strings (seam 1 + 2).  The plain scan goes through crp_scan_stream -- upload,
        scan and table fetch as a pipeline over slices of the genome, the host link busy in both directions (the reference's
        loop is produce-and-consume per contig too, CROPSR.py:409-474); CROPSR_STREAM=0, or the opt-in steps that work on
        resident tables (offtarget, annotation), take the arena calls: upload, one scan, fetch.  annotation
        (annotate.Request): the hit dicts also carry feat_plus / feat_minus, the label-set id of every row, joined on the GPU
        while the tables are resident."""
        import os
        if not offtarget and annotation is None and os.environ.get("CROPSR_STREAM", "1") != "0":
            want_pre = self.finalize == "host"
            hits = self.engine.scan_stream(contig_strings, guide_len, want_pre=want_pre)
            self.last_stream = hits.stream_stats  # (--bench-json)
            out = []
            for k in range(len(contig_strings)):
                h = hits.contig(k)
                if want_pre:  # the f64 column is the pre-sigmoid sum: CROPSR.py:313 on this host's numpy
                    h["score_plus"], h["score_minus"] = host_sigmoid(h["score_plus"]), host_sigmoid(h["score_minus"])
                out.append(h)
            self.last_annotate_s = None
            return out
        genome = self.engine.genome(contig_strings)  # as many arenas as the genome needs
        hits = genome.scan_score(guide_len, want_pre=self.finalize == "host", offtarget=offtarget, annotation=annotation)
        out = [self._finalize(hits.contig(k)) for k in range(len(contig_strings))]
        self.last_annotate_s = genome.annotate_s
        genome.close()
        return out

    def scan_resident(self, texts, guide_len, offtarget=False):
        """The same scan with the tables left in HBM, for parallel.sharded_scan.  offtarget: the off-target step
        follows, so the scan also writes the seed words it works on."""
        want_pre = self.finalize == "host"
        genome = self.engine.genome(texts)
        counts = [a.scan_score_device(guide_len, want_pre, want_seeds=offtarget) for a in genome.arenas]
        return EngineResident(self, genome, counts, guide_len, want_pre)

    def finalize_gathered(self, all_hits):
        """Root, after a sharded scan with --score-finalize=host: the gathered f64 column is `pre`."""
        if self.finalize == "host":
            for h in all_hits:
                for strand in ("plus", "minus"):
                    h["score_" + strand] = host_sigmoid(h["score_" + strand])
        return all_hits

    def rescore(self, rows_u8, order):
        """Seam 2 on a few rows in one of the BLAS tail orders (rows.Dataset.rows)."""
        pre, score = self.engine.score_30mers(rows_u8, order)
        return host_sigmoid(pre) if self.finalize == "host" else score

    def close(self):
        self.engine.close()


class NodeBackend:
    """The same hit provider over SEVERAL MI355X in ONE process: the library's node handle (crp_node_*, node.Node).  The cut
    of the genome into contiguous equal shares, the uploads, the scans on every device, the opt-in off-target scan and
    annotation join and the gatherv to the first device all happen inside libcropsr_hip.so; this class only calls them.
    What scan() returns is what EngineBackend.scan() returns on one GPU, bit for bit."""

    def __init__(self, devices, finalize="gpu"):
        from .node import Node
        self.node = Node(devices)
        self.finalize = finalize
        self.last_annotate_s = None
        self.last_gather = None  # node.gather_stats() of the last scan (--bench-json)
        self._engine = None      # seam 2 on a few rows (rescore): a context of its own on the first device, opened when needed
        self._devices = list(devices)

    def scan(self, contig_strings, guide_len, offtarget=False, annotation=None):
        want_pre = self.finalize == "host"
        node = self.node
        node.load(contig_strings)
        node.scan_score_device(guide_len, want_pre=want_pre, want_seeds=offtarget)
        if offtarget:
            node.offtarget(guide_len)
        if annotation is not None:
            t0 = time.perf_counter()
            node.annotate(annotation)
            self.last_annotate_s = time.perf_counter() - t0
        # CROPSR_GATHER=host (as for the process-per-GPU mode): the consumer is this host's CSV writer, so nothing needs to
        # cross xGMI -- every device's rows come over its own PCIe link (CRP_NODE_HOST_GATHER); default: the gatherv to device 0
        import os
        to_host = os.environ.get("CROPSR_GATHER", "rccl") == "host"
        self.last_gather = node.gather(0, pre=want_pre, offtarget=offtarget, features=annotation is not None, to_host=to_host)
        hits = node.fetch(guide_len)
        out = []
        for k in range(len(contig_strings)):
            h = hits.contig(k)
            if want_pre:  # the f64 column that travelled is the pre-sigmoid sum: CROPSR.py:313 on this host's numpy
                h["score_plus"], h["score_minus"] = host_sigmoid(h["score_plus"]), host_sigmoid(h["score_minus"])
            out.append(h)
        return out

    def rescore(self, rows_u8, order):
        if self._engine is None:
            from .engine import Engine
            self._engine = Engine(self._devices[0])
        pre, score = self._engine.score_30mers(rows_u8, order)
        return host_sigmoid(pre) if self.finalize == "host" else score

    def close(self):
        if self._engine is not None:
            self._engine.close()
        self.node.close()


def host_sigmoid(pre):
    """`1/(1+np.exp(score))` of CROPSR.py:313 -- the reference's own expression, evaluated by the numpy
    of the host this runs on (so its last bit is this host's, like the reference's)."""
    with np.errstate(over="ignore"):
        return 1 / (1 + np.exp(np.asarray(pre, dtype=np.float64)))


def _distributed():
    """The rendezvous.Group of a multi-process launch (python -m torch.distributed.run ... -m cropsr_amd
    or any launcher that exports RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT), else None.  One process
    per GPU; no PyTorch: the control plane is rendezvous.py, the data plane RCCL inside the library."""
    global _ACTIVE_GROUP
    from . import rendezvous
    _ACTIVE_GROUP = rendezvous.Group.from_env()
    return _ACTIVE_GROUP


_ACTIVE_GROUP = None  # the group main() aborts when this rank fails outside an agreed error (Group.check)


NATIVE_GUIDE_LENGTHS = (1, 50)  # what the kernels' window logic and the native row formatter are built for
BATCH_ROWS = 4_000_000  # --each-contig-once: consecutive passes are formatted together until they hold this many rows (CROPSR_BATCH_ROWS)


def device_guide_length(l):
    """The guide length the engine scans with for the reference's `-l l` (any integer, CROPSR.py:38-40).  Inside
    0..50 it is l itself; outside, the nearest end of that range, whose kept hits are a SUPERSET of the reference's
    (the keep-filters of CROPSR.py:419 / :430 only get stricter as |l| grows) -- refilter_hits() then applies
    the literal filter.  No row is scored at those lengths (long_sequence has l + 10 characters, or is cut by the end
    of the string to more than 30 for l > 35): what remains of the reference's work there is the filter."""
    return min(max(int(l), 0), NATIVE_GUIDE_LENGTHS[1])


def refilter_hits(hits, n, l):
    """CROPSR.py:419 and :430, all four clauses each, on the hit tables of one contig string of n characters (match
    indices i of (?=.GG), j of (?=CC.)) -- for guide lengths the engine scanned with a clamped length
    (device_guide_length).  Every surviving row is unscored (-1), like the reference writes it."""
    out = dict(hits)
    for strand in ("plus", "minus"):
        p = np.asarray(hits["pos_" + strand]).astype(np.int64)
        a, b = (p - l, p) if strand == "plus" else (p + 3, p + 3 + l)  # pam_location (:418 / :429)
        keep = (a >= 5) & (a + 5 <= n + 10) & (b >= 5) & (b <= n + 10)
        for key in ("pos_", "score_", "pre_", "ot_", "feat_"):
            col = hits.get(key + strand)
            if col is not None:
                out[key + strand] = np.asarray(col)[keep]
        out["score_" + strand] = np.full(int(keep.sum()), -1.0)
    return out


class _Early:
    """fn() on a helper thread; get() joins and returns its result or raises what it raised."""

    def __init__(self, fn, start=True):
        """start=False: nothing runs ahead; get() calls fn itself (for work that is only worth starting when the
        cheap checks of the command line have passed)."""
        import threading
        self._out = []
        self._fn = fn
        self._thread = threading.Thread(target=self._run, args=(fn,), name="cropsr-early") if start else None
        if start:
            self._thread.start()

    def _run(self, fn):
        try:
            self._out.append((fn(), None))
        except BaseException as e:  # handed to the caller of get()
            self._out.append((None, e))

    def get(self):
        if self._thread is None:
            if not self._out:
                self._run(self._fn)
        else:
            self._thread.join()
        value, error = self._out[0]
        if error is not None:
            raise error
        return value


def run(args, backend=None, out=sys.stdout, group=None):
    """main() of the reference (CROPSR.py:333-486) with the hot path swapped out.

    `backend` provides scan(contig_strings, guide_len) -> [hits dict per contig]
    and rescore(rows_u8, order) -> scores; it defaults to the HIP engine.  The CPU
    test-suite injects the oracle here to pin this host logic against the golden
    CSVs without a GPU.  `group`: the rendezvous.Group of a multi-process run
    (default: from the launcher's environment).
    """
    import os
    begin = time.time()
    if not args.cas9:
        sys.exit("Please select at least one CRISPR system: Cas9")  # CROPSR.py:335-336
    offtarget = bool(getattr(args, "offtarget", False))
    l_dev = device_guide_length(args.l)
    if offtarget and not NATIVE_GUIDE_LENGTHS[0] <= args.l <= NATIVE_GUIDE_LENGTHS[1]:  # (before any side effect, on every rank)
        sys.exit("cropsr_amd: --offtarget needs a guide length between %d and %d (got %d)" % (NATIVE_GUIDE_LENGTHS + (args.l,)))
    finalize = getattr(args, "score_finalize", "gpu")
    stages = {}  # --bench-json
    own_group = group is None
    if own_group:
        group = _distributed()
    max_piece = int(os.environ.get("CROPSR_DIST_MAX_PIECE", "0")) or None
    if getattr(args, "devices", None) and group is not None and group.world > 1:
        # an external launcher started several ranks AND the command line asks for one process over several devices: every
        # rank would open every listed GPU, and only rank 0 would have a use for them
        sys.exit("cropsr_amd: --devices (one process over several GPUs) cannot run under a launcher that started %d ranks "
                 "(one process per GPU): give one of them" % group.world)

    def make_backend():
        if getattr(args, "devices", None):
            return NodeBackend([int(d) for d in str(args.devices).split(",")], finalize)
        device = getattr(args, "device", None)
        if device is None:
            device = group.local_rank if group is not None else 0
        b = EngineBackend(device, group, finalize)
        try:  # (a genome of a few slices at least: below that there is nothing to prepare, one small lane is made on the spot)
            big = isinstance(args.f, str) and os.path.getsize(args.f) >= (256 << 20)
        except OSError:
            big = False
        if big and not offtarget and not getattr(args, "annotate", False):
            b.warm_up()
        return b

    # Opening the GPU (HIP start-up, code object, two pinned staging buffers: ~0.3 s) starts NOW on a helper thread and
    # is collected where the backend is first needed: it overlaps reading and parsing the FASTA.
    # (not for a FASTA that is not there: that run ends with the reference's own exception and needs no GPU)
    readable = isinstance(args.f, str) and os.path.isfile(args.f)
    early = _Early(make_backend, start=readable) if backend is None else None
    # --annotate: the GFF (+ annotation_info) is parsed natively (crp_annotation_build releases the GIL) beside the FASTA read
    annotating = bool(getattr(args, "annotate", False))
    def build_annotation():
        from . import annotate
        t0 = time.perf_counter()
        a = annotate.Annotation(args.g, args.p)
        stages["annotation_build_s"] = time.perf_counter() - t0  # (on its own thread, beside the FASTA read and the GPU start-up)
        return a

    early_annot = _Early(build_annotation, start=readable) if annotating else None

    def annotation_request(data, table):
        """annotate.Request for the contig strings of `table` (every rank builds the same one)."""
        from . import annotate
        formatted = 2 * fasta.count_byte(data, b">") != fasta.count_byte(data, b"\n") + 1  # CROPSR.py:61: dec = 1
        # (the Annotation itself is collected from its helper thread when the join first needs it: after the scan)
        return annotate.Request(early_annot.get, [annotate.contig_name(k) for k, _ in table], 1 if formatted else 0)

    if group is not None and group.rank != 0:
        # Ranks other than 0 of a multi-GPU run: read the same FASTA, scan their share of the contigs,
        # hand the tables to rank 0 -- which alone prints, draws ids and writes files -- and leave.
        # Whatever fails here is reported to every rank (group.check inside sharded_scan).
        from . import parallel
        own_backend = backend is None
        err, strings, request = None, [], None
        try:
            data = fasta.read_text_bytes(args.f)
            table = fasta.table_from_bytes(data)
            strings = [v for _, v in table]
            if annotating:
                request = annotation_request(data, table)
            del data, table
            if own_backend:
                backend = early.get()
        except Exception as e:
            err = "%s: %s" % (type(e).__name__, e)
        try:
            group.check(err)
            if hasattr(backend, "connect"):
                backend.connect()
            parallel.sharded_scan(backend, strings, l_dev, group, max_piece=max_piece, offtarget=offtarget, annotation=request)
        finally:
            if own_backend and backend is not None:
                backend.close()
            if own_group:
                group.close()
        return
    verbose = args.verbose
    if verbose:
        print(BANNER + f"""
        You are currently utilizing the following settings:

        CROPSR version:                                 {__version__}
        Path to genome file in FASTA format:            {args.f}
        Path to output file:                            {args.o}
        Length of the gRNA sequence:                    {args.l}
        Length of flanking region for verification:     {args.L}
        Number of available CPUs:                       {cpu_count()}
        Path to annotation file in GFF format:          {args.g}
        Path to annotation_info file in TXT format:     {args.p}
        Designing for CRISPR system:
            Streptococcus pyogenes Cas9                 {args.cas9}
        """, file=out)

    timing = open("time.txt", "w")  # CROPSR.py:371 (CWD side effect, kept)
    # CROPSR.py:375 reads the GFF into a DataFrame it never uses (pandas: 0.2 s to import, ~1 s per 40 MB of GFF).  Its
    # observable behaviour -- two -v messages, the exceptions of a missing or malformed file -- is kept, at the place the
    # reference has it; the work itself starts NOW on a helper thread, beside the FASTA read and the scan
    import io as _io
    gff_messages = _io.StringIO()

    def reference_gff_import():
        t0 = time.perf_counter()
        try:
            return import_gff_file(args.g, verbose, gff_messages)
        finally:
            stages["reference_gff_import_s"] = time.perf_counter() - t0  # (thread time; what the run waits for: ..._wait_s)

    early_gff = _Early(reference_gff_import)

    # CROPSR.py:374, 54-74 -- read as text mode would (universal newlines), kept as bytes
    t_stage = time.perf_counter()
    data = fasta.read_text_bytes(args.f)
    if verbose:
        print(f"Genome file {args.f} successfully imported", file=out)
        if 2 * fasta.count_byte(data, b">") != fasta.count_byte(data, b"\n") + 1:
            print("formatting genome", file=out)
            print(f"Genome file {args.f} successfully formatted", file=out)
    table = fasta.table_from_bytes(data)  # == fasta.contig_table(text).items(), without printing the genome
    stages["read_fasta_s"] = time.perf_counter() - t_stage
    request, request_err = None, None
    if annotating:
        try:
            request = annotation_request(data, table)
        except Exception as e:  # (single process: raised below, after the reference's own GFF import had its say)
            request_err = e
    del data
    if verbose:
        print("The genome was successfully converted to a dictionary", file=out)
    t_wait = time.perf_counter()
    try:
        early_gff.get()  # CROPSR.py:375 (raises like the reference if -g is missing or unreadable)
    finally:
        out.write(gff_messages.getvalue())
        stages["reference_gff_import_wait_s"] = time.perf_counter() - t_wait

    if verbose:
        # (the blank line in the middle carries the reference's 12 blanks of indentation)
        print("\n            Initiating PAM site detection.\n            \n"
              "            Please wait, this may take a while...\n            ", file=out)

    if getattr(args, "seed", None) is not None:
        np.random.seed(args.seed)

    rows.write_header(args.o, offtarget=offtarget)  # CROPSR.py:402-405

    names = [k for k, _ in table]
    strings = [v for _, v in table]  # contig strings as bytes, one byte per character
    own_backend = backend is None
    t_stage = time.perf_counter()
    if group is None:
        if own_backend:
            backend = early.get()
        if request_err is not None:
            raise request_err
        extra = dict(offtarget=True) if offtarget else {}
        if request is not None:
            extra["annotation"] = request
        all_hits = backend.scan(strings, l_dev, **extra)
    else:  # contigs (cut where longer than a rank's share) over all GPUs, tables gathered here
        from . import parallel
        err = None
        try:
            if request_err is not None:
                raise request_err
            if own_backend:
                backend = early.get()
        except Exception as e:
            err = "%s: %s" % (type(e).__name__, e)
        group.check(err)
        if hasattr(backend, "connect"):
            backend.connect()
        all_hits = parallel.sharded_scan(backend, strings, l_dev, group, max_piece=max_piece, offtarget=offtarget, annotation=request)
        if hasattr(backend, "finalize_gathered"):
            all_hits = backend.finalize_gathered(all_hits)
    if l_dev != args.l:  # a length outside the engine's range: the literal keep-filter, on the host
        all_hits = [refilter_hits(h, len(s), args.l) for h, s in zip(all_hits, strings)]
    world = 1 if group is None else group.world
    if own_group and group is not None:
        # the exchange is over: the other ranks leave now (Group.close is a barrier) instead of waiting, watched by
        # each other's abort channels, for however long this rank formats and writes the CSV
        group.close()
        group = None
    stages["upload_scan_fetch_s"] = time.perf_counter() - t_stage
    if request is not None:
        stages["annotation_join_s"] = getattr(backend, "last_annotate_s", None)  # part of upload_scan_fetch_s (this rank's share)
        stages["annotation_strings"] = len(request.annotation.strings)

    # (the native formatter's row buffers are sized for guide lengths 1..50; other lengths take the csv module)
    native = getattr(args, "csv_writer", "native") == "native" and NATIVE_GUIDE_LENGTHS[0] <= args.l <= NATIVE_GUIDE_LENGTHS[1]
    once = getattr(args, "each_contig_once", False)
    dataset = rows.NativeDataset() if native else rows.Dataset()  # Complete_dataset, CROPSR.py:407
    ids = None
    if native:
        # every pass draws its ids from one RNG stream, in order; the pass sizes are known now,
        # so a worker draws pass k+1's ids while pass k is formatted and written
        per_contig = [int(h["pos_plus"].size + h["pos_minus"].size) for h in all_hits]
        sizes = per_contig if once else np.cumsum(per_contig).tolist()
        ids = rows.IdStream(sizes, reverse=True)
    t_stage = time.perf_counter()
    n_rows_written = 0
    # --each-contig-once with the native writer: a pass is one contig and independent of the others, so consecutive passes
    # go to the formatter TOGETHER until they make up four million rows (one call, one team of threads, blocks that span short
    # contigs: a scaffold of 3 500 rows on its own is formatted by a single thread, and a genome has hundreds of them);
    # the bytes are the passes' bytes one after the other, the per-pass lines of time.txt are written when their rows are
    import os
    batching = native and once and not getattr(args, "reference_sleep", False) and os.environ.get("CROPSR_BATCH_PASSES", "1") != "0"
    batch_limit = int(os.environ.get("CROPSR_BATCH_ROWS", BATCH_ROWS))
    batch, batch_rows = [], 0
    in_flight = []  # at most one: (thread, the passes it writes, [its exception]) -- a batch is formatted and written on a
                    # helper thread while this one builds the next batch's tables and collects its ids

    def finish_write(swallow=False):
        nonlocal n_rows_written
        while in_flight:
            th, passes, err = in_flight.pop()
            th.join()
            if err and not swallow:
                raise err[0]
            for ds, _ in passes:
                n_rows_written += len(ds)
                timing.write("Total runtime of the program is " + str(time.time() - begin))  # CROPSR.py:477, once per pass

    def flush_batch():
        nonlocal batch, batch_rows
        if not batch:
            return
        finish_write()  # (the file takes one writer at a time, in order)
        import threading
        passes, err = batch, []

        def work():
            try:
                rows.write_passes_native(args.o, passes, backend.rescore)
            except BaseException as e:
                err.append(e)

        th = threading.Thread(target=work, name="cropsr-csv")
        in_flight.append((th, passes, err))
        th.start()
        batch, batch_rows = [], 0

    try:
        for name, s, hits in zip(names, strings, all_hits):
            print("Searching on Chromosome: ", name[:25], file=out)  # CROPSR.py:410-411
            print("With start of sequence: ", bytes(s[:25]).decode("latin-1"), file=out)
            feats = None
            if request is not None:  # the device's label-set id per row ('+' rows, then '-' rows) + the string table
                feats = (request.annotation.strings, np.concatenate([hits["feat_plus"], hits["feat_minus"]]))
            block = (rows.ContigTable(name, s, hits, args.l, features=feats) if native
                     else rows.ContigRows(name, bytes(s).decode("latin-1"), hits, args.l, features=feats))
            if once:
                dataset = rows.NativeDataset() if native else rows.Dataset()  # opt-in fix of CROPSR.py:407
            dataset.append(block)
            if verbose:
                # CROPSR.py:436-439 counts regex matches BEFORE the keep-filter; the
                # engine reports kept hits, so re-count only for this message.
                import re
                n_sites = sum(1 for _ in re.finditer(rb"(?=.GG)", s)) + sum(1 for _ in re.finditer(rb"(?=CC.)", s))
                print(f"""
                {n_sites:n} Cas9 PAM sites were found on {name[1::]}
                """, file=out)
            if batching:
                batch.append((dataset, ids.next(len(dataset))))
                batch_rows += len(dataset)
                if batch_rows >= batch_limit:
                    flush_batch()
                continue
            if native:  # CROPSR.py:442-474
                rows.write_pass_native(args.o, dataset, backend.rescore, ids)
            else:
                rows.write_pass(args.o, dataset, backend.rescore)
            n_rows_written += len(dataset)
            end = time.time()
            timing.write("Total runtime of the program is " + str(end - begin))  # CROPSR.py:477
            if getattr(args, "reference_sleep", False):
                time.sleep(5)  # CROPSR.py:478
        flush_batch()
        finish_write()
    except BaseException:
        finish_write(swallow=True)  # (no writer thread outlives a failing run)
        raise
    stages["format_write_s"] = time.perf_counter() - t_stage
    timing.close()
    if ids is not None:
        ids.close()
    if getattr(backend, "last_stream", None):  # the pipelined scan's own account of the upload + scan + fetch stage
        stages["scan_stream"] = backend.last_stream
    if getattr(backend, "last_gather", None):  # one process over several devices (--devices): the node's gatherv, in numbers
        stages["node_gatherv"] = backend.last_gather
        stages["devices"] = list(backend.node.devices)
    if own_backend:
        backend.close()
    if verbose:
        print(f"The output file has been generated at {args.o}", file=out)
    if getattr(args, "bench_json", None):
        import json
        kept = int(sum(h["pos_plus"].size + h["pos_minus"].size for h in all_hits))
        stages.update(total_s=time.time() - begin, contigs=len(strings), characters=int(sum(len(v) for v in strings)),
                      kept_hits=kept, rows_written=int(n_rows_written), world=world,
                      gRNAs_per_s_end_to_end=kept / max(1e-9, time.time() - begin))
        with open(args.bench_json, "w") as f:
            json.dump(stages, f)
            f.write("\n")


def _leave(status):
    """The one way out of main() for a process that opened the GPU: if a communicator bootstrap never returned
    (Engine.comm_init), a helper thread still sits inside RCCL and the runtime's tear-down at a normal exit may wait for
    it -- flush and leave through os._exit, on the error paths too."""
    eng_mod = sys.modules.get(__package__ + ".engine")  # (only a run that opened the GPU has imported it)
    if eng_mod is not None:
        eng_mod.leave_if_comm_stuck(status)
    node_mod = sys.modules.get(__package__ + ".node")  # (--devices: ncclCommInitAll runs on a helper thread of the library)
    if node_mod is not None and node_mod.comm_stuck():
        import os
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(status)


def main(argv=None):
    import os
    args = build_parser().parse_args(argv)
    from . import launch
    if getattr(args, "devices", None) and getattr(args, "gpus", 1) > 1:
        sys.exit("cropsr_amd: --devices (one process over several GPUs) and --gpus N (one process per GPU) are two ways to the "
                 "same result: give one of them")
    if launch.wanted(getattr(args, "gpus", 1)):
        # no launcher in the environment: this process (which never touches the GPU) starts the ranks as fresh
        # children of the same command line and leaves with their status (cropsr_amd/launch.py)
        if getattr(args, "device", None) is not None and os.environ.get("CROPSR_GATHER", "rccl") != "host":
            # every rank would open the same GPU, RCCL would refuse the duplicate device after a whole bootstrap round
            # and the run would crawl on over the host transport without a word
            sys.exit("cropsr_amd: --device names ONE GPU, but --gpus %d puts every rank on the GPU of its own rank; drop "
                     "--device (or, to rehearse several ranks on one GPU, set CROPSR_GATHER=host)" % args.gpus)
        sys.stdout.flush()
        sys.exit(launch.spawn_ranks([sys.executable, "-m", "cropsr_amd"] + list(sys.argv[1:] if argv is None else argv),
                                    args.gpus))
    status = 0
    try:
        run(args)
    except SystemExit as e:  # (sys.exit(message) inside run: the message goes to stderr, the status is 1)
        if isinstance(e.code, str):
            sys.stderr.write(e.code + "\n")
        status = e.code if isinstance(e.code, int) else (1 if e.code else 0)
    except Exception as e:
        import traceback
        from . import rendezvous
        status = 1
        if isinstance(e, rendezvous.RankError):  # agreed on by every rank: all leave the same way, together
            if _ACTIVE_GROUP is not None:
                _ACTIVE_GROUP.close()
            sys.stderr.write("cropsr_amd: " + str(e) + "\n")
        else:
            traceback.print_exc()
            if _ACTIVE_GROUP is not None and _ACTIVE_GROUP.world > 1:
                # this rank alone failed (a HIP or RCCL error in the middle of the exchange, ...): its peers may sit in
                # a collective that will never complete -- take the whole run down with the reason
                _ACTIVE_GROUP.abort("%s: %s" % (type(e).__name__, e))
    # every way out of a process that may have opened the GPU ends here
    _leave(status)
    sys.exit(status)


if __name__ == "__main__":
    main()
