"""Host-side face of the MI355X scan/score engine.

Mirrors the two seams of the reference that the HIP library replaces:

  seam 1  CROPSR.py:413-434 -- for one contig string, the kept '+' and '-' PAM
          hits in regex order                        -> Arena.scan_score()
  seam 2  CROPSR.py:285-313 -- rs1_score(ndarray[n,30] uint8) -> ndarray[n] f64
                                                     -> Engine.rs1_score()

Everything numeric happens in libcropsr_hip.so on the GPU; this module only moves
buffers and slices tables.  There is no CPU fallback.
"""
import atexit
import ctypes
import weakref

import numpy as np

from . import _native as nat


def _as_u8(s):
    if isinstance(s, str):
        # one byte per character, whatever the character: positions must not shift
        s = s.encode("ascii", "replace")
    if isinstance(s, np.ndarray):
        return np.ascontiguousarray(s, dtype=np.uint8)
    return np.frombuffer(s, dtype=np.uint8)


def _add_contigs(handle, bufs, ctx):
    """crp_arena_add_contigs_ascii for a list of uint8 arrays: small contigs share one copy and one pack launch."""
    n = len(bufs)
    offs = np.zeros(n, dtype=np.uint64)
    if n:
        ptrs = (ctypes.c_void_p * n)(*[b.ctypes.data if b.size else None for b in bufs])
        lens = np.array([b.size for b in bufs], dtype=np.uint64)
        nat.check(nat.lib().crp_arena_add_contigs_ascii(handle, ptrs, lens.ctypes.data_as(nat.u64p), n,
                                                        offs.ctypes.data_as(nat.u64p)),
                  "crp_arena_add_contigs_ascii", ctx)
    return offs


class Hits:
    """Hit tables of one Arena.scan_score() call (host copies).

    pos_* are ARENA positions, ascending per strand; `contig(k)` returns the
    slice that belongs to contig k with positions local to its string, i.e.
    exactly the regex match indices the reference iterates over:
      '+': i of (?=.GG): start_pos = i-l, end_pos = i, cutsite = i-3   (CROPSR.py:418,157)
      '-': j of (?=CC.): start_pos = j+3+l, end_pos = j+3, cutsite = j  (CROPSR.py:429,433)
    score == -1 marks rows whose 30-character window is incomplete (CROPSR.py:466-468).
    """

    def __init__(self, offsets, lengths, guide_len, cols):
        self.offsets = offsets
        self.lengths = lengths
        self.guide_len = guide_len
        self.pos_plus, self.pre_plus, self.score_plus = cols[0:3]
        self.pos_minus, self.pre_minus, self.score_minus = cols[3:6]
        self.ot_plus = self.ot_minus = None  # (n, 4) uint32 once an off-target scan has run
        self.feat_plus = self.feat_minus = None  # uint32 label-set ids once the annotation join has run
        # (needles in the tables' own dtype, uint32: arena positions stay below 2^31 -- otherwise numpy converts the tables)
        starts = np.asarray(offsets).astype(np.uint32)
        ends = (np.asarray(offsets) + np.asarray(lengths)).astype(np.uint32)
        self._cut_plus = (np.searchsorted(self.pos_plus, starts, "left"), np.searchsorted(self.pos_plus, ends, "left"))
        self._cut_minus = (np.searchsorted(self.pos_minus, starts, "left"), np.searchsorted(self.pos_minus, ends, "left"))

    @property
    def n_plus(self):
        return int(self.pos_plus.size)

    @property
    def n_minus(self):
        return int(self.pos_minus.size)

    def contig(self, k):
        off = int(self.offsets[k])
        a, b = int(self._cut_plus[0][k]), int(self._cut_plus[1][k])
        c, d = int(self._cut_minus[0][k]), int(self._cut_minus[1][k])
        sl = lambda arr, x, y: None if arr is None else arr[x:y]
        out = dict(
            pos_plus=(self.pos_plus[a:b] - np.uint32(off)), pre_plus=sl(self.pre_plus, a, b),
            score_plus=self.score_plus[a:b],
            pos_minus=(self.pos_minus[c:d] - np.uint32(off)), pre_minus=sl(self.pre_minus, c, d),
            score_minus=self.score_minus[c:d])
        if self.ot_plus is not None:  # off-target counts travel with the rows they belong to
            out["ot_plus"], out["ot_minus"] = self.ot_plus[a:b], self.ot_minus[c:d]
        if self.feat_plus is not None:
            out["feat_plus"], out["feat_minus"] = self.feat_plus[a:b], self.feat_minus[c:d]
        return out


class Arena:
    """Device-resident set of contigs (four bit-planes in HBM)."""

    def __init__(self, engine, handle, offsets, lengths):
        self._engine = engine
        self._h = handle
        self.offsets = offsets
        self.lengths = lengths

    def close(self):
        if self._h:
            nat.lib().crp_arena_destroy(self._h)
            self._h = None

    __del__ = close

    def stats(self):
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        nat.check(nat.lib().crp_arena_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)),
                  "crp_arena_stats")
        return dict(n_contigs=a.value, n_chars=b.value, n_words=c.value)

    def tiles(self):
        """dict(geometry="large" | "small", n_tiles, tile_words): the tile shape this arena was sealed with."""
        g, n, w = ctypes.c_int(), ctypes.c_uint64(), ctypes.c_uint64()
        nat.check(nat.lib().crp_arena_tiles(self._h, ctypes.byref(g), ctypes.byref(n), ctypes.byref(w)), "crp_arena_tiles")
        name = [k for k, v in nat.GEOMETRIES.items() if v == g.value][0]
        return dict(geometry=name, n_tiles=int(n.value), tile_words=int(w.value))

    def composition(self):
        """dict(n_plain, n_other): upper-case A/C/G/T characters of the arena, and all others (counted on the GPU)."""
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        nat.check(nat.lib().crp_arena_composition(self._h, ctypes.byref(a), ctypes.byref(b)), "crp_arena_composition",
                  self._engine._ctx)
        return dict(n_plain=a.value, n_other=b.value)

    def scan_score_device(self, guide_len=20, want_pre=False, want_seeds=False):
        """Run the kernels; tables stay in HBM.  Returns (n_plus, n_minus).  want_seeds: the scan also writes the
        seed words the off-target step works on (CRP_SCAN_SEEDS; offtarget_add then skips its own pass over the
        planes)."""
        npl, nmi = ctypes.c_uint64(), ctypes.c_uint64()
        flags = (nat.SCAN_PRE if want_pre else 0) | (nat.SCAN_SEEDS if want_seeds else 0)
        nat.check(nat.lib().crp_scan_score(self._h, guide_len, flags,
                                           ctypes.byref(npl), ctypes.byref(nmi)),
                  "crp_scan_score", self._engine._ctx)
        return npl.value, nmi.value

    def device_tables(self):
        """Raw device addresses (pos_plus, score_plus, pos_minus, score_minus)."""
        p = [ctypes.c_void_p() for _ in range(4)]
        nat.check(nat.lib().crp_hits_device(self._h, *[ctypes.byref(x) for x in p]), "crp_hits_device")
        return tuple(x.value for x in p)

    def fetch(self, n_plus, n_minus, want_pre=False, out=None):
        """Host copies of the tables: [pos_plus, pre_plus | None, score_plus, pos_minus, pre_minus | None, score_minus].
        out: the list a previous fetch returned -- its arrays are filled again where they are large enough (a caller that
        processes genome after genome keeps its pages: a fresh numpy array costs a page fault per 4 KiB on first touch,
        ~14 GB/s on the MI355X boxes, against the link's 56 GB/s into memory that has been touched).  Arrays over pinned
        memory (Engine.empty_tables) are filled by DMA directly, without the staging copy."""
        cols = []
        for k, n in enumerate((n_plus, n_minus)):
            for j, (dtype, wanted) in enumerate(((np.uint32, True), (np.float64, want_pre), (np.float64, True))):
                old = out[3 * k + j] if out is not None else None
                if not wanted:
                    cols.append(None)
                elif old is not None and old.dtype == dtype and old.size >= n and old.flags.c_contiguous and old.flags.writeable:
                    cols.append(old[:n] if old.size > n else old)
                else:
                    cols.append(np.empty(n, dtype=dtype))
        ptr = lambda a, t: None if a is None else a.ctypes.data_as(t)
        nat.check(nat.lib().crp_fetch_hits(
            self._h, ptr(cols[0], nat.u32p), ptr(cols[1], nat.f64p), ptr(cols[2], nat.f64p),
            ptr(cols[3], nat.u32p), ptr(cols[4], nat.f64p), ptr(cols[5], nat.f64p)),
            "crp_fetch_hits", self._engine._ctx)
        return cols

    def scan_score(self, guide_len=20, want_pre=True):
        """Seam 1 + 2 for every contig of the arena -> Hits (host copies)."""
        n_plus, n_minus = self.scan_score_device(guide_len, want_pre)
        cols = self.fetch(n_plus, n_minus, want_pre)
        return Hits(self.offsets, self.lengths, guide_len, cols)

    def count_scored(self):
        """Rows of the last scan that carry a real score (counted on the GPU)."""
        n = ctypes.c_uint64()
        nat.check(nat.lib().crp_count_scored(self._h, ctypes.byref(n)), "crp_count_scored", self._engine._ctx)
        return n.value

    # ---- annotation join (opt-in; include/cropsr_hip.h)
    def annotate_set_track(self, points, ids):
        """The arena's track: strictly ascending arena positions and the label-set id each one opens
        (annotate.Annotation.arena_track builds it)."""
        points = np.ascontiguousarray(points, dtype=np.uint32)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        if points.shape != ids.shape or points.ndim != 1:
            raise ValueError("points and ids must be 1-d arrays of one length")
        nat.check(nat.lib().crp_annotate_set_track(self._h, points.ctypes.data_as(nat.u32p), ids.ctypes.data_as(nat.u32p),
                                                   points.size), "crp_annotate_set_track", self._engine._ctx)

    def annotate_lookup(self, n_plus, n_minus, fetch=True):
        """Label-set id per kept hit of the last scan (NO_FEATURE: none / no cut site), computed on the GPU from the
        resident tables; fetch=False leaves them in HBM for gather_hits(features=True)."""
        if not fetch:
            nat.check(nat.lib().crp_annotate_lookup(self._h, None, None), "crp_annotate_lookup", self._engine._ctx)
            return None
        fp, fm = np.empty(n_plus, dtype=np.uint32), np.empty(n_minus, dtype=np.uint32)
        nat.check(nat.lib().crp_annotate_lookup(self._h, fp.ctypes.data_as(nat.u32p), fm.ctypes.data_as(nat.u32p)),
                  "crp_annotate_lookup", self._engine._ctx)
        return fp, fm

    # ---- off-target seed scan (opt-in; include/cropsr_hip.h)
    def offtarget_add(self, guide_len=20, own_ranges=None):
        """Add the sites of the last scan to the engine's seed histogram; own_ranges: (n, 2) arena
        positions [begin, end) whose hits count (default: all).  Returns the number of sites."""
        n = ctypes.c_uint64()
        if own_ranges is None:
            ptr, k = None, 0
        else:
            own = np.ascontiguousarray(own_ranges, dtype=np.uint64).reshape(-1, 2)
            ptr, k = own.ctypes.data_as(nat.u64p), own.shape[0]
        nat.check(nat.lib().crp_offtarget_add(self._h, int(guide_len), ptr, k, ctypes.byref(n)),
                  "crp_offtarget_add", self._engine._ctx)
        return n.value

    def offtarget_counts(self, n_plus, n_minus, fetch=True):
        """(n, 4) uint32 per strand: other sites at seed distance 0..3 (0xFFFFFFFF: not a site)."""
        if not fetch:
            nat.check(nat.lib().crp_offtarget_counts(self._h, None, None), "crp_offtarget_counts", self._engine._ctx)
            return None
        cp, cm = np.empty((n_plus, 4), dtype=np.uint32), np.empty((n_minus, 4), dtype=np.uint32)
        nat.check(nat.lib().crp_offtarget_counts(self._h, cp.ctypes.data_as(nat.u32p), cm.ctypes.data_as(nat.u32p)),
                  "crp_offtarget_counts", self._engine._ctx)
        return cp, cm

    def offtarget_seeds(self, n_plus, n_minus):
        sp, sm = np.empty(n_plus, dtype=np.uint32), np.empty(n_minus, dtype=np.uint32)
        nat.check(nat.lib().crp_offtarget_seeds(self._h, sp.ctypes.data_as(nat.u32p), sm.ctypes.data_as(nat.u32p)),
                  "crp_offtarget_seeds", self._engine._ctx)
        return sp, sm


class PinnedTables:
    """Hit tables in pinned host memory (crp_host_alloc), as numpy arrays that keep the allocation alive."""

    def __init__(self, rows_plus, rows_minus):
        self._ptrs = []
        self.arrays = tuple(self._alloc(n, dt) for n, dt in ((rows_plus, np.uint32), (rows_plus, np.float64),
                                                             (rows_minus, np.uint32), (rows_minus, np.float64)))

    def _alloc(self, n, dtype):
        import weakref as _wr
        n = int(n)
        nbytes = max(1, n) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        nat.check(nat.lib().crp_host_alloc(nbytes, ctypes.byref(p)), "crp_host_alloc")
        buf = (ctypes.c_uint8 * nbytes).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=n)
        _wr.finalize(buf, nat.lib().crp_host_free, ctypes.c_void_p(p.value))  # (the array holds `buf`; freed with the last view)
        return arr


# Engines still open when the interpreter exits are closed here, in an atexit handler: that runs
# before module globals are torn down and before the HIP runtime's own static destructors, so no
# __del__ ever calls into a runtime that is already gone.
_LIVE_ENGINES = weakref.WeakSet()
_COMM_STUCK = False  # some Engine.comm_init of this process left a helper thread behind inside RCCL


def leave_if_comm_stuck(status=0):
    """Call when the program's output is written: if a communicator bootstrap never returned (Engine.comm_init), a
    thread of this process still sits inside RCCL, and the runtime's tear-down at a normal exit may wait for it --
    leave through os._exit instead."""
    if _COMM_STUCK:
        import os
        import sys
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(status)


def _comm_init_hook(rank):
    """Called on the helper thread right before crp_comm_init.  Production: nothing.  The GPU tests start their ranks with
    CROPSR_TEST_HOOKS=1 and CROPSR_TEST_COMM_INIT_STALL=<rank> to make that rank's bootstrap hang for ever."""
    import os
    if os.environ.get("CROPSR_TEST_HOOKS") == "1" and os.environ.get("CROPSR_TEST_COMM_INIT_STALL") == str(rank):
        import threading
        threading.Event().wait()


@atexit.register
def _close_live_engines():
    for eng in list(_LIVE_ENGINES):
        try:
            eng.close()
        except Exception:
            pass


class Engine:
    """One HIP device opened through libcropsr_hip.so."""

    def __init__(self, device=0):
        self._ctx = None
        self.comm_stuck = False  # comm_init: a helper thread never came back from RCCL's bootstrap
        L = nat.lib()
        h = ctypes.c_void_p()
        st = L.crp_init(int(device), ctypes.byref(h))
        if st != nat.CRP_OK:
            raise nat.CropsrHipError(st, "crp_init(device=%d)" % device)
        self._ctx = h
        self._arenas = weakref.WeakSet()
        _LIVE_ENGINES.add(self)

    def close(self):
        if self._ctx:
            for a in list(self._arenas):  # arenas hold device memory of this context
                a.close()
            if not self.comm_stuck:  # (a helper thread may still be inside RCCL with this context: comm_init)
                nat.lib().crp_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_info(self):
        name = ctypes.create_string_buffer(128)
        cu = ctypes.c_int()
        mem = ctypes.c_uint64()
        nat.check(nat.lib().crp_device_info(self._ctx, name, 128, ctypes.byref(cu), ctypes.byref(mem)),
                  "crp_device_info")
        return dict(name=name.value.decode(), n_cu=cu.value, hbm_bytes=mem.value)

    # ---- arena construction
    def arena(self, contigs, pack="device", pack_threads=8):
        """Upload contig strings (bytes / str / uint8 arrays).

        pack="device": characters go over PCIe and a ballot kernel packs them;
        pack="host":   crp_pack_ascii packs on the host, planes go over PCIe.
        """
        L = nat.lib()
        bufs = [_as_u8(c) for c in contigs]
        total = sum(int(L.crp_arena_words_for(b.size)) for b in bufs)
        h = ctypes.c_void_p()
        nat.check(L.crp_arena_create(self._ctx, L.crp_arena_words_total(total), ctypes.byref(h)),
                  "crp_arena_create", self._ctx)
        offsets = np.zeros(len(bufs), dtype=np.uint64)
        lengths = np.array([b.size for b in bufs], dtype=np.uint64)
        try:
            if pack == "device":
                offsets = _add_contigs(h, bufs, self._ctx)  # one call: small contigs travel and are packed in batches
            elif pack == "host":
                for k, b in enumerate(bufs):
                    off = ctypes.c_uint64()
                    planes = pack_ascii(b, pack_threads)
                    nat.check(L.crp_arena_add_contig_packed(h, *[p.ctypes.data_as(nat.u64p) for p in planes],
                                                            b.size, ctypes.byref(off)), "crp_arena_add_contig", self._ctx)
                    offsets[k] = off.value
            else:
                raise ValueError("pack must be 'device' or 'host'")
            nat.check(L.crp_arena_seal(h), "crp_arena_seal", self._ctx)
        except Exception:
            L.crp_arena_destroy(h)
            raise
        a = Arena(self, h, offsets, lengths)
        self._arenas.add(a)
        return a

    # ---- seam 1 as a pipeline (crp_scan_stream)
    def stream_prepare(self, slice_chars=0):
        """Open the pipeline's lanes (two further contexts, three slice arenas) ahead of the first scan_stream()."""
        nat.check(nat.lib().crp_scan_stream_prepare(self._ctx, int(slice_chars)), "crp_scan_stream_prepare", self._ctx)

    def empty_tables(self, rows_plus, rows_minus):
        """Four numpy arrays (pos_plus u32, score_plus f64, pos_minus u32, score_minus f64) over PINNED host memory
        (crp_host_alloc): fetches and scan_stream() fill them by DMA, with no staging copy and no first-touch page faults.
        For callers that keep their tables from genome to genome (pinning 0.6 GB costs more than one copy saves).  The
        memory is released when the arrays are garbage-collected."""
        return PinnedTables(rows_plus, rows_minus).arrays

    def scan_stream(self, contigs, guide_len=20, want_pre=False, out=None, slice_chars=0, density=1.0 / 6):
        """All contigs through crp_scan_stream: upload, scan and table fetch as a pipeline over slices of the genome,
        H2D and D2H side by side.  Returns node.NodeHits (one table per strand, contig after contig, positions LOCAL to the
        contig string; hits.contig(k) like Hits.contig(k)) with .stream_stats.  out: (pos_plus, score_plus, pos_minus,
        score_minus) arrays to fill -- e.g. empty_tables(); default: fresh pageable arrays sized for `density` hits per
        character and strand (a multiple of what genomes have; untouched pages cost nothing).  Tables that turn out too small
        (poly-G) are replaced by exactly sized ones and the scan repeated."""
        from .node import NodeHits
        L = nat.lib()
        bufs = [_as_u8(c) for c in contigs]
        n = len(bufs)
        ptrs = (ctypes.c_void_p * max(1, n))(*[b.ctypes.data if b.size else None for b in bufs])
        lens = np.array([b.size for b in bufs], dtype=np.uint64)
        total = int(lens.sum())
        per = np.zeros((max(1, n), 2), dtype=np.uint64)
        stats = np.zeros(12, dtype=np.float64)
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        if out is None:
            cap = int(total * density) + 1024
            out = (np.empty(cap, np.uint32), np.empty(cap, np.float64), np.empty(cap, np.uint32), np.empty(cap, np.float64))
        for attempt in range(2):
            pp, sp, pm, sm = out
            st = L.crp_scan_stream(self._ctx, ptrs, lens.ctypes.data_as(nat.u64p), n, int(guide_len), nat.SCAN_PRE if want_pre else 0,
                                   int(slice_chars), pp.ctypes.data_as(nat.u32p), sp.ctypes.data_as(nat.f64p), min(pp.size, sp.size),
                                   pm.ctypes.data_as(nat.u32p), sm.ctypes.data_as(nat.f64p), min(pm.size, sm.size),
                                   per.ctypes.data_as(nat.u64p), ctypes.byref(a), ctypes.byref(b), stats.ctypes.data_as(nat.f64p))
            if st != nat.CRP_ERR_CAPACITY or attempt:
                break
            out = (np.empty(a.value, np.uint32), np.empty(a.value, np.float64), np.empty(b.value, np.uint32), np.empty(b.value, np.float64))
        nat.check(st, "crp_scan_stream", self._ctx)
        hits = NodeHits(per[:n], [out[0][:a.value], out[1][:a.value], out[2][:b.value], out[3][:b.value]], guide_len)
        keys = ("wall_s", "uploader_busy_s", "drainer_busy_s", "slices", "lanes", "first_slice_on_host_s", "uploader_waiting_s",
                "drainer_waiting_s", "copier_busy_s", "copier_waiting_s", "copier_bytes", "tables_pinned")
        hits.stream_stats = dict(zip(keys, (float(v) for v in stats)))
        return hits

    def genome(self, contigs, max_words=None, pack="device"):
        """Like arena(), for inputs that may exceed one arena (2^31 characters)."""
        return Genome(self, contigs, max_words, pack)

    def arena_builder(self, lengths):
        """Incremental upload for genomes too big to hold as one list: give the
        contig string lengths up front, then add() each string in that order."""
        return ArenaBuilder(self, lengths)

    # ---- seam 2
    def score_30mers(self, rows, order=nat.ORDER_BODY4):
        """(pre, score) for an (n,30) uint8 array; see crp_score_30mers.  `order`
        selects the accumulation order of the reference's BLAS for rows at the
        tail of a batch (include/cropsr_hip.h, CRP_ORDER_*)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        if rows.ndim != 2 or rows.shape[1] != 30:
            raise ValueError("expected an (n, 30) uint8 array")
        n = rows.shape[0]
        pre = np.empty(n, dtype=np.float64)
        score = np.empty(n, dtype=np.float64)
        nat.check(nat.lib().crp_score_30mers(self._ctx, rows.ctypes.data_as(nat.u8p), n, int(order),
                                             pre.ctypes.data_as(nat.f64p), score.ctypes.data_as(nat.f64p)),
                  "crp_score_30mers", self._ctx)
        return pre, score

    def rs1_score(self, sequences):
        """Drop-in for the reference's rs1_score (CROPSR.py:285): same argument,
        same return value -- including which rows of the batch are summed in
        which BLAS order."""
        sequences = np.ascontiguousarray(sequences, dtype=np.uint8)
        n = sequences.shape[0]
        if n == 1:
            return self.score_30mers(sequences, nat.ORDER_DOT1)[1]
        score = self.score_30mers(sequences, nat.ORDER_BODY4)[1]
        base = 4 * (n // 4)
        if n % 4 >= 2:
            score[base:base + 2] = self.score_30mers(sequences[base:base + 2], nat.ORDER_TAIL2)[1]
        return score

    def configure(self, two_pass=None, chain_timeout_us=None, geometry=None):
        """two_pass=False (the default): one launch per scan, table offsets from the chained
        scan inside the emit kernel; True: the count / tile-scan / emit launch sequence.
        Same results either way (every GPU parity test runs in both modes).
        chain_timeout_us: how long a single-launch workgroup may wait for a predecessor."""
        if two_pass is not None:
            nat.check(nat.lib().crp_configure(self._ctx, nat.OPT_TWO_PASS, int(bool(two_pass))), "crp_configure")
        if chain_timeout_us is not None:
            nat.check(nat.lib().crp_configure(self._ctx, nat.OPT_CHAIN_TIMEOUT_US, int(chain_timeout_us)), "crp_configure")
        if geometry is not None:  # "auto" | "large" | "small": the tile shape of arenas sealed from now on
            nat.check(nat.lib().crp_configure(self._ctx, nat.OPT_TILE_GEOMETRY, nat.GEOMETRIES[geometry]), "crp_configure")

    def query(self):
        """Fallback and communicator state: chain_timeouts (single-launch scans repeated as three
        launches), two_pass_active, comm_world, comm_rank."""
        out = {}
        for name, what in (("chain_timeouts", nat.Q_CHAIN_TIMEOUTS), ("two_pass_active", nat.Q_TWO_PASS_ACTIVE),
                           ("comm_world", nat.Q_COMM_WORLD), ("comm_rank", nat.Q_COMM_RANK)):
            v = ctypes.c_int64()
            nat.check(nat.lib().crp_query(self._ctx, what, ctypes.byref(v)), "crp_query")
            out[name] = int(v.value)
        return out

    def hbm(self):
        """(free, total) bytes of device memory right now -- the whole device, whoever holds it (hipMemGetInfo)."""
        out = []
        for what in (nat.Q_HBM_FREE, nat.Q_HBM_TOTAL):
            v = ctypes.c_int64()
            nat.check(nat.lib().crp_query(self._ctx, what, ctypes.byref(v)), "crp_query", self._ctx)
            out.append(int(v.value))
        return tuple(out)

    # ---- multi-GPU: RCCL inside the library (crp_comm.cpp); `group` is a rendezvous.Group
    def comm_init(self, group, timeout_s=None):
        """Create the RCCL communicator of this engine: rank 0 draws the unique id, the group's
        control sockets carry it to the others.  Errors are agreed on before anyone proceeds.

        ncclCommInitRank has no time-out of its own; a bootstrap that never completes on some node must not
        take the run with it (the scan needs no collective, the exchange has a host transport).  The call is
        therefore made on a helper thread and given `timeout_s` seconds (CROPSR_COMM_INIT_TIMEOUT_S, default
        180): a rank whose call has not returned by then reports that as its error, group.check makes it
        every rank's error, and the callers fall back to the host transport.  The helper thread stays behind
        inside RCCL: `comm_stuck` is then set, close() leaves the context alone, and the program should leave
        through os._exit once its output is written (bench.py and the CLI do)."""
        import os
        import threading
        global _COMM_STUCK
        L = nat.lib()
        if timeout_s is None:
            timeout_s = float(os.environ.get("CROPSR_COMM_INIT_TIMEOUT_S", "180"))
        ident, err = None, None
        if group.rank == 0:
            buf = (ctypes.c_uint8 * nat.COMM_ID_BYTES)()
            st = L.crp_comm_unique_id(buf)
            if st != nat.CRP_OK:
                err = "crp_comm_unique_id: " + L.crp_strerror(st).decode()
            else:
                ident = bytes(buf)
        # rank 0's failure to draw an id is agreed on FIRST: nobody enters the blocking ncclCommInitRank
        # bootstrap with an id that was never drawn
        group.check(err)
        ident = group.bcast(ident)
        buf = (ctypes.c_uint8 * nat.COMM_ID_BYTES).from_buffer_copy(ident)
        done = {}

        def init():
            try:
                _comm_init_hook(group.rank)  # (a no-op; the tests replace it to make one rank's bootstrap hang)
                done["st"] = L.crp_comm_init(self._ctx, buf, group.rank, group.world)
            except BaseException as e:  # the thread must not die silently: its error becomes this rank's error below
                done["exc"] = "%s: %s" % (type(e).__name__, e)

        worker = threading.Thread(target=init, name="crp_comm_init", daemon=True)
        worker.start()
        worker.join(timeout_s if timeout_s > 0 else None)
        err = None
        if worker.is_alive():
            self.comm_stuck = _COMM_STUCK = True
            err = "crp_comm_init did not return within %.0f s (RCCL bootstrap)" % timeout_s
        elif "st" not in done:
            err = "crp_comm_init was never made: " + done.get("exc", "the helper thread ended without a result")
        elif done["st"] != nat.CRP_OK:
            err = "crp_comm_init: %s [%s]" % (L.crp_strerror(done["st"]).decode(), L.crp_last_error(self._ctx).decode())
        try:
            group.check(err)
        except Exception:
            # a peer's bootstrap hung: this rank's communicator (if it got one) has no usable peers, and destroying it
            # may wait for them -- treat the context like a stuck one
            if err is None:
                self.comm_stuck = _COMM_STUCK = True
            raise

    def comm_barrier(self):
        nat.check(nat.lib().crp_comm_barrier(self._ctx), "crp_comm_barrier", self._ctx)

    def comm_allreduce(self, values, op="sum"):
        a = (ctypes.c_double * len(values))(*[float(v) for v in values])
        nat.check(nat.lib().crp_comm_allreduce_f64(self._ctx, a, len(values), nat.REDUCE_MAX if op == "max" else nat.REDUCE_SUM),
                  "crp_comm_allreduce_f64", self._ctx)
        return list(a)

    def gather_hits(self, arena, root=0, offtarget=False, pre=False, features=False, pos16=True):
        """The gatherv of the path (crp_gather_hits): every rank's tables of `arena` (None: empty) into
        root's HBM.  pre=True: the f64 column is the pre-sigmoid sum instead of the score.  pos16 (the same on every
        rank): positions cross the links as 16 bits per hit + one word per 65 536 arena positions (CRP_GATHER_POS16:
        exact; 10 B per hit instead of 12).  Returns the (world, 2) counts every rank contributed."""
        world = self.query()["comm_world"]
        counts = np.zeros((world, 2), dtype=np.uint64)
        flags = ((nat.GATHER_OFFTARGET if offtarget else 0) | (nat.GATHER_PRE if pre else 0) | (nat.GATHER_FEATURES if features else 0) |
                 (nat.GATHER_POS16 if pos16 else 0))
        nat.check(nat.lib().crp_gather_hits(self._ctx, arena._h if arena is not None else None, int(root),
                                            flags, counts.ctypes.data_as(nat.u64p)),
                  "crp_gather_hits", self._ctx)
        return counts

    def gather_bytes(self):
        """Bytes this rank sent to the root (a peer) or received from all peers (the root) in the last gather_hits."""
        v = ctypes.c_int64()
        nat.check(nat.lib().crp_query(self._ctx, nat.Q_GATHER_BYTES, ctypes.byref(v)), "crp_query", self._ctx)
        return int(v.value)

    def gathered_fetch(self, rank, counts, offtarget=False, features=False):
        """Root: host copies of what `rank` contributed to the last gather_hits -> column dict."""
        n_plus, n_minus = int(counts[rank][0]), int(counts[rank][1])
        out = {"pos_plus": np.empty(n_plus, np.uint32), "score_plus": np.empty(n_plus, np.float64),
               "pos_minus": np.empty(n_minus, np.uint32), "score_minus": np.empty(n_minus, np.float64)}
        if offtarget:
            out["ot_plus"] = np.empty((n_plus, 4), np.uint32)
            out["ot_minus"] = np.empty((n_minus, 4), np.uint32)
        ptr = lambda k, t: out[k].ctypes.data_as(t) if k in out else None
        nat.check(nat.lib().crp_gathered_fetch(self._ctx, int(rank), ptr("pos_plus", nat.u32p), ptr("score_plus", nat.f64p),
                                               ptr("ot_plus", nat.u32p), ptr("pos_minus", nat.u32p),
                                               ptr("score_minus", nat.f64p), ptr("ot_minus", nat.u32p)),
                  "crp_gathered_fetch", self._ctx)
        if features:
            out["feat_plus"], out["feat_minus"] = np.empty(n_plus, np.uint32), np.empty(n_minus, np.uint32)
            nat.check(nat.lib().crp_gathered_fetch_features(self._ctx, int(rank), out["feat_plus"].ctypes.data_as(nat.u32p),
                                                            out["feat_minus"].ctypes.data_as(nat.u32p)),
                      "crp_gathered_fetch_features", self._ctx)
        return out

    # ---- off-target seed scan, engine-wide steps (per-arena steps: Arena.offtarget_*)
    def offtarget_reset(self):
        nat.check(nat.lib().crp_offtarget_reset(self._ctx), "crp_offtarget_reset", self._ctx)

    def offtarget_reduce(self):
        """RCCL all-reduce of the site histogram over the communicator (no-op without one)."""
        nat.check(nat.lib().crp_offtarget_reduce(self._ctx), "crp_offtarget_reduce", self._ctx)

    def offtarget_solve(self):
        nat.check(nat.lib().crp_offtarget_solve(self._ctx), "crp_offtarget_solve", self._ctx)

    def offtarget_hist(self, new=None):
        """The 4^12 site histogram as a uint32 array; new: replace it (before solve)."""
        if new is not None:
            a = np.ascontiguousarray(new, dtype=np.uint32)
            if a.size != nat.OT_SEEDS:
                raise ValueError("the histogram has 4^12 entries")
            nat.check(nat.lib().crp_offtarget_hist_set(self._ctx, a.ctypes.data_as(nat.u32p)), "crp_offtarget_hist_set", self._ctx)
            return a
        a = np.empty(nat.OT_SEEDS, dtype=np.uint32)
        nat.check(nat.lib().crp_offtarget_hist_get(self._ctx, a.ctypes.data_as(nat.u32p)), "crp_offtarget_hist_get", self._ctx)
        return a

    # ---- measurement
    def profile(self, on=2):
        """0 off, 1 HIP events around the emit+score kernel only, 2 (or True) around all three kernels."""
        level = 2 if on is True else int(on)
        nat.check(nat.lib().crp_profile_enable(self._ctx, level), "crp_profile_enable")

    def profile_read(self, reset=True):
        """{kernel kind: {ms, launches}} since the last reset, for every kind of nat.KINDS."""
        out = {}
        for k, name in enumerate(nat.KINDS):
            ms, n = ctypes.c_double(), ctypes.c_uint64()
            nat.check(nat.lib().crp_profile_read_kind(self._ctx, k, ctypes.byref(ms), ctypes.byref(n), int(reset)),
                      "crp_profile_read_kind")
            out[name] = dict(ms=ms.value, launches=int(n.value))
        return out


class ArenaBuilder:
    """Incremental upload: announce the contig lengths, then add() the strings in that order.  Small contigs are held
    back (by reference) and handed to the library in batches -- one host-to-device copy and one pack launch per batch
    instead of one each per contig; a large contig flushes what is pending and goes at once, so at most one large
    string is alive here at a time."""

    SMALL = 8 << 20       # the library's own threshold (a quarter of its staging buffer)
    FLUSH_BYTES = 24 << 20

    def __init__(self, engine, lengths):
        L = nat.lib()
        self._engine = engine
        self._lengths = [int(n) for n in lengths]
        total = sum(int(L.crp_arena_words_for(n)) for n in self._lengths)
        self._h = ctypes.c_void_p()
        nat.check(L.crp_arena_create(engine._ctx, L.crp_arena_words_total(total), ctypes.byref(self._h)),
                  "crp_arena_create", engine._ctx)
        self._offsets = []
        self._pending, self._pending_bytes, self._n_added = [], 0, 0

    def _flush(self):
        if self._pending:
            self._offsets.extend(int(o) for o in _add_contigs(self._h, self._pending, self._engine._ctx))
            self._pending, self._pending_bytes = [], 0

    def add(self, contig):
        b = _as_u8(contig)
        k = self._n_added
        if k >= len(self._lengths) or b.size != self._lengths[k]:
            raise ValueError("contig %d: length differs from the one announced" % k)
        self._n_added += 1
        self._pending.append(b)
        self._pending_bytes += b.size
        if b.size >= self.SMALL or self._pending_bytes >= self.FLUSH_BYTES:
            self._flush()

    def seal(self):
        if self._n_added != len(self._lengths):
            raise ValueError("not every announced contig was added")
        self._flush()
        nat.check(nat.lib().crp_arena_seal(self._h), "crp_arena_seal", self._engine._ctx)
        a = Arena(self._engine, self._h, np.array(self._offsets, dtype=np.uint64),
                  np.array(self._lengths, dtype=np.uint64))
        self._engine._arenas.add(a)
        self._h = None
        return a


class Genome:
    """Any number of contigs spread over as many arenas as needed (one arena holds
    fewer than 2^31 characters).  Contigs keep their order; a contig never spans
    two arenas."""

    def __init__(self, engine, contigs, max_words=None, pack="device"):
        L = nat.lib()
        limit = int(max_words) if max_words else int(L.crp_arena_max_words())
        self._engine = engine
        bufs = [_as_u8(c) for c in contigs]
        groups, cur, used = [], [], 1
        for k, b in enumerate(bufs):
            need = int(L.crp_arena_words_for(b.size))
            if need + 1 > limit:
                raise ValueError("contig %d (%d characters) does not fit one arena" % (k, b.size))
            if used + need > limit and cur:
                groups.append(cur)
                cur, used = [], 1
            cur.append(k)
            used += need
        if cur or not groups:
            groups.append(cur)
        self.arenas = [engine.arena([bufs[k] for k in g], pack=pack) for g in groups]
        self.groups = groups  # contig indices per arena, in arena order
        self.annotate_s = None
        self._where = {}
        for a, g in enumerate(groups):
            for j, k in enumerate(g):
                self._where[k] = (a, j)
        self.n_contigs = len(bufs)

    def annotate(self, request, counts, fetch=True):
        """The annotation join over every arena's resident tables (annotate.Request; counts: (n_plus, n_minus) per
        arena).  Returns [(feat_plus, feat_minus)] per arena, or None with fetch=False."""
        import time
        t0 = time.perf_counter()
        out = []
        for a, g, (n_plus, n_minus) in zip(self.arenas, self.groups, counts):
            a.annotate_set_track(*request.track([(k, int(a.offsets[j]), int(a.lengths[j])) for j, k in enumerate(g)]))
            out.append(a.annotate_lookup(n_plus, n_minus, fetch=fetch))
        self.annotate_s = time.perf_counter() - t0  # track lay-out + upload + look-up (+ the copy of the ids to the host)
        return out if fetch else None

    def scan_score(self, guide_len=20, want_pre=False, offtarget=False, seeds_from_scan=True, annotation=None):
        """Seam 1 + 2 for every contig.  offtarget=True also runs the genome-wide seed scan over all
        arenas (single process: no reduce) and attaches (n, 4) counts to every contig's hits;
        seeds_from_scan=False makes the off-target step derive its seeds from the planes itself (the
        path guide lengths other than 20 always take) instead of receiving them from the scan."""
        if not offtarget:
            per_arena = [a.scan_score(guide_len, want_pre) for a in self.arenas]
            if annotation is not None:  # (the tables of an arena stay valid until its next scan)
                for h, f in zip(per_arena, self.annotate(annotation, [(h.n_plus, h.n_minus) for h in per_arena])):
                    h.feat_plus, h.feat_minus = f
            return GenomeHits(self, per_arena)
        eng = self._engine
        eng.offtarget_reset()
        counts, per_arena = [], []
        for a in self.arenas:
            n = a.scan_score_device(guide_len, want_pre, want_seeds=seeds_from_scan)
            a.offtarget_add(guide_len)
            counts.append(n)
        eng.offtarget_solve()
        for a, (n_plus, n_minus) in zip(self.arenas, counts):
            # (the tables of an arena stay valid until its next scan)
            h = Hits(a.offsets, a.lengths, guide_len, a.fetch(n_plus, n_minus, want_pre))
            h.ot_plus, h.ot_minus = a.offtarget_counts(n_plus, n_minus)
            per_arena.append(h)
        if annotation is not None:
            for h, f in zip(per_arena, self.annotate(annotation, counts)):
                h.feat_plus, h.feat_minus = f
        return GenomeHits(self, per_arena)

    def close(self):
        for a in self.arenas:
            a.close()


class GenomeHits:
    def __init__(self, genome, per_arena):
        self._genome = genome
        self.per_arena = per_arena
        self.n_plus = sum(h.n_plus for h in per_arena)
        self.n_minus = sum(h.n_minus for h in per_arena)

    def contig(self, k):
        a, j = self._genome._where[k]
        return self.per_arena[a].contig(j)


def pack_ascii(text, n_threads=1):
    """Host packing (crp_pack_ascii): characters -> (hi, lo, up, ac) uint64 planes."""
    b = _as_u8(text)
    n_words = (b.size + 63) // 64
    planes = [np.empty(n_words, dtype=np.uint64) for _ in range(4)]
    nat.check(nat.lib().crp_pack_ascii(b.ctypes.data_as(nat.u8p), b.size,
                                       *[p.ctypes.data_as(nat.u64p) for p in planes], n_threads),
              "crp_pack_ascii")
    return planes
