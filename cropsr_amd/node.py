"""One process over N GPUs: the host-side face of the library's node handle (crp_node_*, include/cropsr_hip.h).

The reference is ONE process with one contig loop (CROPSR.py:333 `main`, :409 `for chromosome, sequence in ...`).
SURVEY.md section 8(b) asks for a handle through which that one process drives the whole node -- "multi-GPU fan-out
happens inside the library (one stream per device), not via Python threads".  This module is only the binding: the cut
of the genome into contiguous equal shares with halos, the uploads (one host thread per device), the scans queued on
every device before any is waited for, and the path's one exchange -- the gatherv of the per-device hit tables to a root
device, on RCCL (ncclCommInitAll, one grouped send/recv) or as device-to-device copies -- all happen inside
libcropsr_hip.so (cropsr_amd/csrc/crp_node.cpp).  What comes back is ONE table per strand, contig after contig,
positions local to the contig string: bit for bit what Engine.arena(contigs).scan_score() returns on one GPU.

    with Node([0, 1, 2, 3]) as node:
        node.load(contig_strings)
        hits = node.scan(guide_len=20)          # NodeHits; hits.contig(k) like engine.Hits.contig(k)

The process-per-GPU path (parallel.sharded_scan over rendezvous.Group) is unchanged beside it.  No CPU fallback.
"""
import ctypes

import numpy as np

from . import _native as nat
from .engine import _as_u8


def plan_shares(lengths, world, min_piece=0):
    """crp_plan_shares: [(contig, start, end, device)] -- the cut crp_node_load makes (host code, no GPU)."""
    L = nat.lib()
    lens = np.ascontiguousarray(lengths, dtype=np.uint64)
    cap = int(lens.size) + int(world)
    out = np.zeros((cap, 4), dtype=np.uint64)
    n = ctypes.c_uint64()
    nat.check(L.crp_plan_shares(lens.ctypes.data_as(nat.u64p), lens.size, int(world), int(min_piece),
                                out.ctypes.data_as(nat.u64p), cap, ctypes.byref(n)), "crp_plan_shares")
    return [tuple(int(v) for v in row) for row in out[:n.value]]


class NodeError(nat.CropsrHipError):
    pass


def comm_stuck():
    """crp_node_comm_stuck: communicator bootstraps of this process whose helper thread never came back from RCCL."""
    return int(nat.lib().crp_node_comm_stuck())


class NodeHits:
    """The gathered tables of one Node.scan(): pos_* are positions LOCAL to their contig string (the regex match indices
    of CROPSR.py:418 / :429), contig after contig; `contig(k)` slices contig k's rows like engine.Hits.contig(k)."""

    def __init__(self, per_contig, cols, guide_len):
        self.guide_len = guide_len
        self.pos_plus, self.score_plus, self.pos_minus, self.score_minus = cols
        self.ot_plus = self.ot_minus = None      # (n, 4) uint32 after a gather with offtarget=True
        self.feat_plus = self.feat_minus = None  # uint32 label-set ids after a gather with features=True
        counts = np.asarray(per_contig, dtype=np.int64).reshape(-1, 2)
        self._cut_plus = np.concatenate(([0], np.cumsum(counts[:, 0])))
        self._cut_minus = np.concatenate(([0], np.cumsum(counts[:, 1])))
        self.counts = counts

    @property
    def n_plus(self):
        return int(self.pos_plus.size)

    @property
    def n_minus(self):
        return int(self.pos_minus.size)

    def contig(self, k):
        a, b = int(self._cut_plus[k]), int(self._cut_plus[k + 1])
        c, d = int(self._cut_minus[k]), int(self._cut_minus[k + 1])
        out = dict(pos_plus=self.pos_plus[a:b], score_plus=self.score_plus[a:b],
                   pos_minus=self.pos_minus[c:d], score_minus=self.score_minus[c:d])
        if self.ot_plus is not None:
            out["ot_plus"], out["ot_minus"] = self.ot_plus[a:b], self.ot_minus[c:d]
        if self.feat_plus is not None:
            out["feat_plus"], out["feat_minus"] = self.feat_plus[a:b], self.feat_minus[c:d]
        return out


class Node:
    """N HIP devices opened through ONE handle of libcropsr_hip.so (crp_node_init).  `devices`: HIP device indices as
    this process sees them; an index may repeat (each entry is a logical device with a context of its own -- a
    rehearsal of the N-device path on one GPU; the exchange then runs as device-to-device copies, RCCL refuses
    duplicates)."""

    def __init__(self, devices):
        self._h = None
        L = nat.lib()
        ids = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        h = ctypes.c_void_p()
        st = L.crp_node_init(len(devices), ids, ctypes.byref(h))
        if st != nat.CRP_OK:
            raise nat.CropsrHipError(st, "crp_node_init(devices=%r)" % (list(devices),))
        self._h = h
        self.devices = [int(d) for d in devices]
        self.n_contigs = 0

    def close(self):
        if self._h:
            nat.lib().crp_node_destroy(self._h)
            self._h = None

    def __del__(self):
        try:  # (at interpreter exit the module globals may already be gone)
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, st, what):
        if st != nat.CRP_OK:
            raise nat.CropsrHipError(st, what, nat.lib().crp_node_last_error(self._h).decode())

    @property
    def size(self):
        return len(self.devices)

    def ctx(self, k):
        """Raw crp_ctx of logical device k (for crp_configure / crp_profile_* / crp_device_info)."""
        return ctypes.c_void_p(nat.lib().crp_node_ctx(self._h, int(k)))

    def configure(self, two_pass=None, geometry=None):
        for k in range(self.size):
            if two_pass is not None:
                nat.check(nat.lib().crp_configure(self.ctx(k), nat.OPT_TWO_PASS, int(bool(two_pass))), "crp_configure")
            if geometry is not None:
                nat.check(nat.lib().crp_configure(self.ctx(k), nat.OPT_TILE_GEOMETRY, nat.GEOMETRIES[geometry]), "crp_configure")

    def device_info(self, k=0):
        name = ctypes.create_string_buffer(128)
        cu = ctypes.c_int()
        mem = ctypes.c_uint64()
        nat.check(nat.lib().crp_device_info(self.ctx(k), name, 128, ctypes.byref(cu), ctypes.byref(mem)), "crp_device_info")
        return dict(name=name.value.decode(), n_cu=cu.value, hbm_bytes=mem.value)

    def profile(self, on):
        for k in range(self.size):
            nat.check(nat.lib().crp_profile_enable(self.ctx(k), int(on)), "crp_profile_enable")

    def profile_read(self, k, reset=True):
        """{kernel kind: {ms, launches}} of logical device k since the last reset."""
        out = {}
        for kind, name in enumerate(nat.KINDS):
            ms, n = ctypes.c_double(), ctypes.c_uint64()
            nat.check(nat.lib().crp_profile_read_kind(self.ctx(k), kind, ctypes.byref(ms), ctypes.byref(n), int(reset)),
                      "crp_profile_read_kind")
            out[name] = dict(ms=ms.value, launches=int(n.value))
        return out

    # ---- the genome
    def load(self, contigs):
        """The contig strings CROPSR.py:409 iterates over (bytes / str / uint8 arrays), in order: cut into
        len(devices) contiguous equal shares with halos and uploaded, every share to its device."""
        bufs = [_as_u8(c) for c in contigs]
        n = len(bufs)
        ptrs = (ctypes.c_void_p * max(1, n))(*[b.ctypes.data if b.size else None for b in bufs])
        lens = np.array([b.size for b in bufs], dtype=np.uint64)
        self._check(nat.lib().crp_node_load(self._h, ptrs, lens.ctypes.data_as(nat.u64p), n), "crp_node_load")
        self.n_contigs = n
        self.lengths = lens

    def plan(self):
        """[dict(contig, start, end, device, arena_offset, halo_before, arena)] -- the cut crp_node_load made (arena: which
        of the device's arenas the piece lies in; a device has more than one only beyond 2^31 characters, or with
        set_option(arena_words=...))."""
        n = ctypes.c_uint64()
        cap = self.n_contigs + self.size
        out = np.zeros((cap, 7), dtype=np.uint64)
        st = nat.lib().crp_node_plan(self._h, out.ctypes.data_as(nat.u64p), cap, ctypes.byref(n))
        if st == nat.CRP_ERR_CAPACITY:  # (more arenas than devices: every further arena may add a piece)
            cap = int(n.value)
            out = np.zeros((cap, 7), dtype=np.uint64)
            st = nat.lib().crp_node_plan(self._h, out.ctypes.data_as(nat.u64p), cap, ctypes.byref(n))
        self._check(st, "crp_node_plan")
        keys = ("contig", "start", "end", "device", "arena_offset", "halo_before", "arena")
        return [dict(zip(keys, (int(v) for v in row))) for row in out[:n.value]]

    def set_option(self, arena_words=None, comm_init_timeout_s=None, collective_timeout_s=None):
        """crp_node_set_option: most words per arena (0: the library's limit), and the two bounds on RCCL waits (seconds;
        <= 0: none)."""
        L = nat.lib()
        if arena_words is not None:
            self._check(L.crp_node_set_option(self._h, nat.NODE_OPT_ARENA_WORDS, int(arena_words)), "crp_node_set_option(arena_words)")
        if comm_init_timeout_s is not None:
            self._check(L.crp_node_set_option(self._h, nat.NODE_OPT_COMM_INIT_TIMEOUT_MS, int(round(comm_init_timeout_s * 1000))),
                        "crp_node_set_option(comm_init_timeout)")
        if collective_timeout_s is not None:
            self._check(L.crp_node_set_option(self._h, nat.NODE_OPT_COLLECTIVE_TIMEOUT_MS, int(round(collective_timeout_s * 1000))),
                        "crp_node_set_option(collective_timeout)")

    def n_arenas(self, k):
        return int(nat.lib().crp_node_arenas(self._h, int(k)))

    def _arenas(self, k):
        L = nat.lib()
        return [ctypes.c_void_p(L.crp_node_arena_at(self._h, int(k), j)) for j in range(self.n_arenas(k))]

    def arena_stats(self, k):
        """dict(n_texts, n_chars, n_tiles, geometry, n_arenas) of logical device k's arenas together, or None if it got no
        piece."""
        arenas = self._arenas(k)
        if not arenas:
            return None
        out = dict(n_texts=0, n_chars=0, n_tiles=0, geometry=None, n_arenas=len(arenas))
        for a in arenas:
            nc, nch, nw = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
            nat.check(nat.lib().crp_arena_stats(a, ctypes.byref(nc), ctypes.byref(nch), ctypes.byref(nw)), "crp_arena_stats")
            g, nt, tw = ctypes.c_int(), ctypes.c_uint64(), ctypes.c_uint64()
            nat.check(nat.lib().crp_arena_tiles(a, ctypes.byref(g), ctypes.byref(nt), ctypes.byref(tw)), "crp_arena_tiles")
            name = [key for key, v in nat.GEOMETRIES.items() if v == g.value][0]
            out["n_texts"] += int(nc.value)
            out["n_chars"] += int(nch.value)
            out["n_tiles"] += int(nt.value)
            out["geometry"] = name if out["geometry"] in (None, name) else "mixed"
        return out

    # ---- seam 1 + 2 on every device at once
    def scan_score_device(self, guide_len=20, want_pre=False, want_seeds=False):
        """crp_node_scan_score: the scan queued on every device, then collected; tables stay in HBM.  Returns the rows
        of all devices' tables together (including the few hits inside halos).  want_seeds: the scan also writes the
        seed words the off-target step works on (CRP_SCAN_SEEDS)."""
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        flags = (nat.SCAN_PRE if want_pre else 0) | (nat.SCAN_SEEDS if want_seeds else 0)
        self._check(nat.lib().crp_node_scan_score(self._h, int(guide_len), flags, ctypes.byref(a), ctypes.byref(b)),
                    "crp_node_scan_score")
        return a.value, b.value

    def offtarget(self, guide_len=20):
        """crp_node_offtarget: the genome-wide off-target seed scan over the node's resident tables (every device its own
        sites, histograms summed over the devices, every device its own hits' counts).  Returns the number of sites."""
        n = ctypes.c_uint64()
        self._check(nat.lib().crp_node_offtarget(self._h, int(guide_len), ctypes.byref(n)), "crp_node_offtarget")
        return n.value

    def annotate(self, request):
        """crp_node_annotate: the annotation join on every device (annotate.Request for the load()ed contigs)."""
        ann = request.annotation
        none = 0xFFFFFFFFFFFFFFFF
        seqids = np.array([ann.seq_index.get(name, none) for name in request.names], dtype=np.uint64)
        if seqids.size != self.n_contigs:
            raise ValueError("the request names %d contigs, the node holds %d" % (seqids.size, self.n_contigs))
        self._check(nat.lib().crp_node_annotate(self._h, ann._h, seqids.ctypes.data_as(nat.u64p), int(request.dec)),
                    "crp_node_annotate")

    def gather(self, root=0, pre=False, pos16=True, peer_copy=False, offtarget=False, features=False, to_host=False):
        """crp_node_gather: every device's owned rows to logical device `root` (one table per strand, contig order,
        contig-local positions); offtarget / features: the columns of offtarget() / annotate() travel too.  to_host:
        CRP_NODE_HOST_GATHER -- nothing crosses xGMI, fetch() pulls every device's rows over that device's own PCIe link
        into their place (for a consumer on the host).  Returns dict(ms_total, ms_exchange, bytes_to_root, transport)."""
        flags = ((nat.GATHER_PRE if pre else 0) | (nat.GATHER_POS16 if pos16 else 0) | (nat.NODE_PEER_COPY if peer_copy else 0) |
                 (nat.GATHER_OFFTARGET if offtarget else 0) | (nat.GATHER_FEATURES if features else 0) |
                 (nat.NODE_HOST_GATHER if to_host else 0))
        self._check(nat.lib().crp_node_gather(self._h, int(root), flags), "crp_node_gather")
        self._gathered = (bool(offtarget), bool(features))
        return self.gather_stats()

    def gather_stats(self):
        ms_t, ms_x, nb, tr = ctypes.c_double(), ctypes.c_double(), ctypes.c_uint64(), ctypes.c_int()
        self._check(nat.lib().crp_node_gather_stats(self._h, ctypes.byref(ms_t), ctypes.byref(ms_x), ctypes.byref(nb), ctypes.byref(tr)),
                    "crp_node_gather_stats")
        return dict(ms_total=ms_t.value, ms_exchange=ms_x.value, bytes_to_root=int(nb.value), transport=nat.TRANSPORTS[tr.value],
                    note=nat.lib().crp_node_transport_note(self._h).decode())  # (why RCCL is not / no longer in use: "" while it is)

    def counts(self):
        """(per_contig (n, 2) uint64, n_plus, n_minus) of the last gather."""
        per = np.zeros((self.n_contigs, 2), dtype=np.uint64)
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(nat.lib().crp_node_counts(self._h, per.ctypes.data_as(nat.u64p), ctypes.byref(a), ctypes.byref(b)),
                    "crp_node_counts")
        return per, a.value, b.value

    def count_scored(self):
        """Rows of the gathered tables that carry a real score (counted on the root device)."""
        n = ctypes.c_uint64()
        self._check(nat.lib().crp_node_count_scored(self._h, ctypes.byref(n)), "crp_node_count_scored")
        return n.value

    def device_counts(self, k):
        """dict(n_plus, n_minus, n_scored) of logical device k's OWN tables (rows inside its halos included), or None."""
        arenas = self._arenas(k)
        if not arenas:
            return None
        out = dict(n_plus=0, n_minus=0, n_scored=0)
        for a in arenas:
            x, y, z = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
            nat.check(nat.lib().crp_hits_counts(a, ctypes.byref(x), ctypes.byref(y)), "crp_hits_counts")
            nat.check(nat.lib().crp_count_scored(a, ctypes.byref(z)), "crp_count_scored")
            out["n_plus"] += x.value
            out["n_minus"] += y.value
            out["n_scored"] += z.value
        return out

    def arena_composition(self, k):
        arenas = self._arenas(k)
        if not arenas:
            return None
        out = dict(n_plain=0, n_other=0)
        for a in arenas:
            x, y = ctypes.c_uint64(), ctypes.c_uint64()
            nat.check(nat.lib().crp_arena_composition(a, ctypes.byref(x), ctypes.byref(y)), "crp_arena_composition")
            out["n_plain"] += x.value
            out["n_other"] += y.value
        return out

    def fetch(self, guide_len=20, out=None):
        """Host copies of the gathered tables -> NodeHits.  out: the NodeHits of an earlier fetch of the same size -- its
        arrays are filled again (a caller that processes genome after genome keeps its pages: a fresh numpy array costs a
        page fault per 4 KiB on first touch, several times what the link takes)."""
        per, n_plus, n_minus = self.counts()
        if out is not None and out.pos_plus.size == n_plus and out.pos_minus.size == n_minus:
            cols = [out.pos_plus, out.score_plus, out.pos_minus, out.score_minus]
        else:
            cols = [np.empty(n_plus, np.uint32), np.empty(n_plus, np.float64), np.empty(n_minus, np.uint32), np.empty(n_minus, np.float64)]
        self._check(nat.lib().crp_node_fetch(self._h, cols[0].ctypes.data_as(nat.u32p), cols[1].ctypes.data_as(nat.f64p),
                                             cols[2].ctypes.data_as(nat.u32p), cols[3].ctypes.data_as(nat.f64p)), "crp_node_fetch")
        hits = NodeHits(per, cols, guide_len)
        with_ot, with_feat = getattr(self, "_gathered", (False, False))
        if with_ot:
            hits.ot_plus, hits.ot_minus = np.empty((n_plus, 4), np.uint32), np.empty((n_minus, 4), np.uint32)
            self._check(nat.lib().crp_node_fetch_offtarget(self._h, hits.ot_plus.ctypes.data_as(nat.u32p),
                                                           hits.ot_minus.ctypes.data_as(nat.u32p)), "crp_node_fetch_offtarget")
        if with_feat:
            hits.feat_plus, hits.feat_minus = np.empty(n_plus, np.uint32), np.empty(n_minus, np.uint32)
            self._check(nat.lib().crp_node_fetch_features(self._h, hits.feat_plus.ctypes.data_as(nat.u32p),
                                                          hits.feat_minus.ctypes.data_as(nat.u32p)), "crp_node_fetch_features")
        return hits

    def scan(self, guide_len=20, root=0, pre=False, pos16=True, peer_copy=False, offtarget=False, annotation=None, to_host=False):
        """load()ed genome -> NodeHits: scan on every device (+ the opt-in off-target scan and annotation join over the
        resident tables), gatherv to `root`, tables to the host."""
        self.scan_score_device(guide_len, want_pre=pre, want_seeds=offtarget)
        if offtarget:
            self.offtarget(guide_len)
        if annotation is not None:
            self.annotate(annotation)
        self.gather(root, pre=pre, pos16=pos16, peer_copy=peer_copy, offtarget=offtarget, features=annotation is not None,
                    to_host=to_host)
        return self.fetch(guide_len)
