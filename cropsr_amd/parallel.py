"""Multi-GPU sharding of the scan/score path: one process per GPU.

The reference is single-process; its only hint of data parallelism is the dead
`parallelize` helper (cropsr_functions.py:256-273: split rows over cores, no
exchange).  Contigs are independent in the hot loop (CROPSR.py:409 carries no
state but the append-only list), so the path shards by contig:

  * partition_contigs: longest-processing-time greedy assignment of contigs to
    ranks by length;
  * cut_contigs / piece_view / stitch_pieces: a contig longer than its fair share is
    cut into pieces that carry HALO characters of context either side, each hit is
    owned by the piece its match index falls in, and the pieces of a contig are
    stitched back in order -- no halo EXCHANGE is needed, every rank reads its
    piece (with halo) from the FASTA it already has;
  * every rank scans/scores its own arena -- no collective on the data path;
  * gather_hit_tables: the one exchange step, a gatherv of the per-rank hit
    tables to the root.  RCCL has no gatherv primitive: an all-gather of the
    counts, then grouped point-to-point send/recv (ncclGroupStart/End through
    torch.distributed.batch_isend_irecv).  Each peer->root transfer rides its own
    xGMI link, so the step is bounded by one link, not by a ring.

torch.distributed is plumbing only (RCCL bootstrap, streams); backend "nccl" is
RCCL on ROCm, and the same code runs on "gloo" with CPU tensors for the tests.

Process set-up note: PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  A
process that uses both PyTorch and libcropsr_hip.so must `import torch` BEFORE the
first cropsr_amd.Engine is created, so that one runtime (torch's) serves both;
the other order leaves torch without a visible GPU.  Without PyTorch the library
simply uses the system ROCm runtime.
"""
import numpy as np

COLUMNS = ("pos_plus", "score_plus", "pos_minus", "score_minus")


def partition_contigs(lengths, world_size):
    """LPT greedy: owner rank per contig, deterministic (ties -> lower rank)."""
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
    load = [0] * world_size
    owner = [0] * len(lengths)
    for k in order.tolist():
        b = min(range(world_size), key=lambda j: (load[j], j))
        owner[k] = b
        load[b] += int(lengths[k])
    return owner


# Everything the reference looks at around a match lies within 58 characters of it for guide
# lengths up to 50 (keep-filter: l + 5 behind, l + 8 ahead; the 30-character scoring window),
# so 128 characters of context make a piece's owned hits identical to the whole contig's.
HALO = 128


def cut_contigs(lengths, world_size, max_piece=None):
    """[(contig index, start, end)] covering every contig: a contig longer than max_piece
    (default: total / world_size, at least 4096) is cut into equal pieces of at most that length."""
    total = int(sum(int(n) for n in lengths))
    if max_piece is None:
        max_piece = max(4096, -(-total // max(1, world_size)))
    pieces = []
    for k, n in enumerate(lengths):
        n = int(n)
        parts = max(1, -(-n // max_piece))
        step = -(-n // parts) if parts > 1 else n
        for q in range(parts):
            pieces.append((k, q * step, min(n, (q + 1) * step)))
    return pieces


def piece_view(contig, start, end):
    """(characters to scan, index of `start` inside them) for the piece [start, end) of a contig
    string (bytes-like, one byte per character): the piece plus HALO characters either side."""
    lo = max(0, start - HALO)
    return contig[lo:min(len(contig), end + HALO)], start - lo


def stitch_pieces(piece_hits):
    """Hit tables of ONE contig from the tables of its pieces.

    piece_hits: [(start, end, shift, hits)] in piece order, hits = dict(pos_plus, score_plus,
    pos_minus, score_minus[, pre_*]) with positions relative to the scanned characters of
    piece_view, shift = the second value piece_view returned.  A hit belongs to the piece whose
    [start, end) contains its match index (the regex match position of CROPSR.py:98-104)."""
    out = {}
    for strand in ("plus", "minus"):
        cols = {c: [] for c in ("pos", "score", "pre")}
        for start, end, shift, hits in piece_hits:
            pos = np.asarray(hits["pos_" + strand]).astype(np.int64) - shift + start  # contig coordinates
            own = (pos >= start) & (pos < end)
            cols["pos"].append(pos[own].astype(np.uint32))
            cols["score"].append(np.asarray(hits["score_" + strand])[own])
            if "pre_" + strand in hits:
                cols["pre"].append(np.asarray(hits["pre_" + strand])[own])
        out["pos_" + strand] = np.concatenate(cols["pos"]) if cols["pos"] else np.empty(0, np.uint32)
        out["score_" + strand] = np.concatenate(cols["score"]) if cols["score"] else np.empty(0)
        if cols["pre"]:
            out["pre_" + strand] = np.concatenate(cols["pre"])
    return out


class _DeviceArray:
    """Zero-copy view of library-owned HBM for torch (CUDA array interface v2)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def device_tables_as_tensors(arena, n_plus, n_minus):
    """torch views (no copy) of the hit tables crp_scan_score left in HBM.

    Positions are exposed as int32 (same bits as the library's uint32) because
    RCCL point-to-point has no unsigned 32-bit type in torch.  Valid until the
    next scan on this arena.
    """
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("torch sees no GPU: import torch before creating the cropsr_amd Engine "
                           "(see the process set-up note in cropsr_amd/parallel.py)")
    pp, sp, pm, sm = arena.device_tables()
    dev = torch.device("cuda", torch.cuda.current_device())

    def view(ptr, n, typestr, dtype):
        if n == 0:
            return torch.empty(0, dtype=dtype, device=dev)
        return torch.as_tensor(_DeviceArray(ptr, n, typestr), device=dev)

    return {"pos_plus": view(pp, n_plus, "<i4", torch.int32),
            "score_plus": view(sp, n_plus, "<f8", torch.float64),
            "pos_minus": view(pm, n_minus, "<i4", torch.int32),
            "score_minus": view(sm, n_minus, "<f8", torch.float64)}


class TableGather:
    """gatherv of per-rank hit tables to `dst`; receive buffers are kept between
    calls (same sizes every bench step)."""

    def __init__(self, dst=0, group=None):
        self.dst = dst
        self.group = group
        self._bufs = {}

    def __call__(self, tables):
        """tables: dict of 1-D tensors (COLUMNS).  Returns on dst a list (one entry
        per rank, in rank order) of such dicts, elsewhere None."""
        import torch
        import torch.distributed as dist
        rank = dist.get_rank(self.group)
        world = dist.get_world_size(self.group)
        if dist.get_backend(self.group) == "gloo" and tables["pos_plus"].is_cuda:
            # gloo has no device point-to-point: stage through host memory (rehearsals only)
            tables = {c: t.cpu() for c, t in tables.items()}
        dev = tables["pos_plus"].device
        counts = torch.tensor([tables["pos_plus"].numel(), tables["pos_minus"].numel()],
                              dtype=torch.int64, device=dev)
        all_counts = [torch.empty_like(counts) for _ in range(world)]
        dist.all_gather(all_counts, counts, group=self.group)
        all_counts = torch.stack(all_counts).cpu().tolist()
        ops = []
        out = None
        if rank == self.dst:
            out = []
            for r in range(world):
                if r == rank:
                    out.append(tables)
                    continue
                n = {"pos_plus": all_counts[r][0], "score_plus": all_counts[r][0],
                     "pos_minus": all_counts[r][1], "score_minus": all_counts[r][1]}
                got = {}
                for c in COLUMNS:
                    key = (r, c)
                    buf = self._bufs.get(key)
                    if buf is None or buf.numel() != n[c]:
                        buf = torch.empty(n[c], dtype=tables[c].dtype, device=dev)
                        self._bufs[key] = buf
                    got[c] = buf
                    if n[c]:
                        ops.append(dist.P2POp(dist.irecv, buf, self._global(r), group=self.group))
                out.append(got)
        else:
            for c in COLUMNS:
                if tables[c].numel():
                    ops.append(dist.P2POp(dist.isend, tables[c].contiguous(), self._global(self.dst),
                                          group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out

    def _global(self, group_rank):
        import torch.distributed as dist
        if self.group is None:
            return group_rank
        return dist.get_global_rank(self.group, group_rank)


def merge_gathered(gathered, owners, per_rank_contigs):
    """Reassemble per-contig tables in ORIGINAL contig order on the root.

    gathered[r]: dict of numpy arrays for rank r (arena positions);
    per_rank_contigs[r]: list of (contig index, arena offset, length) in the order
    rank r loaded them.  Returns {contig index: dict(pos_plus, score_plus,
    pos_minus, score_minus)} with contig-local positions -- '+' before '-' per
    contig exactly as the reference orders rows (CROPSR.py:417-434).
    """
    out = {}
    for r, contigs in enumerate(per_rank_contigs):
        t = gathered[r]
        pp = np.asarray(t["pos_plus"]).view(np.uint32)
        pm = np.asarray(t["pos_minus"]).view(np.uint32)
        for (k, off, ln) in contigs:
            a, b = np.searchsorted(pp, [off, off + ln])
            c, d = np.searchsorted(pm, [off, off + ln])
            out[k] = dict(pos_plus=pp[a:b] - np.uint32(off), score_plus=np.asarray(t["score_plus"])[a:b],
                          pos_minus=pm[c:d] - np.uint32(off), score_minus=np.asarray(t["score_minus"])[c:d])
    return out


def sharded_scan(backend, strings, guide_len, dst=0, group=None, max_piece=None):
    """The scan of `strings` (the same list on every rank) spread over the ranks of `group`.

    Contigs are cut into pieces of at most a rank's fair share (cut_contigs), the pieces are dealt
    to the ranks by LPT (partition_contigs), every rank scans its pieces -- with their halo, no
    exchange -- through backend.scan_tables, and the one exchange of the path, the gatherv of the
    per-rank hit tables (TableGather: RCCL on GPUs), brings them to `dst`, where the pieces of each
    contig are stitched back.  Returns on `dst` what backend.scan(strings, guide_len) returns on
    one GPU -- a list of hit dicts, one per contig, bit for bit -- and None on the other ranks.

    backend.scan_tables(texts, guide_len) -> (tables, layout, release): tables = dict of the four
    COLUMNS as 1-D torch tensors (positions are arena positions), layout = [(offset, length)] of
    every text in that arena, release() frees what the tensors view."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    pieces = cut_contigs([len(s) for s in strings], world, max_piece)
    owner = partition_contigs([e - s for _, s, e in pieces], world)
    mine = [q for q, o in enumerate(owner) if o == rank]
    views = [piece_view(strings[pieces[q][0]], pieces[q][1], pieces[q][2]) for q in mine]
    tables, layout, release = backend.scan_tables([v for v, _ in views], guide_len)
    mine_layout = [(q, int(off), int(ln)) for q, (off, ln) in zip(mine, layout)]
    layouts = [None] * world
    dist.all_gather_object(layouts, mine_layout, group=group)
    gathered = TableGather(dst, group)(tables)
    if rank != dst:
        release()
        return None
    as_numpy = [{c: t.cpu().numpy() for c, t in g.items()} for g in gathered]
    release()
    per_piece = merge_gathered(as_numpy, None, layouts)
    out = []
    for k in range(len(strings)):
        parts = []
        for q, (kk, start, end) in enumerate(pieces):
            if kk == k:
                parts.append((start, end, start - max(0, start - HALO), per_piece[q]))
        out.append(stitch_pieces(parts))
    return out

