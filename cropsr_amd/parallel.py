"""Multi-GPU sharding of the scan/score path: one process per GPU, no PyTorch.

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
  * every rank scans/scores its own arenas -- no collective on the data path;
  * the one exchange step is the gatherv of the per-rank hit tables to the root.  On
    GPUs it runs inside libcropsr_hip.so on RCCL (crp_gather_hits: all-gather of the
    counts, then grouped ncclSend/ncclRecv -- each peer->root transfer on its own xGMI
    link); gather_host is the same exchange over the control sockets of
    rendezvous.Group, for the CPU tests, for ranks that share one GPU, and for
    deployments that prefer every rank's own PCIe link (CROPSR_GATHER=host);
  * with the opt-in off-target scan every rank adds its own sites to a seed histogram
    and the histograms are summed over the ranks (RCCL all-reduce, crp_offtarget_reduce).

A backend (cli.EngineBackend, or the oracle in the tests) provides
scan_resident(texts, guide_len, offtarget=False) -> Resident   (offtarget: the seed scan will follow):
    .layout                          [(arena index, arena offset, length)] per text
    .offtarget(group, own_by_arena)  seed scan over all ranks' sites; own_by_arena: per arena the
                                     [begin, end) arena positions whose hits this rank owns
    .annotate(request)               annotation join over the resident tables (annotate.Request for this rank's texts)
    .gather(group, dst, offtarget[, features])  on dst: per rank, per arena, a column dict (numpy; arena
                                     positions); None elsewhere
    .release()
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


def split_evenly(lengths, world_size, min_piece=4096):
    """(pieces, owner): the contigs, in their order, dealt to the ranks as CONTIGUOUS runs of equal size -- rank r gets the
    characters [r, r + 1) * total / world of the concatenated genome; a contig that straddles a boundary is cut there
    (unless one side would be shorter than min_piece: then it stays whole on the side that holds most of it).  At most
    world - 1 cuts, shares equal to within min_piece -- where dealing whole chromosomes by LPT leaves 16 % (switchgrass-like)
    to 39 % (sorghum-like) on the busiest of 8 ranks."""
    total = int(sum(int(n) for n in lengths))
    bounds = [(r + 1) * total // world_size for r in range(world_size)]
    pieces, owner = [], []
    r, acc = 0, 0
    for k, n in enumerate(lengths):
        n, start = int(n), 0
        while True:
            rest = n - start
            room = bounds[r] - acc
            if r == world_size - 1 or rest <= room:
                pieces.append((k, start, n))
                owner.append(r)
                acc += rest
                break
            if room >= min_piece and rest - room >= min_piece:  # cut at the boundary
                pieces.append((k, start, start + room))
                owner.append(r)
                acc += room
                start += room
                r += 1
            elif 2 * room >= rest:  # a sliver would be left over: the rest of the contig stays here
                pieces.append((k, start, n))
                owner.append(r)
                acc += rest
                break
            else:                   # a sliver would be cut off: the next rank takes the contig from here
                r += 1
        while r < world_size - 1 and acc >= bounds[r]:
            r += 1
    return pieces, owner


def strong_plan(lengths, world_size, max_piece=None):
    """How ONE genome is spread over the ranks (BASELINE.json configs[3], [4]; bench.py's strong-scaling block and
    sharded_scan share it): {"pieces": [(contig, start, end)] in contig order, "owner": rank per piece, "by_rank": piece
    indices per rank in piece order, "bases": owned characters per rank}.  Default: split_evenly (equal contiguous
    shares).  max_piece given (CROPSR_DIST_MAX_PIECE): cut_contigs into pieces of at most that length, dealt by LPT."""
    if max_piece is None:
        pieces, owner = split_evenly(lengths, world_size)
    else:
        pieces = cut_contigs(lengths, world_size, max_piece)
        owner = partition_contigs([e - s for _, s, e in pieces], world_size)
    by_rank = [[q for q, o in enumerate(owner) if o == r] for r in range(world_size)]
    bases = [int(sum(pieces[q][2] - pieces[q][1] for q in qs)) for qs in by_rank]
    return {"pieces": pieces, "owner": owner, "by_rank": by_rank, "bases": bases}


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
        cols = {c: [] for c in ("pos", "score", "pre", "ot", "feat")}
        for start, end, shift, hits in piece_hits:
            pos = np.asarray(hits["pos_" + strand]).astype(np.int64) - shift + start  # contig coordinates
            own = (pos >= start) & (pos < end)
            cols["pos"].append(pos[own].astype(np.uint32))
            cols["score"].append(np.asarray(hits["score_" + strand])[own])
            for extra in ("pre", "ot", "feat"):
                if hits.get(extra + "_" + strand) is not None:
                    cols[extra].append(np.asarray(hits[extra + "_" + strand])[own])
        out["pos_" + strand] = np.concatenate(cols["pos"]) if cols["pos"] else np.empty(0, np.uint32)
        out["score_" + strand] = np.concatenate(cols["score"]) if cols["score"] else np.empty(0)
        for extra in ("pre", "ot", "feat"):
            if cols[extra]:
                out[extra + "_" + strand] = np.concatenate(cols[extra])
    return out


def _table_keys(offtarget, features=False):
    return COLUMNS + (("ot_plus", "ot_minus") if offtarget else ()) + (("feat_plus", "feat_minus") if features else ())


def pack_pos16(pos):
    """An ascending uint32 position table as (lo16, bstart): the low 16 bits of every position and, per 65 536 positions,
    the index of the first entry at or after it -- what crosses the links with CRP_GATHER_POS16 (crp_gather.hip; here in
    numpy for the socket transport): exact, 2 B per hit + 4 B per 65 536 positions instead of 4 B per hit."""
    pos = np.ascontiguousarray(pos, dtype=np.uint32)
    n_buckets = (int(pos[-1]) >> 16) + 1 if pos.size else 1
    needles = (np.arange(n_buckets, dtype=np.uint64) << np.uint64(16)).astype(np.uint32)  # (the tables' own dtype: no conversion)
    return pos.astype(np.uint16), np.searchsorted(pos, needles, "left").astype(np.uint32)


def unpack_pos16(lo16, bstart):
    lo16 = np.asarray(lo16, dtype=np.uint16)
    bstart = np.asarray(bstart, dtype=np.uint32).astype(np.int64)
    per_bucket = np.diff(np.append(bstart, lo16.size))
    high = np.repeat(np.arange(bstart.size, dtype=np.uint32) << np.uint32(16), per_bucket)
    return high | lo16.astype(np.uint32)


def gather_host(group, arenas_cols, dst=0, offtarget=False, features=False, pos16=True):
    """gatherv over the control sockets: arenas_cols = this rank's column dicts (numpy), one per
    arena.  Returns on dst [rank][arena] -> column dict, None elsewhere.  pos16: the position columns travel
    packed (pack_pos16), like CRP_GATHER_POS16 on RCCL.  group.bytes_gathered (on dst) = what the peers sent."""
    if dst != 0:
        raise ValueError("the host transport gathers to rank 0")
    keys = _table_keys(offtarget, features)
    n_arenas = group.all_gather(len(arenas_cols))
    if group.rank != dst:
        for cols in arenas_cols:
            for k in keys:
                if pos16 and k.startswith("pos_"):
                    for part in pack_pos16(cols[k]):
                        group.send_array(part)
                else:
                    group.send_array(cols[k])
        return None
    moved = 0

    def recv(r, k):
        nonlocal moved
        if pos16 and k.startswith("pos_"):
            lo16, bstart = group.recv_array(r), group.recv_array(r)
            moved += lo16.nbytes + bstart.nbytes
            return unpack_pos16(lo16, bstart)
        a = group.recv_array(r)
        moved += a.nbytes
        return a

    out = []
    for r in range(group.world):
        if r == dst:
            out.append([{k: np.asarray(cols[k]) for k in keys} for cols in arenas_cols])
        else:
            out.append([{k: recv(r, k) for k in keys} for _ in range(n_arenas[r])])
    group.bytes_gathered = moved
    return out


def slice_piece(cols, off, ln):
    """The rows of one text out of an arena's tables (arena positions -> positions in the text)."""
    pp = np.asarray(cols["pos_plus"]).view(np.uint32)
    pm = np.asarray(cols["pos_minus"]).view(np.uint32)
    # (needles in the tables' own dtype: with int64 needles numpy converts the WHOLE table for every call -- 10 ms per
    # piece on a 14 M-row table, 8 s for the 868 pieces of the sorghum-like genome)
    bounds = np.array([off, off + ln], dtype=np.uint32)
    a, b = np.searchsorted(pp, bounds)
    c, d = np.searchsorted(pm, bounds)
    out = dict(pos_plus=pp[a:b] - np.uint32(off), score_plus=np.asarray(cols["score_plus"])[a:b],
               pos_minus=pm[c:d] - np.uint32(off), score_minus=np.asarray(cols["score_minus"])[c:d])
    if "ot_plus" in cols:
        out["ot_plus"], out["ot_minus"] = np.asarray(cols["ot_plus"])[a:b], np.asarray(cols["ot_minus"])[c:d]
    if "feat_plus" in cols:
        out["feat_plus"], out["feat_minus"] = np.asarray(cols["feat_plus"])[a:b], np.asarray(cols["feat_minus"])[c:d]
    return out


def merge_gathered(gathered, layouts):
    """{text index: hit dict with text-local positions} from what the root gathered.

    gathered[r][a]: column dict of rank r's arena a (arena positions); layouts[r]: [(text index,
    arena index, arena offset, length)] in the order rank r loaded its texts."""
    out = {}
    for r, layout in enumerate(layouts):
        for (q, a, off, ln) in layout:
            out[q] = slice_piece(gathered[r][a], int(off), int(ln))
    return out


def sharded_scan(backend, strings, guide_len, group, dst=0, max_piece=None, offtarget=False, annotation=None):
    """The scan of `strings` (the same list on every rank) spread over the ranks of `group`.

    The contigs are dealt to the ranks in equal contiguous shares, cut where a share ends inside one (strong_plan; with
    max_piece: cut into pieces of at most that length and dealt by LPT), every rank scans its pieces -- with their halo, no
    exchange -- and the one exchange of the path, the gatherv of the per-rank hit tables, brings
    them to `dst`, where the pieces of each contig are stitched back.  Returns on `dst` what
    backend.scan(strings, guide_len) returns on one GPU -- a list of hit dicts, one per contig,
    bit for bit -- and None on the other ranks.

    annotation (annotate.Request for `strings`, the same on every rank): every rank also joins ITS resident tables with
    the annotation track of its pieces (Resident.annotate) and the label-set ids travel with the tables; the hit dicts
    then carry feat_plus / feat_minus.

    A rank that fails before the exchange (a share that does not fit its GPU, a HIP error, an annotation
    that cannot be built or joined) reports it through group.check, so EVERY rank raises rendezvous.RankError
    with the same message instead of waiting in a collective."""
    rank, world = group.rank, group.world
    plan = strong_plan([len(s) for s in strings], world, max_piece)
    pieces, mine = plan["pieces"], plan["by_rank"][rank]
    res, err = None, None
    try:
        views = [piece_view(strings[pieces[q][0]], pieces[q][1], pieces[q][2]) for q in mine]
        res = backend.scan_resident([v for v, _ in views], guide_len, offtarget=offtarget)
    except Exception as e:  # reported to every rank below
        err = "%s: %s" % (type(e).__name__, e)
    group.check(err)
    try:
        if offtarget:
            # a hit counts as a site only in the piece that owns it (not in a neighbour's halo)
            n_arenas = 1 + max([a for a, _, _ in res.layout], default=-1)
            own = [[] for _ in range(n_arenas)]
            for q, (v, shift), (a, off, _ln) in zip(mine, views, res.layout):
                own[a].append((int(off) + shift, int(off) + shift + pieces[q][2] - pieces[q][1]))
            res.offtarget(group, [sorted(o) for o in own])
        if annotation is not None:
            # What can fail on ONE rank here -- the lazy build of the Annotation (a GFF that does not parse), the track of
            # this rank's pieces, HBM for the track or the ids, a HIP error in the look-up -- is agreed on like the scan's
            # errors are: a rank that raised alone would leave its peers in the gatherv below.
            err = None
            try:
                # a piece's text starts `shift` characters before the piece (its halo): index of its first character
                res.annotate(annotation.pieces([pieces[q][0] for q in mine], [pieces[q][1] - shift for q, (_, shift) in zip(mine, views)]))
            except Exception as e:
                err = "annotation join: %s: %s" % (type(e).__name__, e)
            group.check(err)
        layouts = group.all_gather([(q, int(a), int(off), int(ln)) for q, (a, off, ln) in zip(mine, res.layout)])
        gathered = res.gather(group, dst, offtarget, features=True) if annotation is not None else res.gather(group, dst, offtarget)
    finally:
        res.release()
    if rank != dst:
        return None
    per_piece = merge_gathered(gathered, layouts)
    # pieces come out of cut_contigs in contig order: one pass groups them
    out, q = [], 0
    for k in range(len(strings)):
        parts = []
        while q < len(pieces) and pieces[q][0] == k:
            _, start, end = pieces[q]
            parts.append((start, end, start - max(0, start - HALO), per_piece[q]))
            q += 1
        out.append(stitch_pieces(parts))
    return out
