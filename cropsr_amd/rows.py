"""Hit tables -> CSV rows, byte-for-byte as the reference writes them.

Host-side mirror of the part of CROPSR.py that sits BELOW the accelerated path:
the row fields built at CROPSR.py:418-434, the id generator (:316-318, :448-449),
the 1 000 000-row chunked scorer/writer (:442-474) and the csv dialect.  The hit
positions and scores come from the HIP engine (cropsr_amd.engine); the two
string columns are re-sliced here from the contig string, which keeps every
string quirk of the reference without shipping strings through the GPU.

Reference behaviours preserved on purpose:
  B.3  Complete_dataset is never cleared: pass k re-emits contigs 1..k;
  B.4  the final partial chunk is taken from index count*counter, not
       1e6*counter; a total that is an exact multiple of 1e6 loses its last chunk;
  B.5  ids are consumed backwards (ids[index_range-index-1]);
  B.6  rows whose long_sequence is not 30 characters have 11 fields and -1;
  B.7  CRLF line ends, minimal quoting, shortest-repr floats.
"""
import csv
import ctypes
import os

import numpy as np

HEADER = ["crispr_id", "crispr_sys", "sequence", "long_sequence", "chromosome", "start_pos",
          "end_pos", "cutsite", "strand", "on_site_score", "features", "status"]
OFFTARGET_HEADER = ["offtarget_seed_mm0", "offtarget_seed_mm1", "offtarget_seed_mm2", "offtarget_seed_mm3"]  # --offtarget
NO_FEATURE = 0xFFFFFFFF
CHUNK = 1000000  # CROPSR.py:453
ORDER_BODY4, ORDER_TAIL2, ORDER_DOT1 = 0, 1, 2  # include/cropsr_hip.h CRP_ORDER_*


def _chain_map(steps):
    """Character map equivalent to a chain of str.replace(a, b) calls."""
    table = {}
    for c in set("".join(a + b for a, b in steps)):
        out = c
        for a, b in steps:
            if out == a:
                out = b
        if out != c:
            table[ord(c)] = out
    return table


# get_gRNA_sequence (CROPSR.py:128) and get_reverse_complement (:120), without their [::-1]
_RNA_STEPS = [("A", "U"), ("C", "Z"), ("G", "C"), ("Z", "G"), ("T", "A")]
_REVC_STEPS = _RNA_STEPS + [("U", "T")]
RNA_MAP = _chain_map(_RNA_STEPS)
_REVC_MAP = _chain_map(_REVC_STEPS)
# '-' strand: get_gRNA_sequence(get_reverse_complement(x)) -- the two reversals cancel
MINUS_MAP = {}
for _c in set(chr(k) for k in list(RNA_MAP) + list(_REVC_MAP)):
    _o = _c.translate(_REVC_MAP).translate(RNA_MAP)
    if _o != _c:
        MINUS_MAP[ord(_c)] = _o


def plus_text(s, a, b):
    """get_gRNA_sequence(s[a:b])."""
    return s[a:b].translate(RNA_MAP)[::-1]


def minus_text(s, a, b):
    """get_gRNA_sequence(get_reverse_complement(s[a:b]))."""
    return s[a:b].translate(MINUS_MAP)


class ContigRows:
    """The rows one contig contributes to Complete_dataset, in reference order
    (all '+' hits ascending, then all '-' hits ascending; CROPSR.py:417-434)."""

    def __init__(self, name_token, s, hits, guide_len, features=None):
        """features (opt-in, --annotate): (strings, idx) -- row k's `features` column is
        strings[idx[k]] (NO_FEATURE: ''); hits may carry ot_plus / ot_minus (--offtarget)."""
        l = guide_len
        self.chrom = name_token[1:]  # CROPSR.py:422 chromosome[1::]
        self.features = features
        self.ot = None
        if hits.get("ot_plus") is not None:
            self.ot = np.concatenate([hits["ot_plus"].reshape(-1, 4), hits["ot_minus"].reshape(-1, 4)]).astype(np.int64)
            self.ot[self.ot == 0xFFFFFFFF] = -1
        ip = hits["pos_plus"].astype(np.int64)
        jm = hits["pos_minus"].astype(np.int64)
        self.n = int(ip.size + jm.size)
        self.n_plus = int(ip.size)
        # '+': pam_location = (i-l, i)            stored [start, end]   (:418, :422)
        # '-': pam_location = (j+3, j+3+l)        stored [end', start'] (:429, :433)
        self.start = np.concatenate([ip - l, jm + 3 + l]).tolist()
        self.end = np.concatenate([ip, jm + 3]).tolist()
        self.score = np.concatenate([hits["score_plus"], hits["score_minus"]]).tolist()
        short, long_ = [], []
        for i in ip.tolist():
            short.append(plus_text(s, i - l, i))
            long_.append(plus_text(s, i - l - 5, i + 5))
        for j in jm.tolist():
            short.append(minus_text(s, j + 3, j + 3 + l))
            long_.append(minus_text(s, j + 3 - 5, j + 3 + l + 5))
        self.short = short
        self.long = long_

    def row(self, k, crispr_id):
        strand = "+" if k < self.n_plus else "-"
        extra = () if self.ot is None else tuple(int(v) for v in self.ot[k])
        if len(self.long[k]) == 30:  # CROPSR.py:466
            feat = ""
            if self.features is not None and self.features[1][k] != NO_FEATURE:
                feat = self.features[0][int(self.features[1][k])]
            return (crispr_id, "cas9", self.short[k], self.long[k], self.chrom, self.start[k],
                    self.end[k], self.end[k] - 3, strand, self.score[k], feat, "completed") + extra
        return (crispr_id, "cas9", self.short[k], self.long[k], self.chrom, self.start[k],
                self.end[k], strand, -1, "", "completed") + extra


def flush_plan(size, chunk=None):
    """(index_range, count) of every writerows() call CROPSR.py:451-474 makes for
    a dataset of `size` rows."""
    if chunk is None:
        chunk = CHUNK
    if size <= 0:
        return []
    n_full = (size + chunk - 1) // chunk - 1  # flushed while i < size-1
    plan = [(chunk * k, chunk) for k in range(n_full)]
    rest = size - n_full * chunk
    if rest < chunk:
        plan.append((rest * n_full, rest))  # index_range = count*counter with the CURRENT count
    return plan


def make_ids(size):
    """get_id (CROPSR.py:316-318) + the bytes->str step (:449); draws from the
    GLOBAL numpy RNG exactly as the reference does, so a seeded run reproduces."""
    alphanum = np.array(list("ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"), dtype="|U1")
    ids = np.random.choice(alphanum, [size, 7])
    return np.ascontiguousarray(ids).view("<U7").ravel()


class Dataset:
    """Complete_dataset (CROPSR.py:407): grows by one ContigRows per contig pass
    and is never cleared."""

    def __init__(self):
        self.blocks = []
        self.starts = [0]

    def append(self, block):
        self.blocks.append(block)
        self.starts.append(self.starts[-1] + block.n)

    def __len__(self):
        return self.starts[-1]

    def _locate(self, g):
        b = int(np.searchsorted(self.starts, g, "right") - 1)
        return self.blocks[b], g - self.starts[b]

    def rows(self, lo, count, ids, index_range, rescore):
        """Row tuples for dataset[lo:lo+count] with ids[index_range-index-1].

        The slice is one rs1_score batch (CROPSR.py:456-461).  The reference's BLAS
        sums the rows at the tail of a batch in another order than the body
        (include/cropsr_hip.h, CRP_ORDER_*); those <= 2 rows are re-scored through
        `rescore(rows_u8[n,30], order) -> scores` so the CSV matches to the last bit.
        """
        size = len(self)
        hi = min(lo + count, size)
        out = []
        b = int(np.searchsorted(self.starts, lo, "right") - 1)
        g = lo
        while g < hi:
            blk = self.blocks[b]
            base = self.starts[b]
            for k in range(g - base, min(hi - base, blk.n)):
                index = base + k - lo
                out.append(blk.row(k, ids[index_range - index - 1]))
            g = base + blk.n
            b += 1
        n = len(out)
        if n == 1:
            special, order = [0], ORDER_DOT1
        elif n % 4 >= 2:
            special, order = [4 * (n // 4), 4 * (n // 4) + 1], ORDER_TAIL2
        else:
            special, order = [], ORDER_BODY4
        special = [k for k in special if len(out[k][3]) == 30]  # scored rows only
        if special:
            seqs = np.empty((len(special), 30), dtype=np.uint8)
            for r, k in enumerate(special):
                # the scoring string of CROPSR.py:458
                seqs[r] = np.frombuffer(out[k][3].replace("U", "T").upper().encode("ascii", "replace"), dtype=np.uint8)
            fixed = rescore(seqs, order)
            for r, k in enumerate(special):
                row = list(out[k])
                row[9] = float(fixed[r])
                out[k] = tuple(row)
        return out


def write_header(path, offtarget=False):
    """CROPSR.py:402-405 (plus the four opt-in off-target column names)."""
    with open(path, "w", newline="") as f:
        csv.writer(f).writerow(HEADER + (OFFTARGET_HEADER if offtarget else []))


def write_pass(path, dataset, rescore):
    """One contig pass of CROPSR.py:442-474: fresh ids for the WHOLE dataset, then
    the chunk walk.  `rescore`: see Dataset.rows."""
    size = len(dataset)
    with open(path, "a", newline="") as f:
        writer = csv.writer(f)
        ids = make_ids(size)
        for index_range, count in flush_plan(size):
            writer.writerows(dataset.rows(index_range, count, ids, index_range, rescore))


# ----------------------------------------------------------------------------
# Native output path: the same bytes without a Python object per
# row.  ContigTable keeps a contig's rows as arrays; NativeDataset formats each
# written chunk with crp_format_rows (cropsr_amd/csrc/crp_format.cpp).  The tuple
# path above stays as the executable specification the native one is tested against.
class ContigTable:
    """Array form of ContigRows: same rows, same order, no per-row Python objects."""

    def __init__(self, name_token, s, hits, guide_len, features=None):
        """s: the contig string, as str or (one byte per character) bytes; features / ot_*: as in
        ContigRows (the opt-in columns)."""
        self.guide_len = guide_len
        self.feat_blob = self.feat_off = self.feat_idx = None
        if features is not None:
            if hasattr(features[0], "blob"):  # annotate.StringTable: already the form the writer takes (shared, not copied)
                self.feat_blob, self.feat_off = features[0].blob, features[0].off
            else:
                enc = [t.encode("utf-8") for t in features[0]]
                self.feat_blob = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8)
                self.feat_off = np.concatenate([[0], np.cumsum([len(e) for e in enc])]).astype(np.uint64)
            self.feat_idx = np.ascontiguousarray(features[1], dtype=np.uint32)
        self.ot = None
        if hits.get("ot_plus") is not None:
            self.ot = np.ascontiguousarray(np.concatenate([hits["ot_plus"].reshape(-1, 4), hits["ot_minus"].reshape(-1, 4)]),
                                           dtype=np.uint32)
        self.chrom = name_token[1:].encode("utf-8")
        self.chrom_u8 = np.frombuffer(self.chrom, dtype=np.uint8)  # (the pointer crp_write_segments reads the name through)
        self.text = np.frombuffer(s.encode("ascii", "replace") if isinstance(s, str) else s, dtype=np.uint8)
        self.n_plus = int(hits["pos_plus"].size)
        self.pos = np.ascontiguousarray(np.concatenate([hits["pos_plus"], hits["pos_minus"]]), dtype=np.uint32)
        self.minus = np.zeros(self.pos.size, dtype=np.uint8)
        self.minus[self.n_plus:] = 1
        self.score = np.ascontiguousarray(np.concatenate([hits["score_plus"], hits["score_minus"]]), dtype=np.float64)
        self.n = int(self.pos.size)

    def long_text(self, k):
        l, p = self.guide_len, int(self.pos[k])
        a, b = (p + 3 - 5, p + 3 + l + 5) if self.minus[k] else (p - l - 5, p + 5)
        piece = self.text[max(a, 0):max(b, 0)].tobytes().decode("latin-1")  # Python slice clamping
        return minus_text(piece, 0, len(piece)) if self.minus[k] else plus_text(piece, 0, len(piece))


def default_threads():
    """Formatter threads: the CPUs this process may run on, at most 32."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(32, n))


class NativeDataset:
    """Complete_dataset over ContigTables + the chunk formatter."""

    def __init__(self, n_threads=None):
        self.blocks = []
        self.starts = [0]
        self.n_threads = n_threads or default_threads()

    def append(self, block):
        self.blocks.append(block)
        self.starts.append(self.starts[-1] + block.n)

    def __len__(self):
        return self.starts[-1]

    def _segments(self, lo, count, ids_u8, index_range, rescore, ids_rev=None):
        """dataset[lo:lo+count], one rs1_score batch of the reference, cut at contig borders:
        yields (block, pos, minus, score, ids) per piece (ids consumed backwards, tail rows
        re-scored: see Dataset.rows).  ids_rev, if given, is ids[::-1] as a contiguous array
        and is used instead of ids_u8: row r takes ids[index_range - r - 1], i.e. consecutive
        rows of ids_rev."""
        size = len(self)
        hi = min(lo + count, size)
        n = hi - lo
        if n <= 0:
            return
        if ids_rev is None:
            sel = ids_u8[index_range - np.arange(n) - 1]  # negative indices wrap like Python's
        else:
            m_ids = len(ids_rev)
            first = (m_ids - index_range) % m_ids
            if first + n <= m_ids:
                sel = ids_rev[first:first + n]
            else:
                sel = ids_rev[(first + np.arange(n)) % m_ids]
        if n == 1:
            special, order = [0], ORDER_DOT1
        elif n % 4 >= 2:
            special, order = [4 * (n // 4), 4 * (n // 4) + 1], ORDER_TAIL2
        else:
            special, order = [], ORDER_BODY4
        b = int(np.searchsorted(self.starts, lo, "right") - 1)
        g = lo
        while g < hi:
            blk = self.blocks[b]
            base = self.starts[b]
            k0, k1 = g - base, min(hi - base, blk.n)
            m = k1 - k0
            if m > 0:
                score = blk.score[k0:k1]
                fix = [(i - (g - lo)) for i in special if g - lo <= i < g - lo + m]
                fix = [j for j in fix if len(blk.long_text(k0 + j)) == 30]
                if fix:
                    score = score.copy()
                    seqs = np.empty((len(fix), 30), dtype=np.uint8)
                    for r, j in enumerate(fix):
                        t = blk.long_text(k0 + j).replace("U", "T").upper()  # CROPSR.py:458
                        seqs[r] = np.frombuffer(t.encode("ascii", "replace"), dtype=np.uint8)
                    score[fix] = rescore(seqs, order)
                extras = (None if blk.feat_idx is None else blk.feat_idx[k0:k1], None if blk.ot is None else blk.ot[k0:k1])
                yield blk, blk.pos[k0:k1], blk.minus[k0:k1], score, np.ascontiguousarray(sel[g - lo:g - lo + m]), extras
            g = base + blk.n
            b += 1

    def chunk_bytes(self, lo, count, ids_u8, index_range, rescore, ids_rev=None):
        """CSV bytes of one written chunk (crp_format_rows)."""
        from . import _native as nat
        L = nat.lib()
        out = []
        for blk, pos, minus, score, ids_part, extras in self._segments(lo, count, ids_u8, index_range, rescore, ids_rev):
            if extras[0] is not None or extras[1] is not None:
                raise ValueError("chunk_bytes formats the reference's columns only (use chunk_to_fd)")
            m = pos.size
            cap = m * (170 + 2 * len(blk.chrom)) + 64
            while True:
                buf = np.empty(cap, dtype=np.uint8)
                used = ctypes.c_uint64()
                st = L.crp_format_rows(
                    blk.text.ctypes.data_as(nat.u8p), blk.text.size,
                    ctypes.cast(ctypes.c_char_p(blk.chrom), nat.u8p), len(blk.chrom), blk.guide_len,
                    pos.ctypes.data_as(nat.u32p), minus.ctypes.data_as(nat.u8p),
                    score.ctypes.data_as(nat.f64p), ids_part.ctypes.data_as(nat.u8p), m,
                    buf.ctypes.data_as(nat.u8p), cap, ctypes.byref(used), self.n_threads)
                if st == -6 and used.value > cap:  # CRP_ERR_CAPACITY: retry with the size it asked for
                    cap = int(used.value)
                    continue
                nat.check(st, "crp_format_rows")
                break
            out.append(buf[:used.value].tobytes())
        return b"".join(out)

    def chunk_segments(self, segs, keep, lo, count, ids_u8, index_range, rescore, ids_rev=None):
        """The same chunk as crp_row_segment entries appended to `segs` (for write_segments); every array a segment points
        into is appended to `keep` and must outlive the call that writes them."""
        from . import _native as nat
        for blk, pos, minus, score, ids_part, (feat_idx, ot) in self._segments(lo, count, ids_u8, index_range, rescore, ids_rev):
            g = nat.RowSegment()
            score = np.ascontiguousarray(score)
            keep.extend((blk, pos, minus, score, ids_part))
            g.contig_text, g.contig_len = blk.text.ctypes.data, blk.text.size
            g.chrom, g.chrom_len = (blk.chrom_u8.ctypes.data if len(blk.chrom) else None), len(blk.chrom)
            g.pos, g.minus, g.score, g.ids, g.n_rows = pos.ctypes.data, minus.ctypes.data, score.ctypes.data, ids_part.ctypes.data, pos.size
            if feat_idx is not None:
                feat_idx = np.ascontiguousarray(feat_idx)
                keep.append(feat_idx)
                g.feat_blob, g.feat_off, g.feat_idx = blk.feat_blob.ctypes.data, blk.feat_off.ctypes.data, feat_idx.ctypes.data
            if ot is not None:
                ot = np.ascontiguousarray(ot)
                keep.append(ot)
                g.offtarget = ot.ctypes.data
            segs.append(g)

    def chunk_to_fd(self, fd, lo, count, ids_u8, index_range, rescore, ids_rev=None):
        """The same chunk appended to file descriptor fd (crp_write_segments); returns the byte count."""
        segs, keep = [], []
        self.chunk_segments(segs, keep, lo, count, ids_u8, index_range, rescore, ids_rev)
        return write_segments(fd, segs, self.blocks[0].guide_len if self.blocks else 20, self.n_threads)


def write_segments(fd, segs, guide_len, n_threads):
    """crp_write_segments: the rows of all `segs` (RowSegment entries, in order) appended to fd by one team of formatter
    threads; returns the byte count.  The caller keeps the arrays the segments point into alive."""
    from . import _native as nat
    if not segs:
        return 0
    arr = (nat.RowSegment * len(segs))(*segs)
    written = ctypes.c_uint64()
    st = nat.lib().crp_write_segments(fd, guide_len, ctypes.cast(arr, ctypes.c_void_p), len(segs), ctypes.byref(written), n_threads)
    if st == nat.CRP_ERR_IO:
        err = ctypes.get_errno()
        raise OSError(err, "crp_write_segments: " + os.strerror(err))
    nat.check(st, "crp_write_segments")
    return written.value


def ids_as_bytes(ids_u1):
    """(size,7) '<U1' array from np.random.choice -> (size,7) uint8."""
    a = np.ascontiguousarray(ids_u1)
    return a.view(np.uint32).astype(np.uint8).reshape(a.shape[0], 7) if a.size else np.empty((0, 7), np.uint8)


_ID_LUT = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789", dtype=np.uint8)


def draw_ids(size, piece=1 << 20, reverse=False):
    """(size, 7) uint8 crispr ids of one write pass (CROPSR.py:316-318); with reverse=True the
    same ids last-first (the order the reference consumes them in, CROPSR.py:448-449).

    np.random.choice(alphanum, [size, 7]) IS alphanum[np.random.randint(0, 36, [size, 7])]
    (legacy RandomState.choice, uniform, with replacement), and randint is one masked-rejection
    draw per character on the global MT19937.  crp_legacy_ids makes exactly those draws on the
    state taken from numpy and hands the advanced state back: same characters, same global RNG
    state afterwards (tests/test_format.py), about five times faster, and no UCS-4 string array or
    size x 7 int64 array is ever built.  (Should numpy's global generator ever not be MT19937,
    the draws fall back to np.random.randint in pieces of rows.)"""
    out = np.empty((size, 7), dtype=np.uint8)
    if size == 0:
        return out
    state = np.random.get_state(legacy=True)
    if state[0] == "MT19937":
        from . import _native as nat
        key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
        pos = ctypes.c_int32(int(state[2]))
        nat.check(nat.lib().crp_legacy_ids(key.ctypes.data_as(nat.u32p), ctypes.byref(pos), out.ctypes.data_as(nat.u8p),
                                           size, int(bool(reverse))), "crp_legacy_ids")
        np.random.set_state(("MT19937", key, pos.value, state[3], state[4]))
        return out
    for lo in range(0, size, piece):
        m = min(piece, size - lo)
        draws = np.random.randint(0, 36, size=[m, 7])
        if reverse:
            np.take(_ID_LUT, draws[::-1], out=out[size - lo - m:size - lo])
        else:
            np.take(_ID_LUT, draws, out=out[lo:lo + m])
    return out


class IdStream:
    """The ids of consecutive write passes, drawn ahead of their use on a worker thread.

    The passes of one run draw from one RNG stream in a fixed order and their sizes are known
    once the scan is done, so the ids of the next passes are drawn while a pass is formatted and written: at least
    `depth` passes ahead, and as many more as fit `rows_ahead` rows (a run of short contigs is drawn while the long one
    in front of it is written).  Nothing else may touch np.random while the stream is open."""

    def __init__(self, sizes, depth=2, reverse=False, rows_ahead=8_000_000):
        import collections
        import threading
        self._sizes = list(sizes)
        self._q = collections.deque()
        self._rows = 0
        self._cv = threading.Condition()
        self._stop = False

        def work():
            try:
                for size in self._sizes:
                    with self._cv:
                        self._cv.wait_for(lambda: self._stop or len(self._q) < depth or self._rows + size <= rows_ahead)
                        if self._stop:
                            return
                    item = (size, draw_ids(size, reverse=reverse))
                    with self._cv:
                        self._q.append(item)
                        self._rows += size
                        self._cv.notify_all()
            except BaseException as e:  # handed to the consumer
                with self._cv:
                    self._q.append((None, e))
                    self._cv.notify_all()

        self._thread = threading.Thread(target=work, daemon=True)
        self._thread.start()

    def next(self, size):
        with self._cv:
            self._cv.wait_for(lambda: len(self._q) > 0)
            got, ids = self._q.popleft()
            if got is not None:
                self._rows -= got
            self._cv.notify_all()
        if got is None:
            raise ids
        if got != size:
            raise RuntimeError("IdStream: pass of %d rows, ids drawn for %d" % (size, got))
        return ids

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._thread.join()


def write_pass_native(path, dataset, rescore, ids=None):
    """write_pass with the native formatter: same RNG draws, same chunk walk, same bytes -- every chunk of the pass in ONE call
    of the formatter.  `ids` (an IdStream built with reverse=True) supplies ids drawn ahead; by default they are drawn here."""
    size = len(dataset)
    ids_rev = draw_ids(size, reverse=True) if ids is None else ids.next(size)
    return write_passes_native(path, [(dataset, ids_rev)], rescore)


def write_passes_native(path, passes, rescore):
    """Several write passes -- (dataset, ids_rev) each: the rows of the pass and its ids last-first (draw_ids(reverse=True)) --
    appended to `path` by ONE call of the native formatter: the bytes of write_pass_native pass after pass.  What the CLI does
    with a run of short contigs under --each-contig-once (a pass of 3 500 rows alone is formatted by one thread)."""
    segs, keep = [], []
    guide_len, n_threads = 20, 1
    for dataset, ids_rev in passes:
        size = len(dataset)
        if dataset.blocks:
            guide_len, n_threads = dataset.blocks[0].guide_len, dataset.n_threads
        for index_range, count in flush_plan(size):
            dataset.chunk_segments(segs, keep, index_range, count, None, index_range, rescore, ids_rev=ids_rev)
    with open(path, "ab") as f:
        return write_segments(f.fileno(), segs, guide_len, n_threads)
