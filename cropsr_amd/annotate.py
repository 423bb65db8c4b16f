"""--annotate: the GFF / Phytozome annotation join the reference never performs.

The reference parses the GFF into a DataFrame (CROPSR.py:77-95, called at :375), drops it, and
writes the constant '' into the `features` column (CROPSR.py:466, :468); `-p` is only echoed in
the banner (:364).  A real join changes the CSV, so it is OPT-IN here and the default output
stays byte-identical.  Being absent from the reference, the join has no reference oracle
("parity unpinned"); its definition is this engine's, and tests/test_annotate.py checks it
against a brute-force restatement (oracle/annotate_oracle.py).

Definition.  A row of the CSV that has a cut site (the 12-field rows; cutsite = end_pos - 3,
CROPSR.py:155-158) gets, in `features`, the ';'-joined labels of every GFF row whose type is `gene`
or `CDS`, whose seqid is the contig's FASTA name and whose [start, end] (GFF: 1-based, closed)
contains the cut site's 1-based genome coordinate, in GFF file order, each label once.
  label = "<type>:<ID>"  (ID attribute; else Name, else Parent, else "."), and for a gene whose Name
  (or ID) is a locusName of the Phytozome annotation_info file: + "|<Best-hit-arabi-name>" +
  "|<arabi-defline>" (fields that are empty are left out).
Coordinates: the reference's positions index the string it scans; in the re-formatted path that
string starts with one decoration character, so genome coordinate (1-based) =
cutsite - dec + 1 with dec = 1 there and 0 for an unformatted FASTA.

Where the work is done.  This module only moves tables:
  * crp_annotation_build (csrc/crp_annotation.cpp, host code) reads the two files and, per seqid, cuts the
    coordinate axis at the interval ends into elementary intervals with a constant label set: one string per
    DISTINCT set, one id per interval;
  * crp_annotation_track lays those intervals out in ARENA positions for the texts a rank has loaded (whole contigs,
    or the pieces of cut contigs in a multi-GPU run);
  * crp_annotate_set_track / crp_annotate_lookup (csrc/crp_annotate.hip) map every kept hit of the resident hit
    tables to the id of the interval its cut site falls in -- on the GPU, one streaming pass;
  * crp_write_rows_ex prints string[id] into the `features` column.
"""
import ctypes

import numpy as np

from . import _native as nat
from . import fasta

NO_FEATURE = nat.NO_FEATURE


def contig_name(key):
    """FASTA name of a contig from the reference's dict key: "[('Chr01'," / "('c2',"
    in the re-formatted path, ">name" otherwise."""
    if key.startswith(">"):
        return key[1:]
    k = key[1:] if key.startswith("[") else key
    if k.startswith("('"):
        k = k[2:]
    if k.endswith("',"):
        k = k[:-2]
    return k


def _text_u8(path):
    """(uint8 array, length): the file as text mode hands its lines over; an empty file still gets an address."""
    data = fasta.read_text_bytes(path)
    n = len(data)
    if not isinstance(data, np.ndarray):
        data = np.frombuffer(data or b"\0", dtype=np.uint8)
    return data, n


class StringTable:
    """The distinct label-set strings in the form crp_write_rows_ex takes (blob + offsets); table[k] decodes one."""

    def __init__(self, blob, off):
        self.blob, self.off = blob, off

    def __len__(self):
        return int(self.off.size - 1)

    def __getitem__(self, k):
        return bytes(self.blob[int(self.off[k]):int(self.off[k + 1])]).decode("utf-8", "replace")


class Annotation:
    """The parsed GFF (+ annotation_info): label-set strings and per-seqid elementary intervals (native handle)."""

    def __init__(self, gff_path, phytozome_path=None):
        L = nat.lib()
        gff, n_gff = _text_u8(gff_path)
        info, n_info = _text_u8(phytozome_path) if phytozome_path else (None, 0)
        self._h = ctypes.c_void_p()
        nat.check(L.crp_annotation_build(gff.ctypes.data_as(nat.u8p), n_gff, info.ctypes.data_as(nat.u8p) if info is not None else None,
                                         n_info, ctypes.byref(self._h)), "crp_annotation_build")
        c = [ctypes.c_uint64() for _ in range(5)]
        nat.check(L.crp_annotation_stats(self._h, *[ctypes.byref(x) for x in c]), "crp_annotation_stats")
        n_seq, n_str, n_blob, self.n_genes, self.n_cds = [int(x.value) for x in c]
        blob = np.zeros(max(1, n_blob), dtype=np.uint8)
        off = np.zeros(n_str + 1, dtype=np.uint64)
        nat.check(L.crp_annotation_strings(self._h, blob.ctypes.data_as(nat.u8p), off.ctypes.data_as(nat.u64p)),
                  "crp_annotation_strings")
        self.strings = StringTable(blob, off)
        self.seq_index = {}
        for k in range(n_seq):
            name, _, _ = self._seq(k)
            self.seq_index[name] = k
        self.n_seqids = n_seq

    def _seq(self, k):
        name, points, ids = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        n_name, n = ctypes.c_uint64(), ctypes.c_uint64()
        nat.check(nat.lib().crp_annotation_seqid(self._h, k, ctypes.byref(name), ctypes.byref(n_name), ctypes.byref(points),
                                                 ctypes.byref(ids), ctypes.byref(n)), "crp_annotation_seqid")
        text = ctypes.string_at(name.value, n_name.value).decode("utf-8", "replace") if n_name.value else ""
        return text, (points.value, ids.value), int(n.value)

    def seq_track(self, name):
        """(points int64 -- 1-based genome coordinates, ascending --, ids uint32) of one seqid, or None: interval k =
        [points[k], points[k+1]) carries strings[ids[k]].  Copies (for tests and the oracle's host-side join)."""
        k = self.seq_index.get(name)
        if k is None:
            return None
        _, (p, i), n = self._seq(k)
        if n == 0:
            return np.empty(0, np.int64), np.empty(0, np.uint32)
        points = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_int64)), (n,)).copy()
        ids = np.ctypeslib.as_array(ctypes.cast(i, ctypes.POINTER(ctypes.c_uint32)), (n,)).copy()
        return points, ids

    def arena_track(self, entries, dec):
        """(points uint32, ids uint32) for crp_annotate_set_track: entries = [(FASTA name, index of the text's first
        character inside its contig string, length of the text, arena offset)] in arena order."""
        e = np.empty((len(entries), 4), dtype=np.uint64)
        none = np.uint64(0xFFFFFFFFFFFFFFFF)
        for r, (name, lo, length, off) in enumerate(entries):
            k = self.seq_index.get(name)
            e[r] = (none if k is None else k, lo, length, off)
        n = ctypes.c_uint64()
        L = nat.lib()
        st = L.crp_annotation_track(self._h, e.ctypes.data_as(nat.u64p), e.shape[0], int(dec), None, None, 0, ctypes.byref(n))
        if st not in (nat.CRP_OK, nat.CRP_ERR_CAPACITY):
            nat.check(st, "crp_annotation_track")
        points, ids = np.empty(n.value, dtype=np.uint32), np.empty(n.value, dtype=np.uint32)
        nat.check(L.crp_annotation_track(self._h, e.ctypes.data_as(nat.u64p), e.shape[0], int(dec), points.ctypes.data_as(nat.u32p),
                                         ids.ctypes.data_as(nat.u32p), n.value, ctypes.byref(n)), "crp_annotation_track")
        return points, ids

    def close(self):
        if self._h:
            nat.lib().crp_annotation_destroy(self._h)
            self._h = None

    def __del__(self):
        try:  # (at interpreter exit the module globals may already be gone)
            self.close()
        except Exception:
            pass


class Request:
    """What a backend needs to annotate the texts it scans: the Annotation, the FASTA name of the contig every text
    comes from, where the text starts inside that contig's string, and `dec`."""

    def __init__(self, annotation, names, dec, starts=None):
        """annotation: an Annotation, or a callable that returns one when it is first needed (the CLI builds it on a helper
        thread and the join only needs it once the scan is done)."""
        self._annotation, self.names, self.dec = annotation, list(names), int(dec)
        self.starts = [0] * len(self.names) if starts is None else [int(s) for s in starts]

    @property
    def annotation(self):
        if callable(self._annotation):
            self._annotation = self._annotation()
        return self._annotation

    def pieces(self, which, starts):
        """The same request for texts that are pieces of the contigs: which[k] = contig index of text k."""
        return Request(lambda: self.annotation, [self.names[q] for q in which], self.dec, starts)

    def track(self, layout):
        """layout: [(text index, arena offset, length)] of ONE arena in arena order -> (points, ids)."""
        return self.annotation.arena_track([(self.names[t], self.starts[t], ln, off) for t, off, ln in layout], self.dec)


def features_of(request, hits):
    """(string table, idx) for rows.ContigTable / ContigRows from a contig's hit dict that carries the device's
    feat_plus / feat_minus columns (rows in the reference's order: '+' hits, then '-' hits)."""
    return request.annotation.strings, np.concatenate([hits["feat_plus"], hits["feat_minus"]]).astype(np.uint32, copy=False)
