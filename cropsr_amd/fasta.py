"""FASTA text -> the contig table the reference iterates over.

Host-side mirror of import_fasta_file (reference CROPSR.py:54-74) and the two
helpers it calls, formatted (cropsr_functions.py:221-229) and
generate_dictionary (cropsr_functions.py:190-196).  The reference does not parse
FASTA records: it prints a Python list of (header, sequence) tuples with str(),
splits that text on whitespace and pairs the tokens up.  Every coordinate and the
`chromosome` column of the output depend on the by-products of that:

  * keys look like  [('Chr01',   ('c2',   and values like  'ACGT...'),  'ACGT...')]
    -- the quote/paren/comma characters are part of the strings that get scanned
    (positions are therefore 1 higher than 0-based genome offsets, and the
    decoration leaks into long_sequence near contig ends);
  * a header containing blanks shifts the pairing;
  * a FASTA that is already "two lines per record, no final newline" skips the
    re-formatting and is split as is (keys keep their '>').

This module reproduces that table for any input by building the same printed
form.  (A streaming packer that avoids the printed copy for Gb genomes is the
f2 row of SURVEY.md section 8.)
"""
from itertools import zip_longest


def needs_formatting(text):
    """CROPSR.py:62-63: anything but 2 lines per record without a final newline."""
    return 2 * text.count(">") != text.count("\n") + 1


def printed_records(text):
    """The text cropsr_functions.formatted returns: str() of the record list.

    A record is everything between two '>' (empty pieces dropped); it is split
    once at its first newline into (header, body) and newlines are removed from
    both parts.  A record without any newline yields a 1-tuple.
    """
    records = []
    for piece in text.split(">"):
        if piece:
            records.append(tuple(part.replace("\n", "") for part in piece.split("\n", 1)))
    return str(records)


def contig_table(text):
    """dict name_token -> sequence_string, insertion-ordered (CROPSR.py:70-71).

    Tokens are paired (1st, 2nd), (3rd, 4th), ...; an odd token count leaves the
    last key with "" as its value; a repeated key keeps its first position and
    takes the later value (dict semantics).
    """
    if needs_formatting(text):
        text = printed_records(text)
    tokens = text.split()
    return dict(zip_longest(tokens[0::2], tokens[1::2], fillvalue=""))


def load(path):
    with open(path, "r") as f:
        return contig_table(f.read())


# ----------------------------------------------------------------------------
# Fast loader: the same table without printing a
# Python list of the whole genome.  For the common shape -- every header and every
# sequence is "plain" (printable ASCII without blank, quote or backslash) and every
# record has a header line and a body -- the printed form is predictable:
#     key   = "[('" + header + "',"      (first record)   "('" + header + "',"   (others)
#     value = "'" + body + "'),"         (not last)        "'" + body + "')]"     (last)
# and is built directly from the bytes of the file: by table_from_bytes_python with
# bytes methods (the executable specification), and by table_from_bytes through the
# native loader, which is what the CLI uses.  Anything else falls back to contig_table().
_PLAIN = bytes(c for c in range(33, 127) if c not in (0x27, 0x5C))


def _plain(b):
    return not b.translate(None, _PLAIN)


def _merge(pairs):
    merged = {}
    for k, v in pairs:  # dict semantics: first position, last value
        merged[k] = v
    return list(merged.items())


def table_from_bytes_python(data):
    """[(key str, contig string as bytes)] in dict order, equal to
    list(contig_table(data.decode()).items()) with the values encoded as ASCII.
    Pure-Python form of the fast path (bytes methods); kept as the specification
    the native loader is tested against."""
    formatted = 2 * data.count(b">") != data.count(b"\n") + 1
    fast = None
    if formatted:
        pieces = [p for p in data.split(b">") if p]
        recs = []
        ok = bool(pieces)
        for p in pieces:
            head, sep, body = p.partition(b"\n")
            if not sep:
                ok = False  # a record without a newline prints as a 1-tuple: pairing shifts
                break
            body = body.translate(None, b"\n")
            if not (_plain(head) and _plain(body)):
                ok = False
                break
            recs.append((head, body))
        if ok:
            fast = []
            last = len(recs) - 1
            for k, (head, body) in enumerate(recs):
                key = ("[('" if k == 0 else "('") + head.decode("ascii") + "',"
                fast.append((key, b"'" + body + (b"')]" if k == last else b"'),")))
    elif _plain(data.translate(None, b"\n")):
        tokens = data.split()
        fast = [(tokens[i].decode("ascii"), tokens[i + 1] if i + 1 < len(tokens) else b"")
                for i in range(0, len(tokens), 2)]
    if fast is None:
        table = contig_table(data.decode("utf-8", "surrogateescape"))
        return [(k, v.encode("ascii", "replace")) for k, v in table.items()]
    return _merge(fast)


def table_from_bytes(data, n_threads=None):
    """The same table through the native loader (crp_fasta_table, cropsr_amd/csrc/crp_fasta.cpp):
    one parallel pass over the file's bytes writes every contig string, already decorated,
    into one buffer; the values returned are memoryview slices of it (no per-contig copies).
    Inputs outside the native fast path take table_from_bytes_python."""
    import ctypes

    import numpy as np

    from . import _native as nat
    from .rows import default_threads
    L = nat.lib()
    n = len(data)
    src = np.frombuffer(data, dtype=np.uint8)
    n_threads = n_threads or default_threads()
    out_cap, rec_cap = n + 65536, 16384
    while True:
        out = np.empty(out_cap, dtype=np.uint8)
        recs = np.empty((rec_cap, 4), dtype=np.uint64)
        n_recs, out_len, plain = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int()
        st = L.crp_fasta_table(src.ctypes.data_as(nat.u8p), n, out.ctypes.data_as(nat.u8p), out_cap,
                               recs.ctypes.data_as(nat.u64p), rec_cap, ctypes.byref(n_recs), ctypes.byref(out_len),
                               ctypes.byref(plain), n_threads)
        if st == -6:  # CRP_ERR_CAPACITY: the needed sizes came back
            out_cap, rec_cap = max(out_cap, int(out_len.value)), max(rec_cap, int(n_recs.value))
            continue
        nat.check(st, "crp_fasta_table")
        break
    if not plain.value:
        return table_from_bytes_python(bytes(data))
    view = memoryview(out)
    src_view = memoryview(data)
    pairs = []
    for k, (h0, hl, v0, vl) in enumerate(recs[:n_recs.value].tolist()):
        key = ("[('" if k == 0 else "('") + bytes(src_view[h0:h0 + hl]).decode("ascii") + "',"
        pairs.append((key, view[v0:v0 + vl]))
    return _merge(pairs)


PARALLEL_READ_MIN = 64 << 20
READ_PIECE = 16 << 20


def read_text_bytes(path, n_threads=None):
    """The file's bytes as Python's text mode would hand them over (CROPSR.py:58 opens
    with 'r': universal newlines turn \r\n and \r into \n).

    Returns a bytes-like object (bytes, or a uint8 numpy array for a large file: count_byte() and
    table_from_bytes() take both).  A large file is read by several threads into ONE fresh buffer
    (os.preadv releases the GIL): what a single read() of a 1 GB genome spends its time on is the
    page faults of a destination nobody has touched, and those parallelise (6 -> 13 GB/s on the
    MI355X boxes)."""
    import os
    size = os.path.getsize(path)
    if size < PARALLEL_READ_MIN or not hasattr(os, "preadv"):
        with open(path, "rb") as f:
            data = f.read()
        if b"\r" in data:
            data = data.replace(b"\r\n", b"\n").replace(b"\r", b"\n")
        return data
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    from .rows import default_threads
    buf = np.empty(size, dtype=np.uint8)
    view = memoryview(buf)
    piece = READ_PIECE
    fd = os.open(path, os.O_RDONLY)
    try:
        def read_piece(a):
            b, got = min(size, a + piece), a
            while got < b:
                k = os.preadv(fd, [view[got:b]], got)
                if k <= 0:
                    raise EOFError("%s: shorter than its size says" % path)
                got += k
            return bool((buf[a:b] == 0x0D).any())
        with ThreadPoolExecutor(min(n_threads or default_threads(), 16)) as pool:
            has_cr = any(list(pool.map(read_piece, range(0, size, piece))))
    finally:
        os.close(fd)
    if has_cr:
        return buf.tobytes().replace(b"\r\n", b"\n").replace(b"\r", b"\n")
    return buf


def count_byte(data, ch):
    """data.count(ch) for what read_text_bytes returns (ch: a one-byte bytes object)."""
    if isinstance(data, (bytes, bytearray)):
        return data.count(ch)
    import numpy as np
    c, step = ch[0], 64 << 20
    return int(sum(np.count_nonzero(data[a:a + step] == c) for a in range(0, len(data), step)))


def load_bytes(path):
    return table_from_bytes(read_text_bytes(path))
