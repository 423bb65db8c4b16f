"""ctypes face of the CPU oracle (oracle/crp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build():
    """Compile liborc.so in place (gcc only)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_exp.restype = ctypes.c_double
        L.orc_exp.argtypes = [ctypes.c_double]
        L.orc_exp_many.argtypes = [_f64p, ctypes.c_int64, _f64p]
        L.orc_exp_table_path_many.argtypes = [_f64p, ctypes.c_int64, _f64p]
        L.orc_score30.argtypes = [_u8p, ctypes.c_int64, _f64p, _f64p]
        L.orc_score30_order.argtypes = [_u8p, ctypes.c_int64, ctypes.c_int, _f64p, _f64p]
        L.orc_rs1_batch.argtypes = [_u8p, ctypes.c_int64, _f64p, _f64p]
        L.orc_scan.argtypes = [_u8p, ctypes.c_int64, ctypes.c_int, _u32p, _i64p, _u32p, _i64p]
        L.orc_score_hits.argtypes = [_u8p, ctypes.c_int64, ctypes.c_int, _u32p, ctypes.c_int64,
                                     ctypes.c_int, _f64p, _f64p]
        L.orc_long_sequence.restype = ctypes.c_int64
        L.orc_long_sequence.argtypes = [_u8p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _u8p]
        L.orc_short_sequence.restype = ctypes.c_int64
        L.orc_short_sequence.argtypes = [_u8p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _u8p]
        L.orc_weights.argtypes = [_f64p, _f64p, _f64p]
        _LIB = L
    return _LIB


def _as_u8(s):
    if isinstance(s, str):
        s = s.encode("latin-1")
    a = np.frombuffer(s, dtype=np.uint8) if not isinstance(s, np.ndarray) else s
    return np.ascontiguousarray(a, dtype=np.uint8)


def _p(a, t):
    return a.ctypes.data_as(t)


def exp(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().orc_exp_many(_p(x, _f64p), x.size, _p(out, _f64p))
    return out


def exp_table_path(x):
    """exp through the table path only (what the HIP scorer runs), valid for |x| < 512."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().orc_exp_table_path_many(_p(x, _f64p), x.size, _p(out, _f64p))
    return out


def score30(rows):
    """Seam 2: rs1_score on an (n,30) uint8 array -> (pre, score)."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    assert rows.ndim == 2 and rows.shape[1] == 30
    n = rows.shape[0]
    pre = np.empty(n)
    score = np.empty(n)
    lib().orc_score30(_p(rows, _u8p), n, _p(pre, _f64p), _p(score, _f64p))
    return pre, score


def score30_order(rows, order):
    """Seam 2 with every row summed in BLAS order 0 (body), 1 (pair tail) or 2 (single row)."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n = rows.shape[0]
    pre = np.empty(n)
    score = np.empty(n)
    lib().orc_score30_order(_p(rows, _u8p), n, int(order), _p(pre, _f64p), _p(score, _f64p))
    return pre, score


def rs1_batch(rows):
    """rs1_score on a batch as the reference computes it (row order by batch position)."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n = rows.shape[0]
    pre = np.empty(n)
    score = np.empty(n)
    lib().orc_rs1_batch(_p(rows, _u8p), n, _p(pre, _f64p), _p(score, _f64p))
    return pre, score


def scan(s, l=20):
    """Kept hit positions of one contig string: (plus, minus) uint32 arrays."""
    a = _as_u8(s)
    npl = ctypes.c_int64()
    nmi = ctypes.c_int64()
    L = lib()
    L.orc_scan(_p(a, _u8p), a.size, l, None, ctypes.byref(npl), None, ctypes.byref(nmi))
    plus = np.empty(npl.value, dtype=np.uint32)
    minus = np.empty(nmi.value, dtype=np.uint32)
    L.orc_scan(_p(a, _u8p), a.size, l, _p(plus, _u32p), ctypes.byref(npl), _p(minus, _u32p), ctypes.byref(nmi))
    return plus, minus


def score_hits(s, pos, minus, l=20):
    a = _as_u8(s)
    pos = np.ascontiguousarray(pos, dtype=np.uint32)
    pre = np.empty(pos.size)
    score = np.empty(pos.size)
    lib().orc_score_hits(_p(a, _u8p), a.size, l, _p(pos, _u32p), pos.size, int(bool(minus)),
                         _p(pre, _f64p), _p(score, _f64p))
    return pre, score


def scan_score(s, l=20):
    """Whole seam 1+2 for one contig string.

    Returns dict(pos_plus, pre_plus, score_plus, pos_minus, pre_minus, score_minus);
    rows with an incomplete 30-window carry pre = score = -1.
    """
    a = _as_u8(s)
    plus, minus = scan(a, l)
    pp, sp = score_hits(a, plus, False, l)
    pm, sm = score_hits(a, minus, True, l)
    return dict(pos_plus=plus, pre_plus=pp, score_plus=sp, pos_minus=minus, pre_minus=pm, score_minus=sm)


def long_sequence(s, pos, minus, l=20):
    a = _as_u8(s)
    out = np.empty(512, dtype=np.uint8)
    n = lib().orc_long_sequence(_p(a, _u8p), a.size, int(pos), int(bool(minus)), l, _p(out, _u8p))
    return out[:n].tobytes().decode("latin-1")


def short_sequence(s, pos, minus, l=20):
    a = _as_u8(s)
    out = np.empty(512, dtype=np.uint8)
    n = lib().orc_short_sequence(_p(a, _u8p), a.size, int(pos), int(bool(minus)), l, _p(out, _u8p))
    return out[:n].tobytes().decode("latin-1")


def weights():
    f = np.empty(120)
    s = np.empty(464)
    c = np.empty(2)
    lib().orc_weights(_p(f, _f64p), _p(s, _f64p), _p(c, _f64p))
    return f, s, c


# ---- off-target seed scan (no reference counterpart; see crp_oracle.c) -------------------------
NOT_A_SITE = 0xFFFFFFFF
N_SEEDS = 1 << 24


def _ot_bind():
    L = lib()
    if not getattr(L, "_ot_bound", False):
        L.orc_seed_codes.argtypes = [_u8p, ctypes.c_int64, ctypes.c_int, _u32p, ctypes.c_int64, ctypes.c_int, _u32p]
        L.orc_offtarget_pairs.argtypes = [_u32p, ctypes.c_int64, _u32p]
        L.orc_offtarget_hist_add.argtypes = [_u32p, ctypes.c_int64, _u32p]
        L.orc_offtarget_enum.argtypes = [_u32p, ctypes.c_int64, _u32p, _u32p]
        L._ot_bound = True
    return L


def seed_codes(s, pos, minus, l=20):
    """24-bit seed code (or NOT_A_SITE) of every hit, from its literal `sequence` string."""
    a = _as_u8(s)
    pos = np.ascontiguousarray(pos, dtype=np.uint32)
    out = np.empty(pos.size, dtype=np.uint32)
    _ot_bind().orc_seed_codes(_p(a, _u8p), a.size, int(l), _p(pos, _u32p), pos.size, int(bool(minus)), _p(out, _u32p))
    return out


def seed_code_of_sequence(seq):
    """The same code from a `sequence` string (e.g. a CSV cell of the reference): pure Python."""
    t = seq.replace("U", "T").upper()
    if len(t) < 12:
        return NOT_A_SITE
    code = 0
    for k in range(12):
        b = "ATCG".find(t[k])
        if b < 0:
            return NOT_A_SITE
        code |= b << (2 * k)
    return code


def offtarget_pairs(codes):
    """(n, 4) counts by the definition: all pairs."""
    codes = np.ascontiguousarray(codes, dtype=np.uint32)
    out = np.empty((codes.size, 4), dtype=np.uint32)
    _ot_bind().orc_offtarget_pairs(_p(codes, _u32p), codes.size, _p(out, _u32p))
    return out


def offtarget_hist(code_arrays, hist=None):
    """Histogram of the sites in the given code arrays (added to `hist` if given)."""
    if hist is None:
        hist = np.zeros(N_SEEDS, dtype=np.uint32)
    for codes in code_arrays:
        codes = np.ascontiguousarray(codes, dtype=np.uint32)
        _ot_bind().orc_offtarget_hist_add(_p(codes, _u32p), codes.size, _p(hist, _u32p))
    return hist


def offtarget_enum(codes, hist):
    """(n, 4) counts by histogram + neighbour enumeration (second, independent method)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint32)
    out = np.empty((codes.size, 4), dtype=np.uint32)
    _ot_bind().orc_offtarget_enum(_p(codes, _u32p), codes.size, _p(hist, _u32p), _p(out, _u32p))
    return out


def offtarget_genome(contigs, l=20):
    """Genome-wide scan over contig strings: per contig dict(ot_plus, ot_minus, seed_plus, seed_minus)."""
    per, hist = [], np.zeros(N_SEEDS, dtype=np.uint32)
    for c in contigs:
        plus, minus = scan(c, l)
        sp, sm = seed_codes(c, plus, False, l), seed_codes(c, minus, True, l)
        offtarget_hist([sp, sm], hist)
        per.append((sp, sm))
    return [dict(seed_plus=sp, seed_minus=sm, ot_plus=offtarget_enum(sp, hist), ot_minus=offtarget_enum(sm, hist))
            for sp, sm in per]
