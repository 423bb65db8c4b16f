"""Reference-faithful CPU port of the hot path, used ONLY as the timed CPU baseline
(bench.py cpu_baseline, kind "port") and cross-checked against the C oracle.

TEST / MEASUREMENT INFRASTRUCTURE -- never imported by the product.

It keeps the reference's algorithmic structure so that its speed is the
reference's speed:
  * the scan is Python's `re` on lookahead patterns       (CROPSR.py:98-104, 415, 426)
  * every hit costs Python-level slicing and chains of str.replace (CROPSR.py:116-129, 417-434)
  * the score builds dense one-hot float64 matrices with np.repeat / np.equal /
    np.logical_and and multiplies them by the weight vectors (CROPSR.py:285-313)
  * scoring runs over chunks of at most 1 000 000 rows      (CROPSR.py:451-461)
What it leaves out is everything outside the path: ids, row tuples, csv, sleep.
"""
import re

import numpy as np

from . import oracle as _oracle

_PLUS = re.compile(r"(?=.GG)")
_MINUS = re.compile(r"(?=CC.)")
_W1, _W2, _CONST = None, None, None


def _weights():
    global _W1, _W2, _CONST
    if _W1 is None:
        _W1, _W2, _CONST = _oracle.weights()
    return _W1, _W2, _CONST


def guide_rna(x):
    """Complementary RNA, reversed (CROPSR.py:124-129)."""
    for a, b in (("A", "U"), ("C", "Z"), ("G", "C"), ("Z", "G"), ("T", "A")):
        x = x.replace(a, b)
    return x[::-1]


def reverse_complement(x):
    """CROPSR.py:116-121."""
    for a, b in (("A", "U"), ("C", "Z"), ("G", "C"), ("Z", "G"), ("T", "A"), ("U", "T")):
        x = x.replace(a, b)
    return x[::-1]


def scan(s, l=20):
    """Rows [start, end, short, long, strand] in the reference's order."""
    n = len(s)
    rows = []
    for m in _PLUS.finditer(s):
        a, b = m.start() - l, m.start()
        if a >= 5 and a + 5 <= n + 10 and b >= 5 and b <= n + 10:
            rows.append([a, b, guide_rna(s[a:b]), guide_rna(s[a - 5:b + 5]), "+"])
    for m in _MINUS.finditer(s):
        a, b = m.start() + 3, m.start() + 3 + l
        if a >= 5 and a + 5 <= n + 10 and b >= 5 and b <= n + 10:
            rows.append([b, a, guide_rna(reverse_complement(s[a:b])),
                         guide_rna(reverse_complement(s[a - 5:b + 5])), "-"])
    return rows


def dense_score(seqs):
    """rs1_score with the reference's dense temporaries (CROPSR.py:285-313)."""
    w1, w2, const = _weights()
    n = len(seqs)
    left, right = seqs[:, 0:29], seqs[:, 1:30]
    rep1 = np.repeat(seqs, 4, axis=1)
    rep_l = np.repeat(left, 16, axis=1)
    rep_r = np.repeat(right, 16, axis=1)
    hot1 = np.empty((n, 120))
    hot2 = np.empty((n, 464))
    t1 = np.empty((n, 464))
    t2 = np.empty((n, 464))
    np.equal(rep1, np.array([[65, 84, 67, 71] * 30]), out=hot1)
    first = np.matmul(hot1, w1)
    np.equal(rep_l, np.array([([65] * 4 + [84] * 4 + [67] * 4 + [71] * 4) * 29]), out=t1)
    np.equal(rep_r, np.array([[65, 84, 67, 71] * 4 * 29]), out=t2)
    np.logical_and(t1, t2, out=hot2)
    second = np.matmul(hot2, w2)
    pre = (first + second + const[0] + const[1]) * -1
    return 1 / (1 + np.exp(pre))


def scan_score(s, l=20, chunk=1000000):
    """Whole path for one contig string: (rows, scores); unscored rows get -1."""
    rows = scan(s, l)
    scores = np.full(len(rows), -1.0)
    for lo in range(0, len(rows), chunk):
        part = rows[lo:lo + chunk]
        # like the reference (CROPSR.py:458-459): a row whose long string is not 30 characters becomes an
        # np.empty(30,) of float64 -- and ONE such row makes np.array() below promote the whole batch to
        # float64, which multiplies the size of the np.repeat temporaries by eight.  That is what the
        # reference pays on almost every real contig (a '-' hit near the end of the string suffices),
        # so the timed port must pay it too (profiles/cpu_calibration.json).
        seqs = [np.frombuffer(bytes(r[3].replace("U", "T").upper(), "ascii"), "uint8") if len(r[3]) == 30
                else np.empty(30,) for r in part]
        if not seqs:
            continue
        sc = dense_score(np.array(seqs))
        ok = np.array([len(r[3]) == 30 for r in part])
        scores[lo:lo + len(part)][ok] = sc[ok]
    return rows, scores


def _cutsite(start_pos, end_pos, crispr_sys):
    """CROPSR.py:155-158 (a function call per row there too)."""
    if crispr_sys == "cas9":
        return end_pos - 3


def full_run(s, out_csv, l=20, chunk=1000000):
    """The reference's whole per-contig pass for ONE contig string -- hot path PLUS what surrounds it
    (CROPSR.py:442-474: the 7-element row lists, the id draws, the 12-tuples, csv.writer) -- used only
    by tools/calibrate_cpu_baseline.py, so that the port can be timed against the real reference's own
    whole-run timer like for like.  Returns the number of rows written."""
    import csv
    rows = [[r[0], r[1], "chr", r[2], r[3], "cas9", r[4]] for r in scan(s, l)]
    size = len(rows)
    alphanum = np.array(list("ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"), dtype="|U1")
    written = 0
    with open(out_csv, "a") as f:
        w = csv.writer(f)
        ids = np.random.choice(alphanum, [size, 7])
        # the reference's own conversion of every id row to a str (CROPSR.py:449): characters -> code points
        # -> bytes -> str, one Python-level pass per row
        import array
        ids = [array.array("B", map(ord, z)).tobytes().decode("utf-8") for z in ids.tolist()]
        for lo in range(0, size, chunk):
            part = rows[lo:lo + chunk]
            seqs = [np.frombuffer(bytes(r[4].replace("U", "T").upper(), "ascii"), "uint8") if len(r[4]) == 30
                    else np.empty(30,) for r in part]
            score = dense_score(np.array(seqs))
            out = [(ids[lo - i - 1], r[5], r[3], r[4], r[2], r[0], r[1], _cutsite(r[0], r[1], r[5]), r[6], score[i], "", "completed")
                   if len(r[4]) == 30 else
                   (ids[lo - i - 1], r[5], r[3], r[4], r[2], r[0], r[1], r[6], -1, "", "completed")
                   for i, r in enumerate(part)]
            w.writerows(out)
            written += len(out)
    return written
