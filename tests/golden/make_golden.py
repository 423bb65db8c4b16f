#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the real reference.

Runs only in the development container (needs /root/reference).  Nothing here is
imported by the product or by the tests; the tests read the files this script
writes.  The reference is executed unmodified in a child process:

  * ``sys.argv`` is set before ``import CROPSR`` (argparse runs at import,
    reference CROPSR.py:51),
  * ``time.sleep`` is stubbed (CROPSR.py:478 sleeps 5 s per contig),
  * the global numpy RNG is seeded so ``crispr_id`` (CROPSR.py:316-318,448) is
    reproducible,
  * CWD is a scratch directory because the reference writes ``time.txt`` there
    (CROPSR.py:371).

Two pinned environments:

  libm    OPENBLAS_NUM_THREADS=1 + AVX-512 dispatch of numpy disabled, so that
          np.exp is glibc's exp().  This is what the reference computes on any
          x86-64 host without AVX-512 and is the environment the oracle pins.
  avx512  OPENBLAS_NUM_THREADS=1, numpy's default SIMD dispatch on this host.
          Scores differ from `libm` by <= 2 ulp on a few percent of rows.

Usage:  python tests/golden/make_golden.py [--big]
"""
import argparse
import gzip
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SEED = 20261003

AVX512_OFF = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"

CHILD = r"""
import sys, time
fa, gff, out, seed = sys.argv[1:5]
sys.argv = ['CROPSR.py', '-f', fa, '-g', gff, '-o', out, '--cas9'] + sys.argv[5:]
sys.path.insert(0, %r)
time.sleep = lambda s: None
import numpy as np
import CROPSR
np.random.seed(int(seed))
CROPSR.main()
""" % REF

SCORE_CHILD = r"""
import sys
inp, out = sys.argv[1:3]
sys.argv = ['CROPSR.py', '-f', 'x', '--cas9']
sys.path.insert(0, %r)
import numpy as np
import CROPSR
seqs = np.load(inp)
np.save(out, CROPSR.rs1_score(seqs))
""" % REF


def env_for(kind):
    env = dict(os.environ)
    env["OPENBLAS_NUM_THREADS"] = "1"
    if kind == "libm":
        env["NPY_DISABLE_CPU_FEATURES"] = AVX512_OFF
    else:
        env.pop("NPY_DISABLE_CPU_FEATURES", None)
    return env


def run_reference(fa_text, gff_text, kind, scratch, extra=()):
    """Return (csv_bytes, stdout_text); `extra` = more reference command-line arguments."""
    d = tempfile.mkdtemp(dir=scratch)
    fa = os.path.join(d, "in.fa")
    gff = os.path.join(d, "in.gff")
    out = os.path.join(d, "out.csv")
    with open(fa, "w", newline="") as f:
        f.write(fa_text)
    with open(gff, "w") as f:
        f.write(gff_text)
    p = subprocess.run([sys.executable, "-c", CHILD, fa, gff, out, str(SEED)] + list(extra),
                       cwd=d, env=env_for(kind), capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(p.stderr)
    with open(out, "rb") as f:
        csv_bytes = f.read()
    shutil.rmtree(d)
    return csv_bytes, p.stdout


def run_rs1(seqs, kind, scratch):
    import numpy as np
    d = tempfile.mkdtemp(dir=scratch)
    inp = os.path.join(d, "in.npy")
    out = os.path.join(d, "out.npy")
    np.save(inp, seqs)
    p = subprocess.run([sys.executable, "-c", SCORE_CHILD, inp, out],
                       cwd=d, env=env_for(kind), capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(p.stderr)
    res = np.load(out)
    shutil.rmtree(d)
    return res


MINI_GFF = "##gff-version 3\nc1\tsrc\tgene\t1\t100\t.\t+\t.\tID=g1\n"


def wrap(seq, width):
    return "\n".join(seq[i:i + width] for i in range(0, len(seq), width))


def build_probes():
    """Small synthetic FASTA inputs, one per reference quirk."""
    rnd = random.Random(1)
    c = [''.join(rnd.choice("ACGT") for _ in range(n)) for n in (300, 200, 150)]
    probes = {}
    # (a) three contigs, multi-line records, final newline -> formatted path
    probes["multi"] = "".join(">c%d\n%s\n" % (k + 1, wrap(s, 60)) for k, s in enumerate(c))
    # (b) two-line records, no final newline -> un-formatted path (dec = 0)
    probes["twoline"] = "\n".join(">c%d\n%s" % (k + 1, s) for k, s in enumerate(c))
    # (c) header with whitespace -> token pairing shifts (App. B.2)
    probes["spaces"] = ">c1 some description GGCC here\n%s\n" % wrap(c[0], 60)
    # (d) hits hard against both contig ends
    mid = ''.join(rnd.choice("ACGT") for _ in range(30))
    probes["edges"] = ">e1\nCCAAC%sTGGAGG\n" % mid
    # (e) lowercase / N / IUPAC / odd PAMs over two contigs
    r7 = random.Random(7)
    pick = lambda alpha, n: ''.join(r7.choice(alpha) for _ in range(n))
    mixcase = ''.join(ch if r7.random() < 0.5 else ch.lower() for ch in pick("ACGT", 60))
    mix = (pick("ACGT", 400) + "acgtNGG" + mixcase + "NNNNNNNNNNAGGCCN" +
           pick("ACGTNRYKMSW", 80) + "CCnGGaggccGGcc" + pick("ACGTacgtN", 300) +
           "tGG" + pick("ACGT", 40) + "CCy" + pick("ACGT", 200))
    tail = pick("ACGTacgtN", 120)
    probes["mixed"] = ">mix\n%s\n>tail\n%s\n" % (wrap(mix, 60), wrap(tail, 60))
    # (f) CC close to the right end: dropped / truncated '-' windows (App. A.2)
    body = pick("ACGT", 90)
    probes["rightend"] = ">r1\n%sCCATCCGGACCTTCCAACCGTACCAGTCCA\n>r2\nGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCC\n" % body
    # (g) many tiny contigs, some shorter than one window, one empty-ish
    tiny = []
    for k in range(12):
        n = [3, 8, 22, 29, 30, 31, 35, 44, 63, 64, 65, 129][k]
        tiny.append(">t%d\n%s\n" % (k, wrap(pick("ACGGCC", n), 50)))
    probes["tiny"] = "".join(tiny)
    # (h) duplicate contig names: later value overwrites, first position kept
    probes["dupname"] = ">d\n%s\n>x\n%s\n>d\n%s\n" % (pick("ACGT", 120), pick("ACGT", 90), pick("ACGT", 140))
    return probes


# probes re-run with a non-default guide length (-l): l > 20 scores only windows the end of the
# contig string cuts to 30 characters, l < 20 scores nothing (CROPSR.py:458,466)
LENGTH_CASES = [("mixed", 23), ("rightend", 17), ("multi", 25), ("tiny", 21), ("edges", 24),
                # what the reference accepts beyond the engine's native range 1..50 (CROPSR.py:38-40 takes any integer):
                # every row unscored; for l <= 0 the two clauses of :419 / :430 that are otherwise always true decide
                ("mixed", 0), ("multi", -3), ("tiny", -12), ("multi", 51), ("rightend", 64), ("mixed", 100),
                # and the ends of the range in which the end of the string can still cut a window to 30 (l <= 35)
                ("rightend", 35), ("rightend", 36), ("mixed", 50), ("tiny", 1)]


# probes re-run with -v: the banner, the per-contig site counts and the progress lines on stdout
VERBOSE_CASES = ["multi", "mixed"]


def length_cases(scratch, manifest):
    probes = build_probes()
    # CRLF line ends: the reference opens the FASTA in text mode (CROPSR.py:58), which turns \r\n into \n
    crlf = probes["multi"].replace("\n", "\r\n")
    csv_b, out = run_reference(crlf, MINI_GFF, "libm", scratch)
    with open(os.path.join(HERE, "probe_crlf.fa"), "w", newline="") as f:
        f.write(crlf)
    with open(os.path.join(HERE, "probe_crlf.libm.csv"), "wb") as f:
        f.write(csv_b)
    manifest["cases"]["crlf"] = {"rows": csv_b.count(b"\r\n") - 1, "md5_libm": hashlib.md5(csv_b).hexdigest(),
                                 "stdout": out, "same_csv_as": "multi"}
    assert csv_b == open(os.path.join(HERE, "probe_multi.libm.csv"), "rb").read()
    for name in VERBOSE_CASES:
        csv_b, out = run_reference(probes[name], MINI_GFF, "libm", scratch, extra=("-v",))
        assert hashlib.md5(csv_b).hexdigest() == manifest["cases"][name]["md5_libm"]  # -v does not change the CSV
        manifest["cases"]["%s.verbose" % name] = {"probe": name, "md5_libm": hashlib.md5(csv_b).hexdigest(),
                                                  "stdout": out}
    for name, l in LENGTH_CASES:
        csv_b, out = run_reference(probes[name], MINI_GFF, "libm", scratch, extra=("-l", str(l)))
        with open(os.path.join(HERE, "probe_%s.l%d.libm.csv" % (name, l)), "wb") as f:
            f.write(csv_b)
        manifest["cases"]["%s.l%d" % (name, l)] = {"rows": csv_b.count(b"\r\n") - 1, "guide_len": l, "probe": name,
                                                   "md5_libm": hashlib.md5(csv_b).hexdigest(), "stdout": out}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true",
                    help="also run the >1e6-hit probe (about 100 s, 14 GB RSS); stores checksums only")
    ap.add_argument("--lengths-only", action="store_true",
                    help="only (re)generate the non-default guide-length cases and merge them into manifest.json")
    args = ap.parse_args()
    import numpy as np

    scratch = tempfile.mkdtemp(prefix="golden_")
    if args.lengths_only:
        with open(os.path.join(HERE, "manifest.json")) as f:
            manifest = json.load(f)
        length_cases(scratch, manifest)
        with open(os.path.join(HERE, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        shutil.rmtree(scratch)
        print("wrote guide-length fixtures to", HERE)
        return
    manifest = {"seed": SEED, "numpy": np.__version__, "cases": {}}

    # ---- sample genome: the reference's own data files are copied as fixtures
    with open(os.path.join(REF, "sample_data/sample_genome.fa")) as f:
        sample_fa = f.read()
    with open(os.path.join(REF, "sample_data/sample_genome.gff")) as f:
        sample_gff = f.read()
    with gzip.GzipFile(os.path.join(HERE, "sample_genome.fa.gz"), "wb", mtime=0) as f:
        f.write(sample_fa.encode())
    with open(os.path.join(REF, "sample_data/output.csv"), "rb") as f:
        committed = f.read()
    with gzip.GzipFile(os.path.join(HERE, "sample_output_committed.csv.gz"), "wb", mtime=0) as f:
        f.write(committed)
    # the GFF is never used by the reference beyond being parsed; keep a header-only stub
    gff_head = "".join(sample_gff.splitlines(True)[:12])
    with open(os.path.join(HERE, "sample_head.gff"), "w") as f:
        f.write(gff_head)

    csv_libm, out_libm = run_reference(sample_fa, sample_gff, "libm", scratch)
    csv_avx, _ = run_reference(sample_fa, sample_gff, "avx512", scratch)
    with gzip.GzipFile(os.path.join(HERE, "sample_libm.csv.gz"), "wb", mtime=0) as f:
        f.write(csv_libm)
    # for the avx512 variant keep only the score column (everything else is identical)
    def score_col(b):
        rows = b.decode().split("\r\n")[1:-1]
        return np.array([float(r.rsplit(",", 3)[1]) for r in rows])
    sa, sl = score_col(csv_avx), score_col(csv_libm)
    np.save(os.path.join(HERE, "sample_avx512_scores.npy"), sa)
    manifest["cases"]["sample"] = {
        "rows": len(sl), "md5_libm": hashlib.md5(csv_libm).hexdigest(),
        "md5_avx512": hashlib.md5(csv_avx).hexdigest(),
        "md5_committed": hashlib.md5(committed).hexdigest(),
        "rows_score_differs_libm_vs_avx512": int((sa != sl).sum()),
        "stdout": out_libm,
    }

    # ---- synthetic probes
    probes = build_probes()
    for name, fa in probes.items():
        csv_b, out = run_reference(fa, MINI_GFF, "libm", scratch)
        with open(os.path.join(HERE, "probe_%s.fa" % name), "w") as f:
            f.write(fa)
        with open(os.path.join(HERE, "probe_%s.libm.csv" % name), "wb") as f:
            f.write(csv_b)
        manifest["cases"][name] = {"rows": csv_b.count(b"\r\n") - 1,
                                   "md5_libm": hashlib.md5(csv_b).hexdigest(), "stdout": out}

    length_cases(scratch, manifest)

    # ---- seam-2 vectors: rs1_score on raw (n,30) uint8, incl. non-ATCG bytes
    rng = np.random.default_rng(11)
    n = 4096
    seqs = rng.choice(np.frombuffer(b"ATCG", dtype=np.uint8), size=(n, 30))
    # sprinkle other bytes the host can hand over after .upper(): N, IUPAC, quote chars
    other = np.frombuffer(b"NRYKMSW')],acgtU", dtype=np.uint8)
    m = rng.random((n, 30)) < 0.03
    seqs[m] = rng.choice(other, size=int(m.sum()))
    seqs[:8] = np.frombuffer(b"A" * 30, dtype=np.uint8)  # constant rows
    seqs[8:16] = np.frombuffer(b"N" * 30, dtype=np.uint8)
    seqs = np.ascontiguousarray(seqs.astype(np.uint8))
    s_libm = run_rs1(seqs, "libm", scratch)
    s_avx = run_rs1(seqs, "avx512", scratch)
    np.savez_compressed(os.path.join(HERE, "rs1_vectors.npz"), seqs=seqs, libm=s_libm, avx512=s_avx)
    manifest["cases"]["rs1_vectors"] = {"n": n, "differs": int((s_libm != s_avx).sum())}

    # ---- seam-2 batch-position vectors: the reference's BLAS sums the last rows of a
    # batch (n mod 4 of them) and a batch of one row in other orders than the body
    bchild = SCORE_CHILD.replace("np.save(out, CROPSR.rs1_score(seqs))", """
res = {}
for n in (1, 2, 3, 5, 6, 7, 10, 11):
    m = (len(seqs) // n) * n
    res['n%d' % n] = np.concatenate([CROPSR.rs1_score(seqs[k:k + n].copy()) for k in range(0, m, n)])
np.savez(out, **res)
""")
    d = tempfile.mkdtemp(dir=scratch)
    bseqs = np.ascontiguousarray(seqs[16:16 + 840])
    np.save(os.path.join(d, "in.npy"), bseqs)
    p = subprocess.run([sys.executable, "-c", bchild, os.path.join(d, "in.npy"), os.path.join(d, "out.npz")],
                       cwd=d, env=env_for("libm"), capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(p.stderr)
    with np.load(os.path.join(d, "out.npz")) as z:
        np.savez_compressed(os.path.join(HERE, "rs1_batches.npz"), seqs=bseqs, **{k: z[k] for k in z.files})

    # ---- the weight constants themselves (CROPSR.py:161-283) as data
    wchild = ("import sys; sys.argv=['CROPSR.py','-f','x','--cas9']; sys.path.insert(0,%r); "
              "import numpy as np, CROPSR as C; "
              "np.savez(%r, first=C.first_matrix, second=C.second_matrix, "
              "consts=np.array([C.intersect, C.low_gc, C.high_gc]))") % (REF, os.path.join(HERE, "weights.npz"))
    subprocess.run([sys.executable, "-c", wchild], check=True, cwd=scratch)

    if args.big:
        r = np.random.default_rng(12345)
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, 9_000_000)].tobytes().decode()
        fa = ">chrBig\n" + wrap(seq, 80) + "\n"
        csv_b, _ = run_reference(fa, MINI_GFF, "libm", scratch)
        manifest["cases"]["big9m"] = {"rows": csv_b.count(b"\r\n") - 1,
                                      "md5_libm": hashlib.md5(csv_b).hexdigest()}
    else:
        old = os.path.join(HERE, "manifest.json")
        if os.path.exists(old):
            with open(old) as f:
                prev = json.load(f)
            if "big9m" in prev.get("cases", {}):
                manifest["cases"]["big9m"] = prev["cases"]["big9m"]

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    shutil.rmtree(scratch)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
