#!/usr/bin/env python3
"""Differential fuzz of the host logic against the REAL reference (development container only:
needs /root/reference).  Random small FASTA files of awkward shapes go through the unmodified
reference (child process, `libm` environment of make_golden.py) and through cropsr_amd.cli with
the CPU oracle as hit provider; CSV bytes and stdout must be identical.  Failing inputs are kept
under --keep for promotion to golden probes.

usage: python tests/golden/fuzz_vs_reference.py [-n 200] [--seed 1] [--keep DIR] [--finalize-host]
"""
import argparse
import io
import os
import random
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402


def random_fasta(rnd):
    alpha = rnd.choice(["ACGT", "ACGT", "ACGTacgt", "ACGTN", "ACGTNRYKMSWacgtn", "GC", "ACGTUZ"])
    n_rec = rnd.choice([1, 1, 2, 3, 5, 9])
    width = rnd.choice([60, 70, 80, 7, 1000])
    eol = rnd.choice(["\n", "\n", "\n", "\r\n"])
    recs = []
    for r in range(n_rec):
        n = rnd.choice([0, 1, 5, 22, 29, 30, 31, 40, 64, 65, 127, 300, 1500, 4000])
        if rnd.random() < 0.15:  # PAM-rich
            seq = "".join(rnd.choice(["GG", "CC", "A", "T", "AGG", "CCT"]) for _ in range(n // 2))
        else:
            seq = "".join(rnd.choice(alpha) for _ in range(n))
        head = rnd.choice(["c%d" % r, "Chr%02d" % r, "scaffold_%d" % r, "c%d desc here" % r, "c%d\tx" % r,
                           "it's%d" % r, "a,b%d" % r, 'q"%d' % r, "dup", "c%d|x=1" % r])
        body = eol.join(seq[i:i + width] for i in range(0, len(seq), width))
        if rnd.random() < 0.1:
            body = body.replace(eol, eol + eol, 1)  # a blank line inside the record
        recs.append(">" + head + eol + body)
    text = eol.join(recs)
    if rnd.random() < 0.7:
        text += eol
    if rnd.random() < 0.05:
        text = eol + text
    return text


def ours(fa_text, extra, scratch, finalize="gpu"):
    from conftest import OracleBackend
    from cropsr_amd import cli
    from oracle import oracle as orc
    orc.lib()
    d = tempfile.mkdtemp(dir=scratch)
    fa, gff, out = os.path.join(d, "in.fa"), os.path.join(d, "in.gff"), os.path.join(d, "out.csv")
    with open(fa, "w", newline="") as f:
        f.write(fa_text)
    with open(gff, "w") as f:
        f.write(mg.MINI_GFF)
    args = cli.build_parser().parse_args(["-f", fa, "-g", gff, "-o", out, "--cas9", "--seed", str(mg.SEED)] + list(extra)
                                         + (["--score-finalize", "host"] if finalize == "host" else []))
    cwd = os.getcwd()
    os.chdir(d)
    buf = io.StringIO()
    try:
        cli.run(args, backend=OracleBackend(orc, finalize=finalize), out=buf)
        with open(out, "rb") as f:
            res = f.read(), buf.getvalue()
    except BaseException as e:  # the reference's failure modes count too
        res = ("EXC", type(e).__name__), ""
    os.chdir(cwd)
    shutil.rmtree(d)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--keep", default=None)
    ap.add_argument("--finalize-host", action="store_true",
                    help="compare `--score-finalize host` with the reference in THIS host's default numpy environment "
                         "(AVX-512 exp where the CPU has it) instead of the pinned libm one")
    a = ap.parse_args()
    env_kind, finalize = ("avx512", "host") if a.finalize_host else ("libm", "gpu")
    rnd = random.Random(a.seed)
    scratch = tempfile.mkdtemp(prefix="fuzz_")
    bad = 0
    degenerate = ["", "\n", ">", ">\n", ">x", ">x\n", "ACGT", "ACGTGGCCAGGTTCCAGGACGT\n", ">a\n>b\n", ">a\n\n"]
    for it in range(a.n):
        fa = degenerate[it] if it < len(degenerate) else random_fasta(rnd)
        extra = rnd.choice([(), (), (), ("-l", "17"), ("-l", "23"), ("-l", "30"), ("-v",), ("-v", "-l", "21"),
                            ("-l", "0"), ("-l", "-7"), ("-l", "35"), ("-l", "51"), ("-l", "77"), ("-v", "-l", "-30")])
        try:
            want = mg.run_reference(fa, mg.MINI_GFF, env_kind, scratch, extra=extra)
        except RuntimeError as e:
            want = ("EXC", str(e).strip().splitlines()[-1].split(":")[0]), ""
        got = ours(fa, extra, scratch, finalize)
        if "-v" in extra and not isinstance(want[0], tuple) and not isinstance(got[0], tuple):
            from conftest import normalize_verbose  # paths and the CPU count differ by construction
            want = (want[0], normalize_verbose(want[1]))
            got = (got[0], normalize_verbose(got[1]))
        ok = got == want if not isinstance(want[0], tuple) else (isinstance(got[0], tuple) and got[0][1] == want[0][1])
        if not ok:
            bad += 1
            print("MISMATCH #%d extra=%r fasta=%r" % (it, extra, fa[:120]))
            if isinstance(want[0], tuple) or isinstance(got[0], tuple):
                print("   reference:", want[0] if isinstance(want[0], tuple) else "ok", " ours:", got[0] if isinstance(got[0], tuple) else "ok")
            if a.keep:
                os.makedirs(a.keep, exist_ok=True)
                with open(os.path.join(a.keep, "fuzz_%d.fa" % it), "w", newline="") as f:
                    f.write(fa)
        if (it + 1) % 25 == 0:
            print("%d inputs, %d mismatches" % (it + 1, bad), flush=True)
    shutil.rmtree(scratch)
    print("done: %d inputs, %d mismatches" % (a.n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
