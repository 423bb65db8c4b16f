"""Pins the CPU oracle (oracle/) and the host-side row logic against golden
vectors produced by the real reference (tests/golden/make_golden.py).  CPU only."""
import ctypes
import hashlib
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN, LENGTH_CASES, PROBES, VERBOSE_CASES, golden_fasta_path, normalize_verbose, oracle_scan_provider, read_golden_csv, run_cli


def test_weights_match_reference_constants(oracle):
    w = np.load(os.path.join(GOLDEN, "weights.npz"))
    f, s, c = oracle.weights()
    assert (f == w["first"]).all() and (s == w["second"]).all()
    assert c[0] == w["consts"][0] and c[1] == w["consts"][1]


def test_exp_restatement_equals_host_libm(oracle):
    """orc_exp restates glibc's FMA-variant exp; on a host whose libm picks that
    variant (any x86-64 with FMA) it must agree on every input."""
    flags = open("/proc/cpuinfo").read()
    if " fma " not in flags and " fma\n" not in flags:
        pytest.skip("host CPU has no FMA: libm uses a different exp variant")
    libm = ctypes.CDLL("libm.so.6")
    libm.exp.restype = ctypes.c_double
    libm.exp.argtypes = [ctypes.c_double]
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-12, 20, 400000), rng.normal(0, 1e-3, 20000),
                        np.array([0.0, -0.0, 1e-300, -1e-300, 1e-17, -1e-17, 17.3, -9.3])])
    got = oracle.exp(x)
    want = np.array([libm.exp(v) for v in x.tolist()])
    assert (got.view(np.uint64) == want.view(np.uint64)).all()


def test_rs1_vectors(oracle):
    g = np.load(os.path.join(GOLDEN, "rs1_vectors.npz"))
    pre, score = oracle.score30(g["seqs"])
    assert (score.view(np.uint64) == g["libm"].view(np.uint64)).all()
    # numpy's AVX-512 exp differs from libm by at most 2 ulp on the final score
    ulp = np.abs(score.view(np.int64) - g["avx512"].view(np.int64))
    assert ulp.max() <= 2


@pytest.mark.parametrize("name", PROBES + ["sample"])
def test_cli_rows_equal_reference_csv(name, oracle, manifest, tmp_path, monkeypatch):
    """Oracle hits + product host logic == every byte of the reference's CSV."""
    want = read_golden_csv(name)
    got, stdout = run_cli(tmp_path, monkeypatch, golden_fasta_path(name, tmp_path),
                          oracle_scan_provider(oracle), manifest["seed"])
    assert hashlib.md5(want).hexdigest() == manifest["cases"][name]["md5_libm"]
    assert got == want
    assert stdout == manifest["cases"][name]["stdout"]


def test_sample_scores_vs_committed_output(oracle, sample_fasta_text):
    """The reference's own committed sample_data/output.csv pins everything but
    the random id exactly and the score to <= 15 ulp."""
    import gzip
    with gzip.open(os.path.join(GOLDEN, "sample_output_committed.csv.gz"), "rt", newline="") as f:
        committed = f.read().split("\r\n")[1:-1]
    with gzip.open(os.path.join(GOLDEN, "sample_libm.csv.gz"), "rt", newline="") as f:
        ours = f.read().split("\r\n")[1:-1]
    assert len(committed) == len(ours) == 17314
    worst = 0
    for a, b in zip(committed, ours):
        fa, fb = a.split(",", 1)[1].rsplit(",", 3), b.split(",", 1)[1].rsplit(",", 3)
        assert fa[0] == fb[0] and fa[2:] == fb[2:]  # all columns except id and score
        ua = np.float64(fa[1]).view(np.int64)
        ub = np.float64(fb[1]).view(np.int64)
        worst = max(worst, abs(int(ua) - int(ub)))
    assert worst <= 15


def test_sample_avx512_scores_within_2ulp(oracle, sample_fasta_text):
    seq = "".join(sample_fasta_text.split("\n")[1:])
    s = ("'" + seq + "')]").encode()
    h = oracle.scan_score(s)
    ours = np.concatenate([h["score_plus"], h["score_minus"]])
    avx = np.load(os.path.join(GOLDEN, "sample_avx512_scores.npy"))
    assert ours.size == avx.size
    assert np.abs(ours.view(np.int64) - avx.view(np.int64)).max() <= 2


def test_big_chunk_csv_md5_cpu(oracle, manifest, tmp_path, monkeypatch):
    """> 1 000 000 hits on one contig (reference quirks B.4/B.5: broken final chunk,
    backwards ids): oracle hits + product host logic == md5 of a real reference run."""
    r = np.random.default_rng(12345)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, 9000000)].tobytes().decode()
    fa = tmp_path / "big.fa"
    with open(fa, "w") as f:
        f.write(">chrBig\n")
        f.write("\n".join(seq[i:i + 80] for i in range(0, len(seq), 80)))
        f.write("\n")
    got, _ = run_cli(tmp_path, monkeypatch, str(fa), oracle_scan_provider(oracle), manifest["seed"])
    assert got.count(b"\r\n") - 1 == manifest["cases"]["big9m"]["rows"]
    assert hashlib.md5(got).hexdigest() == manifest["cases"]["big9m"]["md5_libm"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_exp_table_path_needs_no_small_argument_case(oracle):
    """The HIP scorer drops glibc's early `return 1 + x` for |x| < 2^-54 (it exists to keep the
    floating-point flags clean).  The table path must then give the same VALUE there: checked
    against libm (math.exp) on zeros, denormals, powers of two down to the smallest denormal, values just
    either side of 2^-54 -- and against orc_exp on the scorer's working range."""
    tiny = [0.0, -0.0, 5e-324, -5e-324, 2.2250738585072014e-308, -2.2250738585072014e-308]
    tiny += [s * 2.0 ** -k for k in range(40, 1075, 7) for s in (1.0, -1.0)]
    tiny += [s * np.nextafter(2.0 ** -54, d) for d in (0.0, 1.0) for s in (1.0, -1.0)]
    rng = np.random.default_rng(5)
    tiny += list(rng.uniform(-1, 1, 2000) * 2.0 ** rng.integers(-1074, -50, 2000).astype(np.float64))
    tiny = np.array(tiny)
    libm = np.array([math.exp(v) for v in tiny.tolist()])  # math.exp IS libm's exp (numpy may use its own SIMD exp)
    assert (bits(oracle.exp_table_path(tiny)) == bits(libm)).all()
    x = rng.uniform(-40, 40, 200000)
    assert (bits(oracle.exp_table_path(x)) == bits(oracle.exp(x))).all()


@pytest.mark.parametrize("name,guide_len", LENGTH_CASES)
def test_cli_reproduces_reference_csv_other_guide_lengths(name, guide_len, oracle, manifest, tmp_path, monkeypatch):
    """-l 17 ... 25 -- and 1, 35 / 36, 50, and beyond the engine's range 0, -3, -12, 51, 64, 100: the oracle's
    keep-filter / window rules for l != 20, the host's literal refilter for clamped scans and the row assembly
    against the real reference's CSV bytes and stdout."""
    got, stdout = run_cli(tmp_path, monkeypatch, golden_fasta_path(name, tmp_path), oracle_scan_provider(oracle),
                          manifest["seed"], extra=("-l", str(guide_len)))
    case = manifest["cases"]["%s.l%d" % (name, guide_len)]
    assert got == read_golden_csv(name, guide_len)
    assert hashlib.md5(got).hexdigest() == case["md5_libm"] and stdout == case["stdout"]
    import csv
    import io
    body = list(csv.reader(io.StringIO(got.decode(), newline="")))[1:]
    scored = sum(1 for r in body if len(r) == 12)
    assert len(body) == case["rows"] and (scored == 0 if guide_len < 20 else scored < len(body))


@pytest.mark.parametrize("l", [-40, -12, -3, -1, 0, 51, 58, 59, 64, 100, 200])
def test_guide_lengths_beyond_the_engines_range_are_the_clamped_scan_refiltered(oracle, l):
    """The reference takes any integer -l (CROPSR.py:38-40).  The engine scans lengths 0..50; outside, cli.py scans with
    the nearer end of that range and applies the literal keep-filter (CROPSR.py:419 / :430, all four clauses) on the
    host.  That this is the reference's hit set is checked here against the oracle's LITERAL scan at the true length,
    on contigs whose ends, N runs and decorations exercise every clause; no row is scored at those lengths."""
    from cropsr_amd import cli
    rng = np.random.default_rng(abs(l) + 7)
    for n in (0, 3, 9, 40, 75, 130, 700, 5000):
        body = rng.choice(np.frombuffer(b"ACGTGGCCacgtN", dtype=np.uint8), n).tobytes()
        for s in (body, b"'" + body + b"'),", b"GG" + body + b"CC", b"CC" + body + b"GG"):
            want_plus, want_minus = oracle.scan(s, l)
            l_dev = cli.device_guide_length(l)
            assert l_dev in (0, 50)
            got = cli.refilter_hits(oracle.scan_score(s, l_dev), len(s), l)
            assert (got["pos_plus"] == want_plus).all() and (got["pos_minus"] == want_minus).all(), (l, n)
            lit = oracle.scan_score(s, l)
            assert (lit["score_plus"] == -1).all() and (lit["score_minus"] == -1).all()
            assert (got["score_plus"] == -1).all() and got["score_plus"].size == want_plus.size


@pytest.mark.parametrize("name", VERBOSE_CASES)
def test_cli_verbose_output_equals_reference(name, oracle, manifest, tmp_path, monkeypatch):
    """-v: banner, settings block, progress lines and the per-contig PAM-site counts (which the
    reference takes BEFORE its keep-filter, CROPSR.py:436-439) as the real reference prints them;
    paths and the CPU count masked on both sides.  The CSV is the non-verbose one."""
    got, stdout = run_cli(tmp_path, monkeypatch, golden_fasta_path(name, tmp_path), oracle_scan_provider(oracle),
                          manifest["seed"], extra=("-v",))
    assert got == read_golden_csv(name)
    assert normalize_verbose(stdout) == normalize_verbose(manifest["cases"][name + ".verbose"]["stdout"])
