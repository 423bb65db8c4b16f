"""fasta.table_from_bytes (native loader crp_fasta_table) and its pure-Python
specification table_from_bytes_python must equal the literal restatement
fasta.contig_table on every input, plain or not.  CPU only: the loader is host code."""
import gzip
import os

import numpy as np
import pytest

from conftest import GOLDEN, PROBES
from cropsr_amd import fasta


def _want(text):
    return [(k, v.encode("ascii", "replace")) for k, v in fasta.contig_table(text).items()]


class _Both:
    """table_from_bytes through both implementations; values as bytes."""

    @staticmethod
    def table_from_bytes(data):
        py = fasta.table_from_bytes_python(data)
        for threads in (1, 3):
            nat = [(k, bytes(v)) for k, v in fasta.table_from_bytes(data, n_threads=threads)]
            assert nat == py, (threads, data[:80])
        return py


@pytest.mark.parametrize("name", PROBES)
def test_fast_loader_on_probes(name):
    data = open(os.path.join(GOLDEN, "probe_%s.fa" % name), "rb").read()
    assert _Both.table_from_bytes(data) == _want(data.decode())


def test_fast_loader_on_sample():
    data = gzip.open(os.path.join(GOLDEN, "sample_genome.fa.gz"), "rb").read()
    got = _Both.table_from_bytes(data)
    assert got == _want(data.decode())
    assert got[0][0] == "[('Chr01'," and got[0][1].startswith(b"'ccacac") and got[0][1].endswith(b"')]")


CASES = [
    b"", b">", b">a", b">a\n", b">a\nAC", b">a\nAC\n", b">a\nAC\n>b\nGG", b">a\nAC\n>b\nGG\n", b">a\nA\nC\n>b\n",
    b">a b\nAC\n", b">a\nA C\n", b">it's\nAC\n", b">a\\b\nAC\n", b">a\nAC\n>lonely", b">lonely", b"no header\nAC\n",
    b">d\nAA\n>x\nCC\n>d\nGG\n", b">a\r\nAC\r\n>b\r\nGG\r\n", b">a\tq\nAC\n", b">a\n\nAC\n\n", b">a\n>b\nAC",
    b">a\nAC\n>b\n", b"\n>a\nAC\n", b">>a\nAC\n", b">a\nAC>b\nGG\n", b">\xc3\xa9\nAC\n",
]


@pytest.mark.parametrize("k", range(len(CASES)))
def test_fast_loader_odd_inputs(k):
    data = CASES[k]
    assert _Both.table_from_bytes(data) == _want(data.decode("utf-8", "surrogateescape"))


def test_fast_loader_random():
    rng = np.random.default_rng(12)
    alpha = np.frombuffer(b"ACGTacgtN>\n\n\n '\\x", dtype=np.uint8)
    for _ in range(400):
        data = rng.choice(alpha, int(rng.integers(0, 120))).tobytes()
        assert _Both.table_from_bytes(data) == _want(data.decode()), data
    for _ in range(50):  # well-formed multi-line FASTA
        recs = []
        for r in range(int(rng.integers(1, 6))):
            seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(0, 300))).tobytes().decode()
            recs.append(">c%d\n%s\n" % (r, "\n".join(seq[i:i + 60] for i in range(0, len(seq), 60))))
        data = "".join(recs).encode()
        if rng.random() < 0.3:
            data = data.rstrip(b"\n")
        assert _Both.table_from_bytes(data) == _want(data.decode()), data


def test_native_loader_across_pieces():
    """A 14 MB multi-record FASTA (the native loader works in 4 MiB pieces of the input): records
    and lines straddle piece borders; different thread counts give the same table."""
    rng = np.random.default_rng(21)
    parts = []
    for r, (n, width) in enumerate([(5_000_000, 60), (37, 80), (0, 60), (6_300_011, 70), (2_500_000, 61)]):
        seq = rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), n)
        full = (n // width) * width
        body = np.empty((n // width, width + 1), dtype=np.uint8)
        body[:, :width] = seq[:full].reshape(-1, width)
        body[:, width] = 10
        parts.append(b">chr%d\n" % r + body.tobytes() + seq[full:].tobytes() + (b"\n" if n > full else b""))
    data = b"".join(parts)
    want = fasta.table_from_bytes_python(data)
    assert [k for k, _ in want] == ["[('chr0',", "('chr1',", "('chr2',", "('chr3',", "('chr4',"]
    for threads in (1, 2, 8):
        got = fasta.table_from_bytes(data, n_threads=threads)
        assert [(k, bytes(v)) for k, v in got] == want
    # one bad character deep inside a body sends the whole file down the literal path, same table
    bad = bytearray(data)
    bad[9_000_000] = ord(" ") if bad[9_000_000] != 10 else ord("A")
    bad = bytes(bad)
    assert [(k, bytes(v)) for k, v in fasta.table_from_bytes(bad, n_threads=4)] == fasta.table_from_bytes_python(bad)
