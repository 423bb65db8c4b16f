"""fasta.table_from_bytes (fast loader) must equal the literal
restatement fasta.contig_table on every input, plain or not."""
import gzip
import os

import numpy as np
import pytest

from conftest import GOLDEN, PROBES
from cropsr_amd import fasta


def _want(text):
    return [(k, v.encode("ascii", "replace")) for k, v in fasta.contig_table(text).items()]


@pytest.mark.parametrize("name", PROBES)
def test_fast_loader_on_probes(name):
    data = open(os.path.join(GOLDEN, "probe_%s.fa" % name), "rb").read()
    assert fasta.table_from_bytes(data) == _want(data.decode())


def test_fast_loader_on_sample():
    data = gzip.open(os.path.join(GOLDEN, "sample_genome.fa.gz"), "rb").read()
    got = fasta.table_from_bytes(data)
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
    assert fasta.table_from_bytes(data) == _want(data.decode("utf-8", "surrogateescape"))


def test_fast_loader_random():
    rng = np.random.default_rng(12)
    alpha = np.frombuffer(b"ACGTacgtN>\n\n\n '\\x", dtype=np.uint8)
    for _ in range(400):
        data = rng.choice(alpha, int(rng.integers(0, 120))).tobytes()
        assert fasta.table_from_bytes(data) == _want(data.decode()), data
    for _ in range(50):  # well-formed multi-line FASTA
        recs = []
        for r in range(int(rng.integers(1, 6))):
            seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(0, 300))).tobytes().decode()
            recs.append(">c%d\n%s\n" % (r, "\n".join(seq[i:i + 60] for i in range(0, len(seq), 60))))
        data = "".join(recs).encode()
        if rng.random() < 0.3:
            data = data.rstrip(b"\n")
        assert fasta.table_from_bytes(data) == _want(data.decode()), data
