"""Two independent CPU restatements of the reference must agree with each other on
random inputs: the C oracle (oracle/crp_oracle.c: byte loops, chain maps, closed-form
sums) and the reference-faithful port (oracle/faithful_port.py: `re`, str.replace
chains, dense one-hot numpy matmuls).  Both are pinned to the real reference on the
golden fixtures; this widens the input space (alphabets, lengths, guide lengths)."""
import numpy as np
import pytest

from oracle import faithful_port as fp

ALPHABETS = [b"ACGT", b"ACGTacgtN", b"ACGTUZuzNRY')],", b"GGCCA"]


def _rows_from_oracle(oracle, s, l):
    h = oracle.scan_score(s.encode("latin-1"), l)
    rows = []
    for i in h["pos_plus"].tolist():
        rows.append([i - l, i, oracle.short_sequence(s, i, False, l), oracle.long_sequence(s, i, False, l), "+"])
    for j in h["pos_minus"].tolist():
        rows.append([j + 3 + l, j + 3, oracle.short_sequence(s, j, True, l), oracle.long_sequence(s, j, True, l), "-"])
    return rows, np.concatenate([h["score_plus"], h["score_minus"]])


@pytest.mark.parametrize("l", [20, 19, 21, 8, 33])
@pytest.mark.parametrize("alpha", range(len(ALPHABETS)))
def test_oracle_equals_faithful_port(oracle, l, alpha):
    rng = np.random.default_rng(1000 * l + alpha)
    a = np.frombuffer(ALPHABETS[alpha], dtype=np.uint8)
    for n in (0, 1, 2, 24, 25, 26, 29, 30, 31, 47, 64, 65, 200, 1500):
        body = rng.choice(a, n).tobytes().decode("latin-1")
        for s in (body, "'" + body + "'),", "'" + body + "')]"):
            want_rows, want_scores = fp.scan_score(s, l)
            got_rows, got_scores = _rows_from_oracle(oracle, s, l)
            assert got_rows == want_rows, (l, alpha, n)
            assert got_scores.shape == want_scores.shape
            both = (got_scores != -1.0) | (want_scores != -1.0)
            assert ((got_scores == -1.0) == (want_scores == -1.0)).all()
            if both.any():
                # the port runs numpy's own exp and BLAS batch orders: a few ulp at most
                ulp = np.abs(got_scores[both].view(np.int64) - want_scores[both].view(np.int64))
                assert ulp.max() <= 4, (l, alpha, n, int(ulp.max()))
