"""crp_format_rows (native CSV row formatter) against Python's csv
module fed with the reference-style row tuples (rows.ContigRows, itself pinned by
the golden CSVs).  CPU only: the formatter is host code."""
import csv
import os
import io
import struct

import numpy as np
import pytest

from cropsr_amd import rows


def _python_csv(block, ids):
    buf = io.StringIO(newline="")
    csv.writer(buf).writerows([block.row(k, ids[k]) for k in range(block.n)])
    return buf.getvalue().encode("utf-8")


def _native_csv(table, ids_u8, threads=1):
    ds = rows.NativeDataset(n_threads=threads)
    ds.append(table)
    # identity id order: chunk_bytes uses ids[index_range - index - 1]; feed it reversed
    rev = ids_u8[::-1].copy()
    return ds.chunk_bytes(0, table.n, rev, table.n, lambda seqs, order: pytest.fail("no tail rows expected"))


def _ids(n, rng):
    alpha = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789", dtype=np.uint8)
    u8 = rng.choice(alpha, (n, 7))
    return u8, ["".join(map(chr, r)) for r in u8.tolist()]


@pytest.mark.parametrize("alphabet", [b"ACGT", b"ACGTacgtNUZ", b"ACGT,\"'\r\n)]\\ "])
@pytest.mark.parametrize("l", [20, 21, 5])
def test_native_rows_equal_python_csv(oracle, alphabet, l):
    rng = np.random.default_rng(len(alphabet) * 100 + l)
    for n in (0, 40, 400, 5000):
        s = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), n).tobytes().decode("latin-1")
        for name in ("[('c1',", ">plain", '("we"ird,'):
            hits = oracle.scan_score(s.encode("latin-1"), l)
            block = rows.ContigRows(name, s, hits, l)
            table = rows.ContigTable(name, s, hits, l)
            if block.n % 4 in (2, 3) or block.n == 1:  # keep clear of the tail-rescoring rule here
                continue
            ids_u8, ids = _ids(block.n, rng)
            assert _native_csv(table, ids_u8) == _python_csv(block, ids)


def test_float_repr_matches_python():
    """put_repr == repr(float) over the whole double range, not just (0, 1)."""
    rng = np.random.default_rng(3)
    vals = [0.0, -0.0, 1.0, -1.0, 0.1, 1e-4, 9.999e-5, 1e-5, 1e15, 1e16, 9.999999999999999e15, 1e17, 1e22, 1e23,
            5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 0.3, 2 / 3, 1 / 3, 123456789.125, 3e-8,
            0.039739367071031644, 1.0000000000000002, 4.35e-5, 100.0, 1e100, 1.5e-100]
    vals += [struct.unpack("<d", struct.pack("<Q", int(b)))[0]
             for b in rng.integers(1, 0x7FEFFFFFFFFFFFFF, 20000, dtype=np.int64)]
    vals += list(np.exp(rng.uniform(-40, 40, 5000)))
    vals += list(1 / (1 + np.exp(rng.uniform(-10, 18, 5000))))
    vals = [v for v in vals if v == v]
    s = "A" * 25 + "GG" + "A" * 40  # one '+' hit at i = 24 ... make windows complete: build a contig with a known hit
    s = "ACGTACGTACGTACGTACGTACGTAAGGACGTACGT"
    # one row per value: reuse a single real hit position for every row
    hits_pos = 25  # (?=.GG) at index 25: s[26:28] == 'GG'
    assert s[26:28] == "GG"
    n = len(vals)
    table = rows.ContigTable(">x", s, dict(pos_plus=np.full(n, hits_pos, np.uint32), pos_minus=np.empty(0, np.uint32),
                                           score_plus=np.array(vals), score_minus=np.empty(0)), 20)
    ids_u8 = np.full((n, 7), ord("A"), dtype=np.uint8)
    out = _native_csv(table, ids_u8, threads=4).decode().split("\r\n")[:-1]
    assert len(out) == n
    for line, v in zip(out, vals):
        assert line.split(",")[9] == repr(float(v)), (line, v)


def test_threads_do_not_change_bytes(oracle):
    rng = np.random.default_rng(6)
    s = "'" + rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), 400000).tobytes().decode() + "'),"
    hits = oracle.scan_score(s.encode(), 20)
    table = rows.ContigTable("[('c',", s, hits, 20)
    if table.n % 4 in (2, 3):
        table.pos, table.minus, table.score, table.n = table.pos[:-2], table.minus[:-2], table.score[:-2], table.n - 2
    ids_u8, _ = _ids(table.n, rng)
    one = _native_csv(table, ids_u8, threads=1)
    assert one == _native_csv(table, ids_u8, threads=7)
    assert one.count(b"\r\n") == table.n


@pytest.mark.parametrize("writer", ["native", "python"])
def test_cli_writers_agree_on_golden(writer, oracle, manifest, tmp_path, monkeypatch):
    from conftest import golden_fasta_path, oracle_scan_provider, read_golden_csv, run_cli
    for name in ("tiny", "rightend", "mixed"):
        got, _ = run_cli(tmp_path, monkeypatch, golden_fasta_path(name, tmp_path), oracle_scan_provider(oracle),
                         manifest["seed"], extra=("--csv-writer", writer))
        assert got == read_golden_csv(name), (writer, name)


def test_id_draws_equal_reference_choice():
    """write_pass_native draws ids with randint + LUT; that must be the reference's
    np.random.choice(alphanum, [size, 7]) (CROPSR.py:316-318) draw for draw."""
    alphanum = np.array(list("ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"), dtype="|U1")
    lut = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789", dtype=np.uint8)
    for size in (0, 1, 2, 37, 10000):
        np.random.seed(size + 5)
        a = np.random.choice(alphanum, [size, 7])
        after_a = np.random.random()
        np.random.seed(size + 5)
        b = lut[np.random.randint(0, 36, size=[size, 7])]
        after_b = np.random.random()
        assert (rows.ids_as_bytes(a) == b).all() and after_a == after_b


def test_each_contig_once_option(oracle, manifest, tmp_path, monkeypatch):
    """--each-contig-once (opt-in, not reference behaviour): the rows of every contig, once,
    in contig order == the LAST pass of the reference's own output (which re-emits contigs
    1..K), up to the random id and the batch-position-dependent last bits of the score."""
    from conftest import golden_fasta_path, oracle_scan_provider, read_golden_csv, run_cli
    got, _ = run_cli(tmp_path, monkeypatch, golden_fasta_path("multi", tmp_path), oracle_scan_provider(oracle),
                     manifest["seed"], extra=("--each-contig-once",))
    ours = got.decode().split("\r\n")[1:-1]
    ref = read_golden_csv("multi").decode().split("\r\n")[1:-1]
    assert len(ref) == 169 and len(ours) == 76  # 33 + 60 + 76 rows in the reference, 33 + 27 + 16 here
    last_pass = ref[-len(ours):]
    for a, b in zip(ours, last_pass):
        fa, fb = a.split(",", 1)[1].rsplit(",", 3), b.split(",", 1)[1].rsplit(",", 3)
        assert fa[0] == fb[0] and fa[2:] == fb[2:]
        ulp = abs(int(np.float64(fa[1]).view(np.int64)) - int(np.float64(fb[1]).view(np.int64)))
        assert ulp <= 4


def _table_for(oracle, n, seed, alphabet=b"ACGTacgtN,\""):
    rng = np.random.default_rng(seed)
    s = "'" + rng.choice(np.frombuffer(alphabet, dtype=np.uint8), n).tobytes().decode() + "'),"
    table = rows.ContigTable("[('c,\"x',", s, oracle.scan_score(s.encode(), 20), 20)
    ids_u8, _ = _ids(table.n, rng)
    return table, ids_u8


def _format_direct(table, ids_u8, threads):
    import ctypes
    from cropsr_amd import _native as nat
    cap = table.n * 400 + 64
    buf = np.empty(cap, dtype=np.uint8)
    used = ctypes.c_uint64()
    st = nat.lib().crp_format_rows(
        table.text.ctypes.data_as(nat.u8p), table.text.size, ctypes.cast(ctypes.c_char_p(table.chrom), nat.u8p),
        len(table.chrom), 20, table.pos.ctypes.data_as(nat.u32p), table.minus.ctypes.data_as(nat.u8p),
        table.score.ctypes.data_as(nat.f64p), ids_u8.ctypes.data_as(nat.u8p), table.n,
        buf.ctypes.data_as(nat.u8p), cap, ctypes.byref(used), threads)
    assert st == 0
    return buf[:used.value].tobytes()


def _write_direct(fd, table, ids_u8, threads):
    import ctypes
    from cropsr_amd import _native as nat
    written = ctypes.c_uint64(12345)
    st = nat.lib().crp_write_rows(
        fd, table.text.ctypes.data_as(nat.u8p), table.text.size, ctypes.cast(ctypes.c_char_p(table.chrom), nat.u8p),
        len(table.chrom), 20, table.pos.ctypes.data_as(nat.u32p), table.minus.ctypes.data_as(nat.u8p),
        table.score.ctypes.data_as(nat.f64p), ids_u8.ctypes.data_as(nat.u8p), table.n, ctypes.byref(written), threads)
    return st, written.value


@pytest.mark.parametrize("n,threads", [(0, 4), (300, 4), (200000, 1), (1500000, 5)])
def test_write_rows_to_fd_equals_format_rows(oracle, tmp_path, n, threads):
    """crp_write_rows appends exactly the bytes crp_format_rows returns (blocks of 16384 rows
    committed in order by several workers), after whatever the file already holds."""
    table, ids_u8 = _table_for(oracle, n, seed=n + threads)
    want = _format_direct(table, ids_u8, 3)
    assert want.count(b"\r\n") >= table.n
    path = tmp_path / "rows.csv"
    path.write_bytes(b"header\r\n")
    with open(path, "ab") as f:
        st, written = _write_direct(f.fileno(), table, ids_u8, threads)
    assert st == 0 and written == len(want)
    assert path.read_bytes() == b"header\r\n" + want


def test_write_rows_reports_io_errors(oracle, tmp_path):
    import ctypes
    from cropsr_amd import _native as nat
    table, ids_u8 = _table_for(oracle, 5000, seed=1)
    path = tmp_path / "ro.csv"
    path.write_bytes(b"")
    with open(path, "rb") as f:  # not open for writing: write(2) fails with EBADF
        st, written = _write_direct(f.fileno(), table, ids_u8, 2)
    assert st == nat.CRP_ERR_IO and written == 0 and ctypes.get_errno() == 9
    assert _write_direct(-1, table, ids_u8, 2)[0] == -1  # CRP_ERR_INVALID
    ds = rows.NativeDataset(n_threads=2)
    ds.append(table)
    with open(path, "rb") as f, pytest.raises(OSError):
        ds.chunk_to_fd(f.fileno(), 0, table.n, ids_u8, table.n, lambda seqs, order: np.zeros(len(seqs)))


def test_ids_drawn_in_pieces_and_ahead():
    """draw_ids (pieces of rows) and IdStream (worker thread, several passes) consume the global
    legacy stream exactly as one np.random.randint(0, 36, [size, 7]) per pass does."""
    lut = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789", dtype=np.uint8)
    sizes = [0, 5, 2500, 1, 777]
    np.random.seed(11)
    want = [lut[np.random.randint(0, 36, size=[n, 7])] for n in sizes]
    after = np.random.random()
    np.random.seed(11)
    got = [rows.draw_ids(n, piece=1000) for n in sizes]
    assert after == np.random.random() and all((a == b).all() for a, b in zip(want, got))
    np.random.seed(11)
    stream = rows.IdStream(sizes)
    got = [stream.next(n) for n in sizes]
    stream.close()
    assert after == np.random.random() and all((a == b).all() for a, b in zip(want, got))
    np.random.seed(11)
    rev = [rows.draw_ids(n, piece=1000, reverse=True) for n in sizes]
    assert after == np.random.random() and all((a[::-1] == b).all() for a, b in zip(want, rev))
    stream = rows.IdStream([3, 4])
    with pytest.raises(RuntimeError):
        stream.next(5)
    stream.close()


def test_native_legacy_id_draws_equal_numpy():
    """crp_legacy_ids continues numpy's global MT19937 stream exactly like
    np.random.randint(0, 36, [size, 7]) does: same characters, same state afterwards -- from a fresh
    seed, from the middle of a 624-word block, across block refills, after gaussian draws."""
    lut = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789", dtype=np.uint8)
    for seed, warm, size in ((0, 0, 1), (1, 0, 100), (2, 3, 89), (3, 623, 200), (4, 1000, 5000), (20261003, 17, 70000)):
        np.random.seed(seed)
        np.random.random(warm)
        if seed == 4:
            np.random.normal(size=3)  # leaves a cached gaussian in the legacy state
        before = np.random.get_state()
        want = lut[np.random.randint(0, 36, size=[size, 7])]
        after = np.random.get_state()
        for reverse in (False, True):
            np.random.set_state(before)
            got = rows.draw_ids(size, reverse=reverse)
            now = np.random.get_state()
            assert (got == (want[::-1] if reverse else want)).all(), (seed, reverse)
            assert now[0] == after[0] and (now[1] == after[1]).all() and now[2:] == after[2:]
        assert np.random.randint(0, 1 << 30) == (np.random.set_state(after), np.random.randint(0, 1 << 30))[1]


def test_write_segments_equals_the_segments_one_after_the_other(oracle, tmp_path):
    """crp_write_segments: several segments in one call -- short ones that share a 16384-row block, an empty one, one that
    spans several blocks, one with the four off-target columns and one with a features column -- == the bytes of one
    crp_write_rows_ex call per segment, whatever the thread count."""
    import ctypes
    from cropsr_amd import _native as nat
    rng = np.random.default_rng(5)
    tables = [_table_for(oracle, n, seed=100 + k) for k, n in enumerate([900, 0, 40, 230000, 7, 30000, 5000, 2500])]
    blob = np.frombuffer(b"geneA,\"x\" CDS", dtype=np.uint8)
    off = np.array([0, 5, 9, 13], dtype=np.uint64)
    extras = []
    for k, (t, _) in enumerate(tables):
        feat = rng.integers(0, 4, t.n).astype(np.uint32) if k == 5 else None
        if feat is not None:
            feat[feat == 3] = 0xFFFFFFFF
        ot = rng.integers(0, 50, (t.n, 4)).astype(np.uint32) if k == 6 else None
        extras.append((feat, ot))

    def one_by_one(path):
        with open(path, "ab") as f:
            for (t, ids_u8), (feat, ot) in zip(tables, extras):
                written = ctypes.c_uint64()
                st = nat.lib().crp_write_rows_ex(
                    f.fileno(), t.text.ctypes.data_as(nat.u8p), t.text.size, ctypes.cast(ctypes.c_char_p(t.chrom), nat.u8p), len(t.chrom), 20,
                    t.pos.ctypes.data_as(nat.u32p), t.minus.ctypes.data_as(nat.u8p), t.score.ctypes.data_as(nat.f64p),
                    ids_u8.ctypes.data_as(nat.u8p), t.n, None if feat is None else blob.ctypes.data_as(nat.u8p),
                    None if feat is None else off.ctypes.data_as(nat.u64p), None if feat is None else feat.ctypes.data_as(nat.u32p),
                    None if ot is None else ot.ctypes.data_as(nat.u32p), ctypes.byref(written), 3)
                assert st == 0

    def together(path, threads):
        segs = []
        for (t, ids_u8), (feat, ot) in zip(tables, extras):
            g = nat.RowSegment()
            g.contig_text, g.contig_len, g.chrom, g.chrom_len = t.text.ctypes.data, t.text.size, t.chrom_u8.ctypes.data, len(t.chrom)
            g.pos, g.minus, g.score, g.ids, g.n_rows = t.pos.ctypes.data, t.minus.ctypes.data, t.score.ctypes.data, ids_u8.ctypes.data, t.n
            if feat is not None:
                g.feat_blob, g.feat_off, g.feat_idx = blob.ctypes.data, off.ctypes.data, feat.ctypes.data
            if ot is not None:
                g.offtarget = ot.ctypes.data
            segs.append(g)
        with open(path, "ab") as f:
            return rows.write_segments(f.fileno(), segs, 20, threads)

    a = tmp_path / "a.csv"
    a.write_bytes(b"h\r\n")
    one_by_one(a)
    want = a.read_bytes()
    assert want.count(b"\r\n") >= sum(t.n for t, _ in tables)
    for threads in (1, 4, 9):
        b = tmp_path / ("b%d.csv" % threads)
        b.write_bytes(b"h\r\n")
        assert together(b, threads) == len(want) - 3
        assert b.read_bytes() == want
    with open(a, "rb") as f, pytest.raises(OSError):  # not open for writing
        together_fd = f.fileno()
        g = nat.RowSegment()
        t, ids_u8 = tables[0]
        g.contig_text, g.contig_len, g.chrom, g.chrom_len = t.text.ctypes.data, t.text.size, t.chrom_u8.ctypes.data, len(t.chrom)
        g.pos, g.minus, g.score, g.ids, g.n_rows = t.pos.ctypes.data, t.minus.ctypes.data, t.score.ctypes.data, ids_u8.ctypes.data, t.n
        rows.write_segments(together_fd, [g], 20, 2)
    assert rows.write_segments(-1, [], 20, 2) == 0  # nothing to write: no call


def test_passes_written_together_equal_passes_written_one_by_one(oracle, tmp_path):
    """rows.write_passes_native (what the CLI does with a run of short contigs under --each-contig-once) == one
    write_pass_native per pass: same ids from the same RNG stream, same chunk walk, same re-scored tail rows, same bytes."""
    tables = [_table_for(oracle, n, seed=300 + k)[0] for k, n in enumerate([1200, 50, 0, 64000, 333, 9000])]
    calls = []

    def rescore(seqs, order):  # (stands in for seam 2: a value that depends on the rows and the order asked for)
        calls.append((len(seqs), order))
        return seqs[:, :4].sum(axis=1) / 1000.0 + order

    def datasets():
        out = []
        for t in tables:
            ds = rows.NativeDataset(n_threads=3)
            ds.append(t)
            out.append(ds)
        return out

    a, b = tmp_path / "a.csv", tmp_path / "b.csv"
    np.random.seed(99)
    for ds in datasets():
        rows.write_pass_native(str(a), ds, rescore)
    after_a, calls_a = np.random.random(), list(calls)
    del calls[:]
    np.random.seed(99)
    sizes = [t.n for t in tables]
    ids = rows.IdStream(sizes, reverse=True)
    batch = [(ds, ids.next(len(ds))) for ds in datasets()]
    rows.write_passes_native(str(b), batch, rescore)
    ids.close()
    assert np.random.random() == after_a and calls == calls_a and len(calls) >= 3
    assert a.read_bytes() == b.read_bytes() and a.read_bytes().count(b"\r\n") >= sum(sizes)


@pytest.mark.parametrize("name", ["multi", "mixed"])
def test_each_contig_once_batched_passes_equal_single_passes(name, oracle, manifest, tmp_path, monkeypatch):
    """--each-contig-once: the CLI hands consecutive passes to the formatter together (cli.BATCH_ROWS); the CSV, the
    stdout text and the number of lines in time.txt are those of one formatter call per pass (CROPSR_BATCH_PASSES=0) --
    with the default limit (everything in one batch here) and with a limit of 20 rows (several batches, the last one partial)."""
    from conftest import golden_fasta_path, oracle_scan_provider, run_cli
    fasta = golden_fasta_path(name, tmp_path)
    got = {}
    for label, env in (("single", {"CROPSR_BATCH_PASSES": "0"}), ("batched", {}), ("small batches", {"CROPSR_BATCH_ROWS": "20"})):
        for k in ("CROPSR_BATCH_PASSES", "CROPSR_BATCH_ROWS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        d = tmp_path / label.replace(" ", "_")
        d.mkdir()
        csv_bytes, text = run_cli(d, monkeypatch, fasta, oracle_scan_provider(oracle), manifest["seed"], extra=("--each-contig-once",))
        got[label] = (csv_bytes, text, (d / "time.txt").read_text().count("Total runtime"))
    assert got["single"][0].count(b"\r\n") > 20
    assert got["batched"] == got["single"] and got["small batches"] == got["single"]


def test_id_draws_vector_and_portable_paths_agree():
    """crp_legacy_ids picks AVX-512 code at run time where the CPU has it (VBMI2: the GPU boxes' EPYCs, this container's
    Xeon); CRP_IDS_SCALAR=1 forces the portable loop.  Both continue numpy's stream identically: same characters for
    forward and last-first order, same MT19937 state afterwards -- sizes around the 64-output step, the 624-word block and
    the 4096-row staging buffer."""
    import hashlib
    import subprocess
    import sys
    code = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from cropsr_amd import rows
h = hashlib.sha256()
for seed, warm, size in ((5, 0, 1), (6, 7, 9), (7, 100, 10), (8, 0, 89), (9, 623, 4095), (10, 1, 4096), (11, 2, 4097), (12, 0, 70001), (13, 55, 300000)):
    for reverse in (False, True):
        np.random.seed(seed)
        np.random.random(warm)
        h.update(rows.draw_ids(size, reverse=reverse).tobytes())
        h.update(np.random.get_state()[1].tobytes())
        h.update(str(np.random.get_state()[2]).encode())
print(h.hexdigest())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for label, value in (("auto", ""), ("portable", "1")):
        env = dict(os.environ, CRP_IDS_SCALAR=value)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        out[label] = p.stdout.strip()
    lut = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789", dtype=np.uint8)
    h = hashlib.sha256()
    for seed, warm, size in ((5, 0, 1), (6, 7, 9), (7, 100, 10), (8, 0, 89), (9, 623, 4095), (10, 1, 4096), (11, 2, 4097), (12, 0, 70001), (13, 55, 300000)):
        for reverse in (False, True):
            np.random.seed(seed)
            np.random.random(warm)
            want = lut[np.random.randint(0, 36, size=[size, 7])]
            h.update((want[::-1] if reverse else want).tobytes())
            h.update(np.random.get_state()[1].tobytes())
            h.update(str(np.random.get_state()[2]).encode())
    assert out["auto"] == out["portable"] == h.hexdigest()
