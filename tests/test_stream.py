"""Seam 1 as a pipeline: crp_scan_stream (cropsr_amd/csrc/crp_stream.cpp; Engine.scan_stream).

The reference's loop produces and consumes contig by contig (CROPSR.py:409-474).  crp_scan_stream sends the genome through in
slices -- upload of slice k + 1, scan of slice k and table fetch of slice k - 1 side by side -- and must return, contig by
contig, exactly what the loop appends: the oracle's rows, bit for bit, whatever the slice size and wherever a contig is cut.
"""
import ctypes
import hashlib
import os
import time

import numpy as np
import pytest

from test_node import ALPHA, _check_against_oracle, _genome, bits


@pytest.mark.gpu
def test_scan_stream_vs_oracle_all_slice_sizes(oracle):
    """Genomes with contigs shorter than, around and far longer than a slice (empty contigs, 60 scaffolds, an empty genome):
    slices of 20 000, 70 000 and 300 000 characters and the default (everything in one slice), guide lengths 7 / 20 / 23, the
    pre-sigmoid column -- per contig the oracle's rows.  The same engine serves every call (lanes and arenas are reused and
    resized)."""
    from cropsr_amd import Engine
    rng = np.random.default_rng(2026)
    genomes = [
        [300_000, 5, 0, 70_000, 9_000, 123_457, 64, 1, 40_000],
        [1_500_000],
        [2_000, 3_000] + [int(v) for v in rng.integers(1, 6_000, 60)] + [400_000],
        [10, 20, 30],
        [0],
        [],
    ]
    with Engine(0) as eng:
        for g, lengths in enumerate(genomes):
            contigs = _genome(rng, lengths)
            for slice_chars in (20_000, 70_000, 300_000, 0):
                for l, pre in ((20, False), (20, True), (23, False), (7, False)):
                    hits = eng.scan_stream(contigs, l, want_pre=pre, slice_chars=slice_chars)
                    _check_against_oracle(hits, contigs, oracle, l, (g, slice_chars, l, pre), pre=pre)
                    st = hits.stream_stats
                    total = sum(len(c) for c in contigs)
                    if slice_chars and total > 3 * slice_chars:
                        assert st["slices"] >= 3 and st["lanes"] == min(4, st["slices"]), st
                    if total:
                        assert st["wall_s"] > 0 and st["first_slice_on_host_s"] <= st["wall_s"]
        # the classic calls on the same engine still work (lane 0 is the engine's own context)
        contigs = _genome(rng, [120_000, 3_000])
        arena = eng.arena(contigs)
        one = arena.scan_score(20)
        for k, c in enumerate(contigs):
            want = oracle.scan_score(c, 20)
            assert (one.contig(k)["pos_plus"] == want["pos_plus"]).all() and (bits(one.contig(k)["score_minus"]) == bits(want["score_minus"])).all()
        arena.close()


@pytest.mark.gpu
def test_scan_stream_capacity_protocol_and_pinned_tables(oracle):
    """Tables that are too small: CRP_ERR_CAPACITY, nothing written beyond the capacity, the totals to come back with (the
    Python wrapper comes back by itself; a poly-G contig has a hit at every position).  Pinned tables (Engine.empty_tables,
    crp_host_alloc): filled by DMA, the same rows; a NULL column is skipped."""
    from cropsr_amd import Engine, _native as nat
    rng = np.random.default_rng(7)
    contigs = _genome(rng, [200_000, 50_000]) + [b"'" + b"G" * 90_000 + b"')]"]
    with Engine(0) as eng:
        L = nat.lib()
        bufs = [np.frombuffer(c, dtype=np.uint8) for c in contigs]
        ptrs = (ctypes.c_void_p * 3)(*[b.ctypes.data for b in bufs])
        lens = np.array([b.size for b in bufs], dtype=np.uint64)
        cap = 1000
        guard = np.full(cap + 64, 0xDEADBEEF, dtype=np.uint32)
        sc = np.full(cap + 64, -7.0)
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        st = L.crp_scan_stream(eng._ctx, ptrs, lens.ctypes.data_as(nat.u64p), 3, 20, 0, 30_000, guard.ctypes.data_as(nat.u32p),
                               sc.ctypes.data_as(nat.f64p), cap, None, None, 0, None, ctypes.byref(a), ctypes.byref(b), None)
        assert st == nat.CRP_ERR_CAPACITY and (guard[cap:] == 0xDEADBEEF).all() and (sc[cap:] == -7.0).all()
        want = [oracle.scan_score(c, 20) for c in contigs]
        assert a.value == sum(w["pos_plus"].size for w in want) and b.value == sum(w["pos_minus"].size for w in want)
        assert "rows are needed" in L.crp_last_error(eng._ctx).decode()
        # the wrapper: the default estimate (1/6 of the characters per strand) is too small for poly-G -> exact sizes, again
        hits = eng.scan_stream(contigs, 20, slice_chars=30_000, density=0.05)
        _check_against_oracle(hits, contigs, oracle, 20, "retry")
        # pinned tables, reused for two genomes
        out = eng.empty_tables(a.value + 10, b.value + 10)
        assert nat.lib().crp_host_alloc(0, ctypes.byref(ctypes.c_void_p())) == 0
        for rep in range(2):
            hits = eng.scan_stream(contigs if rep == 0 else contigs[::-1], 20, out=out, slice_chars=50_000)
            _check_against_oracle(hits, contigs if rep == 0 else contigs[::-1], oracle, 20, ("pinned", rep))
            assert hits.pos_plus.ctypes.data == out[0].ctypes.data  # (views of the caller's arrays, no copy)
        # bad arguments
        assert L.crp_scan_stream(None, ptrs, lens.ctypes.data_as(nat.u64p), 3, 20, 0, 0, None, None, 0, None, None, 0, None, None, None, None) == -1
        assert L.crp_scan_stream(eng._ctx, ptrs, lens.ctypes.data_as(nat.u64p), 3, 20, 4, 0, None, None, 0, None, None, 0, None, None, None, None) == -1
        assert L.crp_scan_stream(eng._ctx, ptrs, lens.ctypes.data_as(nat.u64p), 3, 77, 0, 0, None, None, 0, None, None, 0, None, None, None, None) != 0


@pytest.mark.gpu
def test_scan_stream_randomised_genomes_vs_oracle(oracle):
    """Seeded fuzz: random genomes (contig lengths around halo, word, tile and slice borders; five alphabets; decoration),
    random slice sizes from a few hundred characters up, random guide lengths -- every contig's rows equal the oracle's."""
    from conftest import fuzz_settings
    from cropsr_amd import Engine
    trials, seed, tick = fuzz_settings(60, 20261007)
    rng = np.random.default_rng(seed)
    alphabets = [b"ACGT", b"ACGTacgtN", b"GGCC", b"ACGTUZuzN')],", b"GGGGGGCCCCCCAT"]
    anchors = [0, 1, 30, 64, 127, 128, 129, 255, 320, 384, 4095, 4096, 4097, 8192, 16384, 65535, 65536, 65537, 131072, 200000]
    total = cuts = 0
    with Engine(0) as eng:
        for trial in range(trials):
            tick("stream", trial)
            contigs = []
            for _ in range(int(rng.integers(0, 13))):
                n = max(0, int(anchors[rng.integers(len(anchors))] + rng.integers(-70, 71)))
                if rng.random() < 0.3:
                    n = int(rng.integers(0, 3000))
                body = rng.choice(np.frombuffer(alphabets[rng.integers(len(alphabets))], dtype=np.uint8), n).tobytes()
                deco = rng.integers(3)
                contigs.append(body if deco == 0 else b"'" + body + (b"')," if deco == 1 else b"')]"))
            l = 20 if rng.random() < 0.7 else int(rng.integers(0, 51))
            pre = bool(rng.random() < 0.25)
            slice_chars = int(rng.choice([1, 400, 1000, 5000, 30_000, 100_000, 0]))
            # both scan modes and both tile shapes behind the pipeline (the three-launch sequence runs inside the drainer's wait)
            eng.configure(two_pass=bool(rng.random() < 0.3), geometry=str(rng.choice(["auto", "large", "small"])))
            hits = eng.scan_stream(contigs, l, want_pre=pre, slice_chars=slice_chars, density=float(rng.choice([0.02, 0.2, 1.0])))
            total += _check_against_oracle(hits, contigs, oracle, l, (trial, l, pre, slice_chars), pre=pre)
            cuts += int(hits.stream_stats["slices"] > 1)
    assert total > 20000 * trials // 60 and cuts > trials // 3


@pytest.mark.gpu
@pytest.mark.slow
def test_scan_stream_switchgrass_like_equals_one_arena_scan():
    """VERDICT r05 #4's acceptance: BASELINE.json configs[4]'s stand-in (1.13 Gb, 644 contigs, 52.4 M hits) through the
    pipeline == the one-arena scan's tables by SHA-256 per contig -- with fresh pageable tables and with pinned ones -- and
    the host-to-host time of both ways, printed."""
    import bench_workload as bw
    from cropsr_amd import Engine

    def digest(h):
        d = hashlib.sha256()
        for key in ("pos_plus", "score_plus", "pos_minus", "score_minus"):
            d.update(np.ascontiguousarray(h[key]).tobytes())
        return d.hexdigest()

    wl = bw.switchgrass_like()
    strings = [wl.contig_string(k) for k in range(len(wl.specs))]
    with Engine(0) as eng:
        t0 = time.perf_counter()
        arena = eng.arena(strings)
        t1 = time.perf_counter()
        one = arena.scan_score(20)
        t2 = time.perf_counter()
        want = [digest(one.contig(k)) for k in range(len(strings))]
        n_hits = one.n_plus + one.n_minus
        n_plus, n_minus = one.n_plus, one.n_minus
        arena.close()
        del one
        eng.stream_prepare()
        walls = []
        for rep in range(3):
            hits = eng.scan_stream(strings, 20)
            walls.append(hits.stream_stats["wall_s"])
            if rep == 0:
                got = [digest(hits.contig(k)) for k in range(len(strings))]
                assert got == want, [k for k in range(len(strings)) if got[k] != want[k]][:10]
                assert hits.n_plus + hits.n_minus == n_hits
            stats = hits.stream_stats
            del hits
        out = eng.empty_tables(n_plus, n_minus)
        pinned = []
        for rep in range(3):
            hits = eng.scan_stream(strings, 20, out=out)
            pinned.append(hits.stream_stats["wall_s"])
        got = [digest(hits.contig(k)) for k in range(len(strings))]
        assert got == want
    print("switchgrass-like host to host: serial upload %.1f ms + scan and fetch %.1f ms; pipelined, fresh pageable tables %s ms; "
          "pinned tables %s ms; last run: %s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, ["%.1f" % (w * 1e3) for w in walls],
                                                 ["%.1f" % (w * 1e3) for w in pinned], stats))


@pytest.mark.gpu
@pytest.mark.slow
def test_scan_stream_maize_size_genome_equals_engine_genome():
    """A 2.4 Gb maize-size stand-in -- more than one arena addresses, chromosomes of 150-300 Mb that the default 64 Mi-character
    slices cut into pieces with halos -- through the pipeline == Engine.genome's tables (arena calls, a genome spread over
    arenas) by SHA-256 per contig, and the same into tables that are too small first (the capacity protocol at full size)."""
    import bench_workload as bw
    from cropsr_amd import Engine

    def digest(h):
        d = hashlib.sha256()
        for key in ("pos_plus", "score_plus", "pos_minus", "score_minus"):
            d.update(np.ascontiguousarray(h[key]).tobytes())
        return d.hexdigest()

    wl = bw.maize_like()
    strings = [wl.contig_string(k) for k in range(len(wl.specs))]
    total = sum(s.size for s in strings)
    assert total > (1 << 31) and max(s.size for s in strings) > (64 << 20)
    with Engine(0) as eng:
        genome = eng.genome(strings)
        one = genome.scan_score(20)
        want = [digest(one.contig(k)) for k in range(len(strings))]
        n_hits = one.n_plus + one.n_minus
        del one
        genome.close()
        hits = eng.scan_stream(strings, 20)
        stats = hits.stream_stats
        assert stats["slices"] >= total // (64 << 20)
        got = [digest(hits.contig(k)) for k in range(len(strings))]
        assert got == want, [k for k in range(len(strings)) if got[k] != want[k]][:10]
        assert hits.n_plus + hits.n_minus == n_hits
        del hits
        hits = eng.scan_stream(strings, 20, density=1 / 64)  # tables sized for a 64th of the characters: one retry with exact sizes
        got = [digest(hits.contig(k)) for k in range(len(strings))]
        assert got == want and hits.n_plus + hits.n_minus == n_hits
    print("maize-like through the pipeline: %d characters, %d slices on %d lanes, %d hits, %.1f ms host to host"
          % (total, stats["slices"], stats["lanes"], n_hits, stats["wall_s"] * 1e3))


@pytest.mark.gpu
def test_integration_md_stream_snippet_runs(oracle):
    """The ctypes patch INTEGRATION.md shows for the pipelined seam 1, executed as printed (after the seam-2 block, which opens
    `_crp` and `_ctx`): the tables it ends up with are the oracle's, contig after contig."""
    import types
    from conftest import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()

    def block(after):
        at = text.index(after)
        code = text[text.index("```python", at) + len("```python"):]
        return code[:code.index("```")]
    seam2 = block("### Seam 2").replace('"/path/to/cropsr_amd/libcropsr_hip.so"', repr(os.path.join(ROOT, "cropsr_amd", "libcropsr_hip.so")))
    stream = block("### Seam 1 as a pipeline")
    assert "crp_scan_stream(" in stream
    rng = np.random.default_rng(78)
    contigs = _genome(rng, [220_000, 3_000, 64, 0, 90_000])
    env = {"fasta_file": {"k%d" % k: c.decode("ascii") for k, c in enumerate(contigs)}, "args": types.SimpleNamespace(l=20)}
    exec(seam2, env)
    exec(stream, env)
    per = list(env["per_contig"])
    a = b = 0
    for k, c in enumerate(contigs):
        want = oracle.scan_score(c, 20)
        n_p, n_m = per[2 * k], per[2 * k + 1]
        assert (env["pos_p"][a:a + n_p] == want["pos_plus"]).all() and (bits(env["sc_p"][a:a + n_p]) == bits(want["score_plus"])).all()
        assert (env["pos_m"][b:b + n_m] == want["pos_minus"]).all() and (bits(env["sc_m"][b:b + n_m]) == bits(want["score_minus"])).all()
        a, b = a + n_p, b + n_m
    assert a == env["n_plus"].value and b == env["n_minus"].value and a + b > 20_000
    env["_crp"].crp_destroy(env["_ctx"])
