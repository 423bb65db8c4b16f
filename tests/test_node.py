"""The single-process node handle (crp_node_*): ONE process, N logical devices, the cut, the fan-out
and the gatherv inside the library.  The reference is one process with one contig loop (CROPSR.py:333, :409); what the
node returns must be, contig by contig, exactly what that loop appends -- i.e. what one GPU alone returns and what the
oracle says.

CPU part (no GPU): the cut itself (crp_plan_shares) against the Python statement of the same rule that the
process-per-GPU path uses (parallel.split_evenly), and the properties crp_node_gather relies on.
GPU part: on the one-GPU box the N devices are the same physical GPU listed N times ({0, 0, 0, 0}); RCCL refuses
duplicate devices, so the exchange runs as the library's device-to-device copies there -- the same stand-in role the
host transport plays for the process-per-GPU path -- and RCCL itself is exercised with a one-device node
(ncclCommInitAll with world 1).
"""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


# ------------------------------------------------------------------ the cut (host code)
def _random_lengths(rng):
    kind = int(rng.integers(0, 4))
    n = int(rng.integers(0, 40))
    if kind == 0:
        return [int(v) for v in rng.integers(0, 50_000, n)]
    if kind == 1:  # a few chromosomes and many scaffolds
        return [int(v) for v in rng.integers(1_000_000, 9_000_000, max(1, n // 6))] + [int(v) for v in rng.integers(1, 30_000, n)]
    if kind == 2:  # one contig
        return [int(rng.integers(0, 5_000_000))]
    return [int(v) for v in np.exp(rng.uniform(0, 16, n)).astype(np.int64)]


def test_plan_shares_equals_split_evenly():
    """crp_plan_shares (C++, what crp_node_load cuts by) == parallel.split_evenly (what sharded_scan cuts by)."""
    from cropsr_amd import node, parallel
    rng = np.random.default_rng(5)
    n_cut = 0
    for trial in range(400):
        lengths = _random_lengths(rng)
        world = int(rng.integers(1, 10))
        min_piece = int(rng.choice([0, 1, 64, 4096, 100_000]))
        got = node.plan_shares(lengths, world, min_piece)
        pieces, owner = parallel.split_evenly(lengths, world, min_piece if min_piece else 4096)
        assert got == [(k, s, e, o) for (k, s, e), o in zip(pieces, owner)], (trial, lengths, world, min_piece)
        n_cut += len(got) - len(lengths)
    assert n_cut > 300  # (the trials did cut contigs)


def test_plan_shares_properties():
    """What crp_node_gather relies on: the pieces cover every contig exactly once in order; owners never decrease; only
    the FIRST piece of a device can start inside a contig and only its LAST can end inside one (so the owned rows of a
    device's tables are one run); at most world - 1 cuts; shares equal to within min_piece."""
    from cropsr_amd import node
    rng = np.random.default_rng(6)
    for trial in range(300):
        lengths = _random_lengths(rng)
        world = int(rng.integers(1, 10))
        plan = node.plan_shares(lengths, world)
        assert [p[0] for p in plan] == sorted(p[0] for p in plan)
        assert [p[3] for p in plan] == sorted(p[3] for p in plan) and all(0 <= p[3] < world for p in plan)
        for k, n in enumerate(lengths):
            mine = [p for p in plan if p[0] == k]
            assert mine and mine[0][1] == 0 and mine[-1][2] == n
            assert all(a[2] == b[1] for a, b in zip(mine, mine[1:]))
        assert len(plan) <= len(lengths) + world - 1
        for r in range(world):
            own = [p for p in plan if p[3] == r]
            for j, (k, s, e, _) in enumerate(own):
                assert s == 0 or j == 0, (trial, r, own)
                assert e == lengths[k] or j == len(own) - 1, (trial, r, own)
        total = sum(lengths)
        if total >= world * 3 * 4096 and max(lengths) >= 4096 * 4 and len(lengths) < 4:
            shares = [sum(e - s for _, s, e, o in plan if o == r) for r in range(world)]
            assert max(shares) - min(shares) <= 2 * 4096 + world, (trial, shares)


def test_plan_shares_capacity_protocol_and_bad_arguments():
    import ctypes
    from cropsr_amd import _native as nat
    L = nat.lib()
    lens = np.array([100_000, 50_000], dtype=np.uint64)
    n = ctypes.c_uint64()
    assert L.crp_plan_shares(lens.ctypes.data_as(nat.u64p), 2, 4, 0, None, 0, ctypes.byref(n)) == nat.CRP_ERR_CAPACITY
    assert n.value == 5
    out = np.zeros((5, 4), dtype=np.uint64)
    assert L.crp_plan_shares(lens.ctypes.data_as(nat.u64p), 2, 4, 0, out.ctypes.data_as(nat.u64p), 5, ctypes.byref(n)) == 0
    assert L.crp_plan_shares(lens.ctypes.data_as(nat.u64p), 2, 0, 0, out.ctypes.data_as(nat.u64p), 5, ctypes.byref(n)) == -1
    assert L.crp_plan_shares(None, 2, 2, 0, out.ctypes.data_as(nat.u64p), 5, ctypes.byref(n)) == -1
    assert L.crp_node_size(None) == -1 and not L.crp_node_ctx(None, 0) and L.crp_node_destroy(None) == 0
    assert L.crp_node_gather(None, 0, 0) == -1 and L.crp_node_last_error(None) == b""


def test_node_has_no_cpu_fallback_without_gpu():
    """Without a HIP device the node handle fails loudly too (CRP_ERR_NO_DEVICE), and the single-process bench and CLI with it."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    from cropsr_amd import _native as nat, node as nd
    with pytest.raises(nat.CropsrHipError) as e:
        nd.Node([0, 1])
    assert e.value.status == nat.CRP_ERR_NO_DEVICE
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--scale", "0.01"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "no usable HIP device" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_pos16_host_transport_round_trip():
    """The 16-bit position packing of the exchange as the host transport (sockets) does it in numpy: exact for any
    ascending table, including empty buckets, runs longer than 65 536 positions without a hit, and the table's ends."""
    from cropsr_amd import parallel
    rng = np.random.default_rng(9)
    cases = [np.empty(0, np.uint32), np.array([0], np.uint32), np.array([65535, 65536, 65537, 1 << 20, (1 << 31) - 1], np.uint32)]
    for _ in range(30):
        n = int(rng.integers(1, 20000))
        span = int(rng.choice([1000, 70_000, 10_000_000, (1 << 31) - 1]))
        cases.append(np.unique(rng.integers(0, span, n)).astype(np.uint32))
    for pos in cases:
        lo16, bstart = parallel.pack_pos16(pos)
        assert lo16.dtype == np.uint16 and bstart.dtype == np.uint32 and lo16.size == pos.size
        back = parallel.unpack_pos16(lo16, bstart)
        assert back.dtype == np.uint32 and (back == pos).all()


# ------------------------------------------------------------------ GPU
ALPHA = np.frombuffer(b"ACGTACGTACGTGGCCacgtN", dtype=np.uint8)


def _genome(rng, lengths):
    out = []
    for k, n in enumerate(lengths):
        tail = b"')]" if k == len(lengths) - 1 else b"'),"
        out.append(b"'" + rng.choice(ALPHA, int(n)).tobytes() + tail)
    return out


def _check_against_oracle(hits, contigs, oracle, l, ctx, pre=False):
    total = 0
    for k, c in enumerate(contigs):
        want = oracle.scan_score(c, l)
        got = hits.contig(k)
        for strand in ("plus", "minus"):
            assert got["pos_" + strand].shape == want["pos_" + strand].shape, (ctx, k, strand)
            assert (got["pos_" + strand] == want["pos_" + strand]).all(), (ctx, k, strand)
            col = "pre_" if pre else "score_"
            assert (bits(got["score_" + strand]) == bits(want[col + strand])).all(), (ctx, k, strand, col)
            total += want["pos_" + strand].size
    assert hits.n_plus + hits.n_minus == total, ctx
    return total


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3, 4, 7])
def test_node_logical_devices_vs_oracle(oracle, world):
    """Small genomes (contigs shorter than, around and far longer than a share; empty contigs; many scaffolds between
    chromosomes) over `world` logical devices on GPU 0, every option of the exchange: positions packed to 16 bits or
    raw, the pre-sigmoid column, another root -- each contig's rows == the oracle's, bit for bit."""
    from cropsr_amd import node as nd
    rng = np.random.default_rng(100 + world)
    genomes = [
        [300_000, 5, 0, 70_000, 9_000, 123_457, 64, 1, 40_000],
        [1_500_000],
        [2_000, 3_000] + [int(v) for v in rng.integers(1, 6_000, 60)] + [400_000],
        [10, 20, 30],
        [],
    ]
    with nd.Node([0] * world) as node:
        assert node.size == world
        for g, lengths in enumerate(genomes):
            contigs = _genome(rng, lengths)
            node.load(contigs)
            plan = node.plan()
            assert [(p["contig"], p["start"], p["end"], p["device"]) for p in plan] == nd.plan_shares([len(c) for c in contigs], world)
            for l, kw in ((20, {}), (20, {"pos16": False}), (20, {"pre": True}), (20, {"root": world - 1}), (23, {}), (7, {"pos16": False, "root": world // 2}),
                          (20, {"to_host": True}), (20, {"to_host": True, "pre": True}), (9, {"to_host": True})):
                hits = node.scan(l, **kw)
                _check_against_oracle(hits, contigs, oracle, l, (world, g, l, kw), pre=kw.get("pre", False))
                st = node.gather_stats()
                if kw.get("to_host"):  # nothing crossed between devices: every device's rows came over its own link
                    assert st["transport"].startswith("none: every device") and st["bytes_to_root"] == 0
                    assert node.count_scored() == int((hits.score_plus != -1).sum() + (hits.score_minus != -1).sum())
                else:
                    assert st["transport"] == ("device-to-device copies" if world > 1 else "none (one device)")
            # what crossed to the root: 10 B per hit packed (+ the bucket starts), 12 B raw
            node.scan_score_device(20)
            raw = node.gather(0, pos16=False)["bytes_to_root"]
            packed = node.gather(0, pos16=True)["bytes_to_root"]
            if raw > 100_000:
                assert packed < raw * 0.87, (raw, packed)


@pytest.mark.gpu
def test_node_equals_single_engine_tables(oracle):
    """The same genome through Engine.arena (one GPU, the N = 1 path) and through a 4-device node: per contig the very
    same bytes; and scanning twice / gathering twice changes nothing."""
    from cropsr_amd import Engine, node as nd
    rng = np.random.default_rng(77)
    contigs = _genome(rng, [900_000, 40_000, 350_000, 7, 650_000, 12_000])
    with Engine(0) as eng:
        arena = eng.arena(contigs)
        one = arena.scan_score(20, want_pre=False)
        want = [one.contig(k) for k in range(len(contigs))]
        arena.close()
    with nd.Node([0, 0, 0, 0]) as node:
        node.load(contigs)
        digests = set()
        for rep in range(3):
            hits = node.scan(20, pos16=rep != 1)
            d = hashlib.sha256()
            for k in range(len(contigs)):
                got = hits.contig(k)
                for key in ("pos_plus", "score_plus", "pos_minus", "score_minus"):
                    assert (bits(got[key]) == bits(want[k][key])).all(), (rep, k, key)
                    d.update(bits(got[key]).tobytes())
            digests.add(d.hexdigest())
        assert len(digests) == 1
        # a gather without a fresh scan is a state the library accepts (the tables are still there); one before any scan is not
        node.load(contigs)
        with pytest.raises(Exception) as e:
            node.gather(0)
        assert "scan first" in str(e.value)
        with pytest.raises(Exception):
            node.fetch()


@pytest.mark.gpu
@pytest.mark.slow
def test_node_tair10_like_four_logical_devices(oracle):
    """VERDICT r04 #1's acceptance test: the TAIR10-like genome (BASELINE.json configs[2] stand-in, 119.7 Mb, 7 contigs:
    every chromosome straddles a share boundary or fills most of a share) over the node handle on {0, 0, 0, 0} ==
    the N = 1 tables of the same genome == the oracle, every one of the 7.7 M hits, by SHA-256 per contig."""
    from concurrent.futures import ThreadPoolExecutor
    import bench_workload as bw
    from cropsr_amd import Engine, node as nd

    def digest(h):
        d = hashlib.sha256()
        for key in ("pos_plus", "score_plus", "pos_minus", "score_minus"):
            d.update(np.ascontiguousarray(h[key]).tobytes())
        return d.hexdigest()

    wl = bw.tair10_like()
    strings = [wl.contig_string(k) for k in range(len(wl.specs))]
    threads = max(2, min(16, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(threads) as pool:
        want = [pool.submit(lambda t=s: digest(oracle.scan_score(t, 20))) for s in strings]
        with Engine(0) as eng:
            arena = eng.arena(strings)
            one = arena.scan_score(20, want_pre=False)
            n1 = [digest(one.contig(k)) for k in range(len(strings))]
            n1_hits = one.n_plus + one.n_minus
            arena.close()
        with nd.Node([0, 0, 0, 0]) as node:
            node.load(strings)
            plan = node.plan()
            assert len(plan) == len(strings) + 3  # three cuts
            hits = node.scan(20)
            got = [digest(hits.contig(k)) for k in range(len(strings))]
            stats = node.gather_stats()
            per_dev = [node.arena_stats(k) for k in range(4)]
        want = [f.result() for f in want]
    assert got == n1, [k for k in range(len(strings)) if got[k] != n1[k]]
    assert got == want, [k for k in range(len(strings)) if got[k] != want[k]]
    assert hits.n_plus + hits.n_minus == n1_hits > 7_000_000
    chars = [d["n_chars"] for d in per_dev]
    assert max(chars) - min(chars) < 10_000  # equal shares
    print("node tair10-like on 4 logical devices: %d hits, %d B to the root (%.2f B per peer hit), exchange %.2f ms, %s"
          % (n1_hits, stats["bytes_to_root"], stats["bytes_to_root"] / (n1_hits * 0.75), stats["ms_exchange"], stats["transport"]))


def _node_rccl_world1(out_path):
    """A one-device node with CRP_NODE_TRANSPORT=rccl: ncclCommInitAll (world 1) and an empty send/recv group on the real
    RCCL, in a process of its own (RCCL's teardown belongs to that process)."""
    os.environ["CRP_NODE_TRANSPORT"] = "rccl"
    import json
    from cropsr_amd import node as nd
    from oracle import oracle as orc
    rng = np.random.default_rng(3)
    contigs = _genome(rng, [200_000, 3_000])
    with nd.Node([0]) as node:
        node.load(contigs)
        hits = node.scan(20)
        st = node.gather_stats()
        n = _check_against_oracle(hits, contigs, orc, 20, "rccl world 1")
    with open(out_path, "w") as f:
        json.dump({"hits": n, "transport": st["transport"], "bytes_to_root": st["bytes_to_root"]}, f)


@pytest.mark.gpu
def test_node_on_rccl_with_one_device(tmp_path):
    import json
    out = tmp_path / "node_rccl.json"
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_node; test_node._node_rccl_world1(%r)" % (
        ROOT, os.path.join(ROOT, "tests"), str(out))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(out.read_text())
    assert d["hits"] > 10_000 and d["transport"].startswith("RCCL") and d["bytes_to_root"] == 0


def _bench(*argv, timeout=900):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CROPSR_LAUNCHED")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_single_process_four_logical_devices():
    """`bench.py --gpus 4 --single-process`: the same line through the node handle -- ONE process, no launcher, no sockets.
    On the one-GPU box the four devices are GPU 0 four times (--share-gpu0): four contexts, four arenas, the exchange as
    device-to-device copies; the strong block's stitched tables equal device 0's own N = 1 scan, contig by contig."""
    d = _bench("--gpus", "4", "--single-process", "--share-gpu0", "--scale", "0.05", "--steps", "3", "--warmup", "1")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 4 and d["scaling"] == "weak" and d["value"] > 0 and d["gatherv_ok"] is True
    assert "ONE process" in d["config"]["parallelism"] and d["config"]["genomes"] == 4
    assert [r["rank"] for r in d["per_rank"]] == [0, 1, 2, 3] and all(r["kernel_ms"] > 0 for r in d["per_rank"])
    chars = [r["characters_with_halos"] for r in d["per_rank"]]
    assert max(chars) - min(chars) <= 6 * 4096
    g = d["gatherv"]
    assert g["transport"] == "device-to-device copies" and 0 < g["bytes_to_root"] < 0.86 * g["raw_u32_positions"]["bytes_to_root"]
    assert d["value_with_final_gatherv"] < d["value"]
    st = d["strong"]
    assert st["digest_ok"] is True and st["genomes"] == 1 and st["contigs_cut"] >= 1 and st["kept_hits"] == st["n1"]["kept_hits"]
    assert st["bytes_to_root"] > 0 and st["value"] < st["value_scan_only"] and len(st["per_rank"]) == 4
    th = st["tables_to_the_host"]
    assert th["every_device_over_its_own_link"]["ms"] > 0 and th["gatherv_to_device_0_then_one_link"]["ms_gather"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12


@pytest.mark.gpu
def test_bench_real_fasta_and_bytes_per_survey_8d(tmp_path):
    """--fasta: a REAL genome through the bench (SURVEY.md 8d "real FASTA may be substituted"): the reference's own sample
    (sample_data/sample_genome.fa, committed here gzipped as a fixture) gives its 17 314 kept hits, `data` says real and the
    workload names the file.  And the bytes of the roofline follow SURVEY.md 8(d) for every input: the two 1-bit planes
    count for the soft-masked sample and are left out for the entirely upper-case E. coli-like genome (cfg 2)."""
    import gzip
    from conftest import GOLDEN
    fa = tmp_path / "sample_genome.fa"
    with gzip.open(os.path.join(GOLDEN, "sample_genome.fa.gz"), "rb") as f:
        fa.write_bytes(f.read())
    d = _bench("--fasta", str(fa), "--steps", "3", "--warmup", "1", "--cpu-sample-bases", "20000", "--offtarget-steps", "1")
    assert d["data"] == "real" and d["config"]["kept_hits_total"] == 17314 and d["config"]["workload"].startswith("sample_genome.fa")
    assert d["config"]["bases_total"] == 230218 and "annotate" not in d and d["cpu_baseline"]["value"] > 0
    r = d["roofline"]
    n = r["characters"]["N"]
    assert n == 230218 + 4 and r["characters"]["other"] > 20000  # 12.9 % lower case
    assert r["algorithmic_bytes_per_launch"] == (n + 3) // 4 + 2 * ((n + 7) // 8) + 12 * 17314
    # with its GFF: the annotate block joins the user's own annotation
    d = _bench("--fasta", str(fa), "--gff", os.path.join(GOLDEN, "sample_head.gff"), "--steps", "2", "--warmup", "1",
               "--cpu-sample-bases", "0", "--offtarget-steps", "0")
    assert d["annotate"]["gff"]["data"] == "sample_head.gff" and d["annotate"]["roofline"]["frac"] > 0
    # cfg 2: upper-case ACGT only
    d = _bench("--workload", "ecoli", "--steps", "3", "--warmup", "1", "--cpu-sample-bases", "0", "--offtarget-steps", "0",
               "--annotate-steps", "0")
    r = d["roofline"]
    n, h = r["characters"]["N"], d["config"]["kept_hits_total"]
    assert n == 4641652 + 4 and r["characters"]["other"] == 4
    assert r["algorithmic_bytes_per_launch"] == (n + 3) // 4 + 12 * h and "12*H" in r["algorithmic_bytes"] and "2*ceil" not in r["algorithmic_bytes"]
    assert "three_launch" in r and r["all_kernels_ms"] == r["kernel_ms"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 3, 4])
def test_node_offtarget_and_annotation_vs_oracle(oracle, world, tmp_path):
    """The two opt-in steps through the node handle (crp_node_offtarget, crp_node_annotate): every device adds the sites it
    OWNS (a hit inside a halo is its neighbour's), the site histograms are summed over the devices, every device joins its
    own pieces with the track of THOSE pieces; counts and label-set ids travel with the gatherv.  Per contig: the same
    counts as the oracle's genome-wide enumeration and the same ids as the oracle's numpy join -- whatever the cut."""
    from cropsr_amd import annotate, node as nd
    from oracle import annotate_oracle
    rng = np.random.default_rng(500 + world)
    lengths = [300_000, 5, 70_000, 123_457, 40_000]
    contigs = _genome(rng, lengths)
    # repeats across the cuts: the same 40 kb block in three contigs, so seeds recur on several devices
    block = rng.choice(ALPHA, 40_000).tobytes()
    contigs[0] = contigs[0][:100_000] + block + contigs[0][140_000:]
    contigs[3] = contigs[3][:60_000] + block + contigs[3][100_000:]
    contigs[4] = b"'" + block + b"')]"
    gff = tmp_path / "node.gff"
    rows = ["##gff-version 3"]
    for k, n in enumerate(lengths):
        for g in range(max(1, n // 20_000)):
            a = 1 + g * 20_000 + int(rng.integers(0, 5_000))
            b = min(n, a + int(rng.integers(300, 12_000)))
            if b > a:
                rows.append("c%d\tsrc\tgene\t%d\t%d\t.\t+\t.\tID=g%d_%d" % (k, a, b, k, g))
                rows.append("c%d\tsrc\tCDS\t%d\t%d\t.\t+\t0\tID=g%d_%d.cds" % (k, a + 10, max(a + 11, b - 10), k, g))
    gff.write_text("\n".join(rows) + "\n")
    ann = annotate.Annotation(str(gff))
    req = annotate.Request(ann, ["c%d" % k for k in range(len(contigs))], 1)
    want_ot = oracle.offtarget_genome(contigs, 20)
    with nd.Node([0] * world) as node:
        node.load(contigs)
        for kw in ({}, {"pos16": False, "root": world - 1}, {"to_host": True}):
            hits = node.scan(20, offtarget=True, annotation=req, **kw)
            n_feat = 0
            for k, c in enumerate(contigs):
                got = hits.contig(k)
                want = oracle.scan_score(c, 20)
                for key in ("pos_plus", "score_plus", "pos_minus", "score_minus"):
                    assert (bits(got[key]) == bits(want[key])).all(), (world, k, key)
                assert (got["ot_plus"] == want_ot[k]["ot_plus"]).all() and (got["ot_minus"] == want_ot[k]["ot_minus"]).all(), (world, k)
                fp, fm = annotate_oracle.host_join(ann, "c%d" % k, 0, 1, got, 20, len(c))
                assert (got["feat_plus"] == fp).all() and (got["feat_minus"] == fm).all(), (world, k)
                n_feat += int((fp != annotate.NO_FEATURE).sum() + (fm != annotate.NO_FEATURE).sum())
            assert n_feat > 5_000
        # the seeds may also come from the planes (no CRP_SCAN_SEEDS): same counts
        node.scan_score_device(20)
        sites = node.offtarget(20)
        node.gather(0, offtarget=True)
        again = node.fetch()
        assert sites > 20_000 and (again.ot_plus == hits.ot_plus).all() and (again.ot_minus == hits.ot_minus).all()
        # a gather that asks for columns nobody computed is refused
        node.scan_score_device(20)
        with pytest.raises(Exception):
            node.gather(0, features=True)
    ann.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 4])
def test_node_several_arenas_per_device_vs_oracle(oracle, world, tmp_path):
    """A share larger than one arena (forced here with a tiny per-arena limit; for real beyond 2^31 characters per device --
    the reference reads any genome whole, CROPSR.py:59): the share goes on in further arenas, a piece no arena can hold is cut
    again with halos, and scan, ownership cuts, gatherv (packed, raw, to the host, another root, pre-sigmoid), off-target
    counts and label-set ids all run arena by arena -- per contig the oracle's rows, whatever the limit."""
    from cropsr_amd import annotate, node as nd
    from oracle import annotate_oracle
    rng = np.random.default_rng(900 + world)
    lengths = [300_000, 5, 0, 70_000, 9_000, 123_457, 64, 1, 40_000]
    contigs = _genome(rng, lengths)
    block = rng.choice(ALPHA, 30_000).tobytes()  # a repeat across arenas and devices: seeds recur
    contigs[0] = contigs[0][:100_000] + block + contigs[0][130_000:]
    contigs[5] = contigs[5][:60_000] + block + contigs[5][90_000:]
    gff = tmp_path / "arenas.gff"
    rows = ["##gff-version 3"]
    for k, n in enumerate(lengths):
        for g in range(n // 15_000):
            a = 1 + g * 15_000 + int(rng.integers(0, 4_000))
            b = min(n, a + int(rng.integers(300, 12_000)))
            if b > a:
                rows.append("c%d\tsrc\tgene\t%d\t%d\t.\t+\t.\tID=g%d_%d" % (k, a, b, k, g))
    gff.write_text("\n".join(rows) + "\n")
    ann = annotate.Annotation(str(gff))
    req = annotate.Request(ann, ["c%d" % k for k in range(len(contigs))], 1)
    want_ot = oracle.offtarget_genome(contigs, 20)
    with nd.Node([0] * world) as node:
        with pytest.raises(Exception):
            node.set_option(arena_words=3)  # (not even one piece between two halos)
        for words in (600, 137, 2000):  # 38 400, 8 768 and 128 000 characters per arena
            node.set_option(arena_words=words)
            node.load(contigs)
            plan = node.plan()
            n_arenas = [node.n_arenas(k) for k in range(world)]
            assert sum(n_arenas) == len({(p["device"], p["arena"]) for p in plan}) and max(n_arenas) > 1, (words, n_arenas)
            assert len(plan) > len(contigs) + world - 1  # pieces were cut again at arena ends
            for k, c in enumerate(contigs):  # the pieces still cover every contig once, in order
                mine = [p for p in plan if p["contig"] == k]
                assert mine[0]["start"] == 0 and mine[-1]["end"] == len(c) and all(a["end"] == b["start"] for a, b in zip(mine, mine[1:]))
            st = node.arena_stats(0)
            assert st["n_arenas"] == n_arenas[0] and st["n_chars"] <= n_arenas[0] * words * 64
            for l, kw in ((20, {}), (20, {"pos16": False}), (20, {"to_host": True}), (20, {"pre": True, "root": world - 1}),
                          (23, {"to_host": True, "pre": True}), (7, {"pos16": False})):
                hits = node.scan(l, **kw)
                _check_against_oracle(hits, contigs, oracle, l, (world, words, l, kw), pre=kw.get("pre", False))
                if kw.get("to_host"):
                    assert node.count_scored() == int((hits.score_plus != -1).sum() + (hits.score_minus != -1).sum())
            for kw in ({}, {"to_host": True}, {"pos16": False, "root": world - 1}):
                hits = node.scan(20, offtarget=True, annotation=req, **kw)
                n_feat = 0
                for k, c in enumerate(contigs):
                    got, want = hits.contig(k), oracle.scan_score(c, 20)
                    for key in ("pos_plus", "score_plus", "pos_minus", "score_minus"):
                        assert (bits(got[key]) == bits(want[key])).all(), (world, words, kw, k, key)
                    assert (got["ot_plus"] == want_ot[k]["ot_plus"]).all() and (got["ot_minus"] == want_ot[k]["ot_minus"]).all(), (world, words, kw, k)
                    fp, fm = annotate_oracle.host_join(ann, "c%d" % k, 0, 1, got, 20, len(c))
                    assert (got["feat_plus"] == fp).all() and (got["feat_minus"] == fm).all(), (world, words, kw, k)
                    n_feat += int((fp != annotate.NO_FEATURE).sum() + (fm != annotate.NO_FEATURE).sum())
                assert n_feat > 5_000
        node.set_option(arena_words=0)  # back to the library's limit: one arena per device again
        node.load(contigs)
        assert [node.n_arenas(k) for k in range(world)] == [1 if any(p["device"] == k for p in node.plan()) else 0 for k in range(world)]
        _check_against_oracle(node.scan(20), contigs, oracle, 20, (world, "default"))
    ann.close()


@pytest.mark.gpu
@pytest.mark.slow
def test_node_maize_size_genome_on_one_device_equals_engine_genome():
    """VERDICT r05 #3's full-size run: a 2.4 Gb maize-size stand-in (more than the 2^31 characters one arena addresses)
    through Node([0]) -- one device, two arenas inside the handle -- == Engine.genome's tables (the path that already spread
    a genome over arenas) by SHA-256 per contig, gathered on the device and over the host links."""
    import bench_workload as bw
    from cropsr_amd import Engine, node as nd

    def digest(h):
        d = hashlib.sha256()
        for key in ("pos_plus", "score_plus", "pos_minus", "score_minus"):
            d.update(np.ascontiguousarray(h[key]).tobytes())
        return d.hexdigest()

    wl = bw.maize_like()
    strings = [wl.contig_string(k) for k in range(len(wl.specs))]
    total = sum(s.size for s in strings)
    assert total > (1 << 31)
    with Engine(0) as eng:
        genome = eng.genome(strings)
        assert len(genome.arenas) >= 2
        one = genome.scan_score(20)
        want = [digest(one.contig(k)) for k in range(len(strings))]
        n_hits = one.n_plus + one.n_minus
        del one
        genome.close()
    with nd.Node([0]) as node:
        node.load(strings)
        assert node.n_arenas(0) >= 2 and node.arena_stats(0)["n_chars"] == total
        for kw in ({}, {"to_host": True}):
            hits = node.scan(20, **kw)
            got = [digest(hits.contig(k)) for k in range(len(strings))]
            assert got == want, (kw, [k for k in range(len(strings)) if got[k] != want[k]][:10])
            assert hits.n_plus + hits.n_minus == n_hits
            del hits
    print("node maize-like on one device: %d characters, %d arenas, %d hits" % (total, 2, n_hits))


def _more_devices_than_pieces(out_path):
    """a genome with fewer pieces than devices -- some devices hold no arena and have only QUEUED the zeroing of
    their site histogram -- with the histograms summed through device 0 (CRP_NODE_TRANSPORT=peer): the oracle's counts, on a
    node that is used for three genomes in a row (a stale histogram on an idle device would show in the second and third)."""
    os.environ["CRP_NODE_TRANSPORT"] = "peer"
    from cropsr_amd import node as nd
    from oracle import oracle as orc
    rng = np.random.default_rng(12)
    total = 0
    with nd.Node([0] * 6) as node:
        for lengths in ([3_000, 2_500], [150_000], [1_000]):
            contigs = _genome(rng, lengths)
            want_ot = orc.offtarget_genome(contigs, 20)
            node.load(contigs)
            assert sum(1 for k in range(6) if node.n_arenas(k) == 0) >= 1 or len(lengths) == 1 and lengths[0] > 100_000
            hits = node.scan(20, offtarget=True)
            total += _check_against_oracle(hits, contigs, orc, 20, ("idle devices", lengths))
            for k in range(len(contigs)):
                got = hits.contig(k)
                assert (got["ot_plus"] == want_ot[k]["ot_plus"]).all() and (got["ot_minus"] == want_ot[k]["ot_minus"]).all(), (lengths, k)
    with open(out_path, "w") as f:
        f.write(str(total))


@pytest.mark.gpu
def test_node_offtarget_with_more_devices_than_pieces(tmp_path):
    out = tmp_path / "idle.txt"
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_node; test_node._more_devices_than_pieces(%r)" % (
        ROOT, os.path.join(ROOT, "tests"), str(out))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert int(out.read_text()) > 5_000


@pytest.mark.gpu
@pytest.mark.parametrize("name,extra", [("sample", ()), ("multi", ()), ("mixed", ("--score-finalize", "host")),
                                        ("mixed", ("-l", "23")), ("rightend", ("-l", "64"))])
def test_cli_devices_csv_bytes_equal_reference(name, extra, manifest, tmp_path, monkeypatch):
    """`python -m cropsr_amd --devices 0,0,0`: the CLI's whole job through the node handle, ONE process -- the CSV is the
    reference's, byte for byte (fixtures made by the real CROPSR.py), like the one-GPU run's."""
    from conftest import golden_fasta_path, read_golden_csv, run_cli
    from cropsr_amd import cli
    guide_len = int(extra[1]) if extra[:1] == ("-l",) else None
    host = "host" in extra
    if name == "multi":
        monkeypatch.setenv("CROPSR_GATHER", "host")  # every device's rows over its own link (CRP_NODE_HOST_GATHER)
    backend = cli.NodeBackend([0, 0, 0], finalize="host" if host else "gpu")
    try:
        got, _ = run_cli(tmp_path, monkeypatch, golden_fasta_path(name, tmp_path), backend, manifest["seed"], extra=extra)
    finally:
        backend.close()
    if host:
        # the f64 column that crossed the node is the pre-sigmoid sum and THIS host's numpy applies CROPSR.py:313: the
        # reference's bytes where np.exp is glibc's; elsewhere (numpy's AVX-512 exp) the same rows, last bits this host's
        from oracle import oracle
        x = np.random.default_rng(5).uniform(-9.3, 17.3, 200000)
        if (np.exp(x).view(np.uint64) == oracle.exp(x).view(np.uint64)).all():
            assert got == read_golden_csv(name)
        else:
            assert got.count(b"\r\n") == read_golden_csv(name).count(b"\r\n") and len(got) > 0
    else:
        assert got == read_golden_csv(name, guide_len)


@pytest.mark.gpu
def test_cli_devices_offtarget_annotate_equals_one_device(tmp_path, monkeypatch, manifest):
    """--devices with --offtarget --annotate == the same run on one device (EngineBackend), byte for byte."""
    from conftest import GOLDEN, run_cli
    from cropsr_amd import cli
    gff = tmp_path / "mixed.gff"
    gff.write_text("##gff-version 3\nmix\tsrc\tgene\t40\t410\t.\t+\t.\tID=g1;Name=L1\nmix\tsrc\tCDS\t95\t105\t.\t+\t0\tID=g1.cds1\n"
                   "mix\tsrc\tgene\t400\t1123\t.\t-\t.\tID=g2\ntail\tsrc\tgene\t1\t60\t.\t+\t.\tID=t1\n")
    extra = ("-g", str(gff), "--annotate", "--offtarget")
    fa = os.path.join(GOLDEN, "probe_mixed.fa")
    outs = []
    for backend in (cli.EngineBackend(0), cli.NodeBackend([0, 0, 0, 0])):
        d = tmp_path / type(backend).__name__
        d.mkdir()
        try:
            outs.append(run_cli(d, monkeypatch, fa, backend, manifest["seed"], extra=extra)[0])
        finally:
            backend.close()
    assert outs[0] == outs[1] and b"gene:g1" in outs[0] and outs[0].count(b"\r\n") > 100


# ------------------------------------------------------------------ boxes with two or more GPUs
def _second_gpu():
    """True when HIP device 1 can be opened (the driver's multi-GPU node; the one-GPU boxes skip these tests)."""
    from cropsr_amd import _native as nat
    import ctypes
    h = ctypes.c_void_p()
    if nat.lib().crp_init(1, ctypes.byref(h)) != nat.CRP_OK:
        return False
    nat.lib().crp_destroy(h)
    return True


@pytest.mark.gpu
def test_node_two_real_devices_on_rccl(oracle, tmp_path):
    """Two DISTINCT devices (skipped on a one-GPU box): the node's exchange on the real transport -- ncclCommInitAll and
    one grouped send/recv between two GPUs -- and as device-to-device copies over xGMI, with every column (positions packed
    and raw, scores, off-target counts, label-set ids): the oracle's rows either way (nothing but a one-rank
    communicator had ever carried these columns)."""
    if not _second_gpu():
        pytest.skip("needs two GPUs")
    from cropsr_amd import annotate, node as nd
    from oracle import annotate_oracle
    rng = np.random.default_rng(42)
    contigs = _genome(rng, [900_000, 40_000, 350_000, 7, 650_000])
    gff = tmp_path / "two.gff"
    gff.write_text("##gff-version 3\n" + "".join("c%d\tsrc\tgene\t%d\t%d\t.\t+\t.\tID=g%d_%d\n" % (k, a, a + 5000, k, a)
                                               for k in (0, 2, 4) for a in range(1000, 300_000, 9000)))
    ann = annotate.Annotation(str(gff))
    req = annotate.Request(ann, ["c%d" % k for k in range(len(contigs))], 1)
    want_ot = oracle.offtarget_genome(contigs, 20)
    with nd.Node([0, 1]) as node:
        node.load(contigs)
        for kw, transport in (({}, "RCCL"), ({"pos16": False}, "RCCL"), ({"peer_copy": True}, "device-to-device"), ({"root": 1}, "RCCL")):
            hits = node.scan(20, offtarget=True, annotation=req, **kw)
            assert node.gather_stats()["transport"].startswith(transport), (kw, node.gather_stats())
            for k, c in enumerate(contigs):
                got, want = hits.contig(k), oracle.scan_score(c, 20)
                for key in ("pos_plus", "score_plus", "pos_minus", "score_minus"):
                    assert (bits(got[key]) == bits(want[key])).all(), (kw, k, key)
                assert (got["ot_plus"] == want_ot[k]["ot_plus"]).all() and (got["ot_minus"] == want_ot[k]["ot_minus"]).all(), (kw, k)
                fp, fm = annotate_oracle.host_join(ann, "c%d" % k, 0, 1, got, 20, len(c))
                assert (got["feat_plus"] == fp).all() and (got["feat_minus"] == fm).all(), (kw, k)
    ann.close()


@pytest.mark.gpu
def test_cli_two_real_gpus_rccl_equals_one_process(tmp_path, manifest):
    """`python -m cropsr_amd --gpus 2 --offtarget --annotate` on two DISTINCT GPUs (skipped on a one-GPU box): one process per
    GPU, the tables -- with the off-target counts and the label-set ids -- cross on RCCL between two ranks
    (crp_gather_hits with CRP_GATHER_OFFTARGET | CRP_GATHER_FEATURES | CRP_GATHER_POS16), the site histogram is all-reduced:
    the bytes of the one-process run."""
    if not _second_gpu():
        pytest.skip("needs two GPUs")
    from conftest import GOLDEN
    gff = tmp_path / "mixed.gff"
    gff.write_text("##gff-version 3\nmix\tsrc\tgene\t40\t410\t.\t+\t.\tID=g1;Name=L1\nmix\tsrc\tCDS\t95\t105\t.\t+\t0\tID=g1.cds1\n"
                   "mix\tsrc\tgene\t400\t1123\t.\t-\t.\tID=g2\ntail\tsrc\tgene\t1\t60\t.\t+\t.\tID=t1\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CROPSR_LAUNCHED", "CROPSR_GATHER")}
    env.update(PYTHONPATH=ROOT, CROPSR_DIST_MAX_PIECE="100")
    common = ["-f", os.path.join(GOLDEN, "probe_mixed.fa"), "-g", str(gff), "--cas9", "--seed", str(manifest["seed"]),
              "--offtarget", "--annotate"]
    outs = []
    for tag, extra in (("two", ["--gpus", "2"]), ("one", []), ("node", ["--devices", "0,1"])):
        out = tmp_path / (tag + ".csv")
        d = tmp_path / tag
        d.mkdir()
        p = subprocess.run([sys.executable, "-m", "cropsr_amd", "-o", str(out)] + common + extra, capture_output=True, text=True,
                           timeout=900, cwd=str(d), env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        outs.append(out.read_bytes())
    assert outs[0] == outs[1] == outs[2] and b"gene:g1" in outs[0]


@pytest.mark.gpu
def test_node_randomised_genomes_vs_oracle(oracle):
    """Seeded fuzz of the node handle: random genomes (contig counts 0..12, lengths around the cut's thresholds -- the 4 096
    characters below which nothing is cut, the halo, word and tile borders --, random alphabets and decoration), 1..7 logical
    devices, random guide length / packing / root / pre-sigmoid column; one node per world size serves many genomes.  Every
    contig's rows equal the oracle's.  CROPSR_FUZZ_TRIALS raises the trial count for a soak."""
    from conftest import fuzz_settings
    from cropsr_amd import node as nd
    trials, seed, tick = fuzz_settings(60, 20261005)
    rng = np.random.default_rng(seed)
    alphabets = [b"ACGT", b"ACGTacgtN", b"GGCC", b"ACGTUZuzN')],", b"GGGGGGCCCCCCAT"]
    anchors = [0, 1, 30, 64, 127, 128, 129, 255, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 16384, 65535, 65536, 65537, 131072, 200000]
    nodes = {}
    total = cuts = 0
    try:
        for trial in range(trials):
            tick("node", trial)
            world = int(rng.integers(1, 8))
            if world not in nodes:
                nodes[world] = nd.Node([0] * world)
            node = nodes[world]
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
            node.load(contigs)
            cuts += len(node.plan()) - len(contigs)
            hits = node.scan(l, root=int(rng.integers(0, world)), pre=pre, pos16=bool(rng.random() < 0.7), to_host=bool(rng.random() < 0.3))
            total += _check_against_oracle(hits, contigs, oracle, l, (trial, world, l, pre), pre=pre)
    finally:
        for node in nodes.values():
            node.close()
    assert total > 30000 * trials // 60 and cuts > trials // 4


@pytest.mark.gpu
def test_integration_md_node_snippet_runs(oracle):
    """The ctypes patch INTEGRATION.md shows for reaching the whole node from the one-process reference script is executed as
    printed (the eight devices being GPU 0 eight times here) on a small genome: the tables it ends up with are the oracle's."""
    import ctypes
    import types
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    start = text.index("### The whole node from the one process")
    code = text[text.index("```python", start) + len("```python"):]
    code = code[:code.index("```")]
    assert "crp_node_init(8, ids" in code and "crp_node_fetch" in code
    code = code.replace("(*range(8))", "(*([0] * 8))")
    rng = np.random.default_rng(31)
    contigs = _genome(rng, [400_000, 9_000, 250_000])
    env = {"ctypes": ctypes, "np": np, "_crp": ctypes.CDLL(os.path.join(ROOT, "cropsr_amd", "libcropsr_hip.so")),
           "fasta_file": {"c%d" % k: c.decode("ascii") for k, c in enumerate(contigs)}, "args": types.SimpleNamespace(l=20)}
    exec(code, env)
    per = list(env["per_contig"])
    a = b = 0
    for k, c in enumerate(contigs):
        want = oracle.scan_score(c, 20)
        n_p, n_m = per[2 * k], per[2 * k + 1]
        assert (env["pos_p"][a:a + n_p] == want["pos_plus"]).all() and (bits(env["sc_p"][a:a + n_p]) == bits(want["score_plus"])).all()
        assert (env["pos_m"][b:b + n_m] == want["pos_minus"]).all() and (bits(env["sc_m"][b:b + n_m]) == bits(want["score_minus"])).all()
        a, b = a + n_p, b + n_m
    assert a == env["n_plus"].value and b == env["n_minus"].value and a + b > 30_000
    env["_crp"].crp_node_destroy(env["node"])


def _threaded_scan_on_one_gpu(out_path):
    """CRP_NODE_SCAN_THREADS=1: the per-device scan threads (what a node of DISTINCT GPUs uses) forced on for seven logical
    devices on GPU 0 -- many scans back to back, a reload in between, every result against the oracle."""
    os.environ["CRP_NODE_SCAN_THREADS"] = "1"
    from cropsr_amd import node as nd
    from oracle import oracle as orc
    rng = np.random.default_rng(8)
    total = 0
    with nd.Node([0] * 7) as node:
        for lengths in ([500_000, 30_000, 260_000], [90_000] * 9, [1_200_000]):
            contigs = _genome(rng, lengths)
            node.load(contigs)
            for rep in range(40):
                node.scan_score_device(20)
            for l in (20, 23):
                total += _check_against_oracle(node.scan(l, pos16=bool(l == 20)), contigs, orc, l, ("threads", lengths, l))
    with open(out_path, "w") as f:
        f.write(str(total))


@pytest.mark.gpu
def test_node_threaded_scan_forced_on_one_gpu(tmp_path):
    """On a one-GPU box the node scans its logical devices one after the other (N kernels launched at once on one GPU only get
    in each other's way); the worker threads a real multi-GPU node uses are forced on here so that they stay under test."""
    out = tmp_path / "threads.txt"
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_node; test_node._threaded_scan_on_one_gpu(%r)" % (
        ROOT, os.path.join(ROOT, "tests"), str(out))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert int(out.read_text()) > 300_000


@pytest.mark.gpu
def test_bench_single_process_one_device_and_strong_only():
    """`--single-process` with ONE device is the N = 1 line through the node handle (CPU baseline included); `--strong-only`
    is what rank 0 of a process-per-GPU run starts as its closing node block: just the strong block, digest-checked."""
    d = _bench("--gpus", "1", "--single-process", "--scale", "0.01", "--steps", "3", "--warmup", "1", "--cpu-sample-bases", "20000")
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["gpu_over_cpu"] > 10
    assert "gatherv" not in d and "strong" not in d and d["roofline"]["frac"] > 0 and d["roofline"]["traffic_source"]
    d = _bench("--gpus", "3", "--single-process", "--strong-only", "--share-gpu0", "--scale", "0.02", "--steps", "2", "--warmup", "1")
    st = d["strong"]
    assert set(d) == {"strong", "n_gpus", "config"} and st["digest_ok"] is True and len(st["per_rank"]) == 3 and st["devices"] == [0, 0, 0]
    assert st["tables_to_the_host"]["every_device_over_its_own_link"]["ms"] > 0


@pytest.mark.gpu
def test_integration_md_seam_snippets_run(oracle):
    """The two ctypes patches INTEGRATION.md section B shows for `CROPSR.py` -- seam 2 (`rs1_score`) and seam 1 (the scan loops) --
    executed as printed: `rs1_score` returns the oracle's bits for a batch (incl. the rows OpenBLAS sums in its tail order), and
    the tables seam 1 ends up with are the oracle's, contig by contig."""
    import ctypes
    import types
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()

    def block(after):
        at = text.index(after)
        code = text[text.index("```python", at) + len("```python"):]
        return code[:code.index("```")]
    seam2 = block("### Seam 2").replace('"/path/to/cropsr_amd/libcropsr_hip.so"', repr(os.path.join(ROOT, "cropsr_amd", "libcropsr_hip.so")))
    seam1 = block("### Seam 1")
    rng = np.random.default_rng(77)
    contigs = _genome(rng, [120_000, 3_000, 64])
    env = {"fasta_file": {"k%d" % k: c.decode("ascii") for k, c in enumerate(contigs)}, "args": types.SimpleNamespace(l=20)}
    exec(seam2, env)
    rows = rng.choice(np.frombuffer(b"ATCGN", dtype=np.uint8), size=(1003, 30))
    got = env["rs1_score"](rows)
    _, want = oracle.rs1_batch(rows)
    assert (bits(got) == bits(want)).all()
    assert (bits(env["rs1_score"](rows[:1])) == bits(oracle.rs1_batch(rows[:1])[1])).all()
    exec(seam1, env)
    offs, pos_p, sc_p, pos_m, sc_m = env["offs"], env["pos_p"], env["sc_p"], env["pos_m"], env["sc_m"]
    for k, c in enumerate(contigs):
        want = oracle.scan_score(c, 20)
        sel = (pos_p >= offs[k]) & (pos_p < offs[k] + len(c))
        assert (pos_p[sel] - offs[k] == want["pos_plus"]).all() and (bits(sc_p[sel]) == bits(want["score_plus"])).all()
        sel = (pos_m >= offs[k]) & (pos_m < offs[k] + len(c))
        assert (pos_m[sel] - offs[k] == want["pos_minus"]).all() and (bits(sc_m[sel]) == bits(want["score_minus"])).all()
    env["_crp"].crp_arena_destroy(env["arena"])
    env["_crp"].crp_destroy(env["_ctx"])


@pytest.mark.gpu
def test_cli_devices_command_line(tmp_path, manifest):
    """The program itself: `python -m cropsr_amd --devices 0,0 ...` (no launcher, one process) writes the reference's CSV for
    the sample genome, with and without CROPSR_GATHER=host; `--devices` together with `--gpus N` is refused."""
    from conftest import GOLDEN, golden_fasta_path, read_golden_csv
    fa = golden_fasta_path("sample", tmp_path)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CROPSR_LAUNCHED", "CROPSR_GATHER")}
    env["PYTHONPATH"] = ROOT
    common = ["-f", fa, "-g", os.path.join(GOLDEN, "sample_head.gff"), "--cas9", "--seed", str(manifest["seed"])]
    for tag, extra_env in (("rccl", {}), ("host", {"CROPSR_GATHER": "host"})):
        d = tmp_path / tag
        d.mkdir()
        out = d / "out.csv"
        p = subprocess.run([sys.executable, "-m", "cropsr_amd", "--devices", "0,0", "-o", str(out)] + common, capture_output=True,
                           text=True, timeout=600, cwd=str(d), env=dict(env, **extra_env))
        assert p.returncode == 0, p.stderr[-2000:]
        assert out.read_bytes() == read_golden_csv("sample") and manifest["cases"]["sample"]["stdout"] in p.stdout
    p = subprocess.run([sys.executable, "-m", "cropsr_amd", "--devices", "0,0", "--gpus", "2", "-o", str(tmp_path / "x.csv")] + common,
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert p.returncode != 0 and "give one of them" in p.stderr
