"""The N > 1 path on CPU, without PyTorch: contig partitioning, halo cuts, the gatherv of per-rank
hit tables and the off-target histogram sum (cropsr_amd.parallel) over the control sockets of
cropsr_amd.rendezvous, world_size 2 and 3, as separate processes.  Per-rank tables come from the
oracle here (no GPU in this suite); on the GPUs the same flow moves the HIP tables over RCCL inside
libcropsr_hip.so (crp_gather_hits), which tests/test_gpu_parity.py exercises."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_contigs():
    rng = np.random.default_rng(21)
    lens = [5000, 120, 90000, 31, 2500, 2500, 0, 40000, 777]
    a = np.frombuffer(b"ACGTacgtNGGCC", dtype=np.uint8)
    return [b"'" + rng.choice(a, n).tobytes() + b"')," for n in lens]


def _spawn(target, world, *args):
    """Run target(rank, world, port, *args) in `world` processes the way a launcher would."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=target, args=(rank, world, port) + args) for rank in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    return [p.exitcode for p in procs]


def _group(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), CROPSR_RDZV_TIMEOUT="120")
    from cropsr_amd import rendezvous
    return rendezvous.Group.from_env()


def _worker(rank, world, port, out_path, offtarget, max_piece):
    group = _group(rank, world, port)
    from conftest import OracleBackend
    from cropsr_amd import parallel
    from oracle import oracle
    contigs = _make_contigs()
    backend = OracleBackend(oracle)
    got = None
    for _ in range(2):  # twice: nothing of a first exchange may linger
        got = parallel.sharded_scan(backend, contigs, 20, group, max_piece=max_piece, offtarget=offtarget)
    if rank == 0:
        ok = len(got) == len(contigs)
        want_ot = oracle.offtarget_genome(contigs, 20) if offtarget else None
        for k, c in enumerate(contigs):
            want = oracle.scan_score(c)
            if offtarget:
                want["ot_plus"], want["ot_minus"] = want_ot[k]["ot_plus"], want_ot[k]["ot_minus"]
            for col in parallel.COLUMNS + (("ot_plus", "ot_minus") if offtarget else ()):
                w, g = np.ascontiguousarray(want[col]), np.ascontiguousarray(got[k][col])
                ok = ok and g.shape == w.shape and bool((g.view(np.uint8) == w.view(np.uint8)).all())
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    else:
        assert got is None
    group.barrier()
    group.close()


@pytest.mark.parametrize("world,offtarget,max_piece", [(2, False, None), (3, False, 7000), (2, True, 7000), (3, True, None), (8, True, 7000)])
def test_sharded_scan_over_control_sockets(world, offtarget, max_piece, tmp_path, oracle):
    """Every contig's tables (and, with the seed scan, every hit's genome-wide off-target counts) come
    out of the sharded flow exactly as from one scan of the whole list -- with contigs cut into pieces
    of 7000 characters (hits in a neighbour's halo must not be counted as sites twice) or dealt whole."""
    out = str(tmp_path / "result.txt")
    assert _spawn(_worker, world, out, offtarget, max_piece) == [0] * world
    assert open(out).read() == "ok"


def _failing_worker(rank, world, port, out_dir):
    group = _group(rank, world, port)
    from conftest import OracleBackend
    from cropsr_amd import parallel, rendezvous
    from oracle import oracle

    class Broken(OracleBackend):
        def scan_resident(self, texts, l, offtarget=False):
            if group.rank == 1:
                raise ValueError("this rank's share does not fit")
            return OracleBackend.scan_resident(self, texts, l)

    try:
        parallel.sharded_scan(Broken(oracle), _make_contigs(), 20, group)
        msg = "no error"
    except rendezvous.RankError as e:
        msg = str(e)
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write(msg)
    group.close()


def _failing_annotate_worker(rank, world, port, out_dir):
    group = _group(rank, world, port)
    from conftest import OracleBackend, OracleResident
    from cropsr_amd import annotate, parallel, rendezvous
    from oracle import oracle

    class BrokenResident(OracleResident):
        def annotate(self, request):
            if group.rank == 2:
                raise MemoryError("no room for this rank's track")
            return OracleResident.annotate(self, request)

    class Backend(OracleBackend):
        def scan_resident(self, texts, l, offtarget=False):
            return BrokenResident(self.orc, texts, l)

    gff = os.path.join(out_dir, "a.gff")
    if rank == 0:
        with open(gff + ".tmp", "w") as f:
            f.write("##gff-version 3\nc0\tsrc\tgene\t10\t900\t.\t+\t.\tID=g1\n")
        os.replace(gff + ".tmp", gff)
    group.barrier()
    contigs = _make_contigs()
    ann = annotate.Annotation(gff)
    req = annotate.Request(ann, ["c%d" % k for k in range(len(contigs))], 1)
    try:
        parallel.sharded_scan(Backend(oracle), contigs, 20, group, annotation=req)
        msg = "no error"
    except rendezvous.RankError as e:
        msg = str(e)
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write(msg)
    ann.close()
    group.close()


def test_a_rank_that_fails_in_the_annotation_join_fails_every_rank(tmp_path):
    """between the scan's agreement and the gatherv every rank joins ITS tables with the annotation; a rank
    that fails there (here: rank 2 of 3) must not raise alone -- its peers would wait in the exchange.  Every rank raises
    the same RankError."""
    assert _spawn(_failing_annotate_worker, 3, str(tmp_path)) == [0, 0, 0]
    msgs = [(tmp_path / ("rank%d.txt" % r)).read_text() for r in range(3)]
    assert msgs[0] == msgs[1] == msgs[2] and "rank 2: annotation join: MemoryError: no room for this rank's track" in msgs[0]


def test_one_failing_rank_fails_every_rank_with_the_same_message(tmp_path):
    """A rank that cannot scan its share reports it BEFORE the exchange: no rank is left waiting in a
    collective; all of them raise the same RankError."""
    assert _spawn(_failing_worker, 3, str(tmp_path)) == [0, 0, 0]
    msgs = [(tmp_path / ("rank%d.txt" % r)).read_text() for r in range(3)]
    assert msgs[0] == msgs[1] == msgs[2] and "rank 1: ValueError: this rank's share does not fit" in msgs[0]


def test_rendezvous_collectives(tmp_path):
    assert _spawn(_rdzv_worker, 3, str(tmp_path)) == [0, 0, 0]
    for r in range(3):
        assert (tmp_path / ("r%d" % r)).read_text() == "ok"


def _rdzv_worker(rank, world, port, out_dir):
    g = _group(rank, world, port)
    ok = g.all_gather({"r": rank}) == [{"r": r} for r in range(world)]
    ok = ok and g.bcast(b"id" if rank == 0 else None) == b"id"
    ok = ok and g.allreduce([rank, 2.0 * rank]) == [3.0, 6.0] and g.allreduce([rank], "max") == [2.0]
    g.barrier()
    if rank == 0:
        for r in (1, 2):
            a = g.recv_array(r)
            ok = ok and a.dtype == np.float64 and a.shape == (r, 3) and (a == r).all()
            ok = ok and g.recv_array(r).size == 0
    else:
        g.send_array(np.full((rank, 3), float(rank)))
        g.send_array(np.empty(0, dtype=np.uint32))
    try:
        g.check("boom" if rank == 2 else None)
        ok = False
    except Exception as e:
        ok = ok and "rank 2: boom" in str(e)
    with open(os.path.join(out_dir, "r%d" % rank), "w") as f:
        f.write("ok" if ok else "bad")
    g.close()


def test_many_tiny_contigs_stitch_in_linear_time(oracle):
    """the root groups pieces per contig in one pass (50 000 contigs used to take
    n_contigs x n_pieces steps)."""
    import time
    from cropsr_amd import parallel
    n = 50000
    lens = [40] * n
    pieces = parallel.cut_contigs(lens, 8)
    assert len(pieces) == n
    t0 = time.time()
    q, grouped = 0, 0
    for k in range(n):
        while q < len(pieces) and pieces[q][0] == k:
            q += 1
            grouped += 1
    assert grouped == n and time.time() - t0 < 2.0


@pytest.mark.parametrize("world", [2, 3, 8])
def test_strong_plan_of_the_bench_genome_and_stitching(world, oracle):
    """parallel.strong_plan -- what bench.py's strong-scaling block and the CLI's sharded scan deal out for ONE genome
    (BASELINE.json configs[3], [4]): on the switchgrass-like contig lengths every character is owned exactly once, no
    piece exceeds a rank's fair share, the ranks' shares differ by a few per cent; and on a small genome the oracle's
    tables of every rank's pieces (scanned with their halos), stitched, are the whole contigs' tables."""
    import bench_workload as bw
    from cropsr_amd import parallel
    lengths = [s.length + 4 for s in bw.switchgrass_like().specs]
    plan = parallel.strong_plan(lengths, world)
    total = sum(lengths)
    covered = {}
    for k, a, b in plan["pieces"]:
        assert 0 <= a < b <= lengths[k]
        covered.setdefault(k, []).append((a, b))
    assert len(plan["pieces"]) <= len(lengths) + world - 1 and plan["owner"] == sorted(plan["owner"])
    for k, n in enumerate(lengths):
        spans = sorted(covered[k])
        assert spans[0][0] == 0 and spans[-1][1] == n and all(x[1] == y[0] for x, y in zip(spans, spans[1:]))
    assert sum(plan["bases"]) == total and sorted(q for qs in plan["by_rank"] for q in qs) == list(range(len(plan["pieces"])))
    assert max(plan["bases"]) <= total / world + 4096 and min(plan["bases"]) >= total / world - 2 * 4096 * world, plan["bases"]
    for other in (bw.sorghum_like(), bw.tair10_like(), bw.ecoli_like()):
        ls = [s.length + 4 for s in other.specs]
        shares = parallel.strong_plan(ls, world)["bases"]
        assert sum(shares) == sum(ls) and max(shares) <= sum(ls) / world + 4096, (other.name, shares)
    # any list of lengths (empty contigs, fewer characters than ranks, one giant): every character owned once, in order
    rng = np.random.default_rng(100 + world)
    for trial in range(200):
        ls = [int(x) for x in rng.choice([0, 1, 5, 63, 4096, 5000, 20000, 300000], size=int(rng.integers(0, 12)))]
        pieces, owner = parallel.split_evenly(ls, world, min_piece=int(rng.choice([1, 64, 4096])))
        assert owner == sorted(owner) and all(0 <= o < world for o in owner) and len(pieces) == len(owner)
        seen = {}
        for k, a, b in pieces:
            assert a == seen.get(k, 0) and a <= b <= ls[k]
            seen[k] = b
        assert [seen.get(k, None) for k in range(len(ls))] == ls and [p[0] for p in pieces] == sorted(p[0] for p in pieces)
    # a small genome through the same plan, the oracle as hit provider
    rng = np.random.default_rng(world)
    strings = [b"'" + rng.choice(np.frombuffer(b"ACGTacgtNGGCC", np.uint8), n).tobytes() + b"')," for n in (30000, 9000, 700, 64, 12000)]
    plan = parallel.strong_plan([len(s) for s in strings], world)
    per_piece = {}
    for r in range(world):
        for q in plan["by_rank"][r]:
            k, a, b = plan["pieces"][q]
            view, shift = parallel.piece_view(strings[k], a, b)
            per_piece[q] = (shift, oracle.scan_score(bytes(view), 20))
    assert any(sum(1 for p in plan["pieces"] if p[0] == k) > 1 for k in range(len(strings)))
    q = 0
    for k, s in enumerate(strings):
        parts = []
        while q < len(plan["pieces"]) and plan["pieces"][q][0] == k:
            _, a, b = plan["pieces"][q]
            parts.append((a, b, per_piece[q][0], per_piece[q][1]))
            q += 1
        got, want = parallel.stitch_pieces(parts), oracle.scan_score(s, 20)
        for key in ("pos_plus", "score_plus", "pos_minus", "score_minus"):
            assert (np.asarray(got[key]).view(np.uint8) == np.asarray(want[key]).view(np.uint8)).all(), (k, key)


def test_partition_is_balanced_and_deterministic():
    from cropsr_amd import parallel
    lens = [81, 70, 66, 64, 62, 61, 59, 58, 57, 55] + [1] * 857
    a = parallel.partition_contigs(lens, 8)
    assert a == parallel.partition_contigs(lens, 8)
    load = [sum(n for n, o in zip(lens, a) if o == r) for r in range(8)]
    assert max(load) - min(load) <= max(lens)
    assert parallel.partition_contigs([5, 4, 3], 1) == [0, 0, 0]
    assert sorted(set(parallel.partition_contigs([1] * 16, 4))) == [0, 1, 2, 3]


@pytest.mark.parametrize("guide_len", [20, 25, 50, 5])
def test_cut_contigs_give_the_same_hits(oracle, guide_len):
    """parallel.cut_contigs / piece_view / stitch_pieces: a contig scanned in pieces (each with its
    halo, hits owned by match index) gives exactly the tables of the contig scanned whole -- at
    the true ends too, where the reference's decoration and bounds tests live."""
    from cropsr_amd import parallel
    rng = np.random.default_rng(guide_len)
    a = np.frombuffer(b"ACGTacgtNGGCC", dtype=np.uint8)
    contigs = [b"'" + rng.choice(a, n).tobytes() + b"')," for n in (50000, 300, 9000, 0, 4097)]
    pieces = parallel.cut_contigs([len(c) for c in contigs], 4, max_piece=4096)
    assert sum(e - s for _, s, e in pieces) == sum(len(c) for c in contigs)
    assert max(e - s for _, s, e in pieces) <= 4096 and len([p for p in pieces if p[0] == 0]) == 13
    for k, c in enumerate(contigs):
        parts = []
        for kk, start, end in pieces:
            if kk != k:
                continue
            text, shift = parallel.piece_view(c, start, end)
            parts.append((start, end, shift, oracle.scan_score(text, guide_len)))
        got = parallel.stitch_pieces(parts)
        want = oracle.scan_score(c, guide_len)
        for key in want:
            assert got[key].shape == want[key].shape, (k, key)
            assert (np.ascontiguousarray(got[key]).view(np.uint8) == np.ascontiguousarray(want[key]).view(np.uint8)).all(), (k, key)
    # default piece size: the fair share of one rank
    auto = parallel.cut_contigs([1000000, 10, 20], 4)
    assert [p for p in auto if p[0] == 0] == [(0, 0, 250000), (0, 250000, 500000), (0, 500000, 750000), (0, 750000, 1000000)]
    assert [p for p in auto if p[0] != 0] == [(1, 0, 10), (2, 0, 20)]


def _cli_worker(rank, world, port, probe, max_piece, out_dir, extra):
    group = _group(rank, world, port)
    os.environ["CROPSR_DIST_MAX_PIECE"] = str(max_piece)
    import io
    import json
    from conftest import GOLDEN, OracleBackend
    from cropsr_amd import cli
    from oracle import oracle
    oracle.lib()
    seed = json.load(open(os.path.join(GOLDEN, "manifest.json")))["seed"]
    work = os.path.join(out_dir, "rank%d" % rank)
    os.makedirs(work)
    os.chdir(work)
    args = cli.build_parser().parse_args(["-f", os.path.join(GOLDEN, "probe_%s.fa" % probe), "-g", os.path.join(GOLDEN, "sample_head.gff"),
                                          "-o", os.path.join(work, "out.csv"), "--cas9", "--seed", str(seed)] + list(extra))
    buf = io.StringIO()
    cli.run(args, backend=OracleBackend(oracle), out=buf, group=group)
    with open(os.path.join(work, "stdout.txt"), "w") as f:
        f.write(buf.getvalue())
    group.close()


@pytest.mark.parametrize("probe,l,world,max_piece", [("mixed", 100, 2, 150), ("multi", -3, 3, 60), ("rightend", 35, 2, 40)])
def test_cli_multi_process_other_guide_lengths(probe, l, world, max_piece, manifest, tmp_path):
    """-l beyond the engine's range through the sharded flow: every rank scans its pieces with the clamped length, the
    literal keep-filter runs on rank 0 on the STITCHED per-contig tables (whole-contig coordinates and lengths), and the
    bytes are the real reference's for that -l."""
    from conftest import read_golden_csv
    assert _spawn(_cli_worker, world, probe, max_piece, str(tmp_path), ("-l", str(l))) == [0] * world
    with open(tmp_path / "rank0" / "out.csv", "rb") as f:
        assert f.read() == read_golden_csv(probe, l)
    assert (tmp_path / "rank0" / "stdout.txt").read_text() == manifest["cases"]["%s.l%d" % (probe, l)]["stdout"]


@pytest.mark.parametrize("probe,world,max_piece", [("multi", 2, 0), ("mixed", 3, 100), ("tiny", 2, 40)])
def test_cli_multi_process_equals_reference(probe, world, max_piece, manifest, tmp_path):
    """python -m cropsr_amd launched as `world` processes (what torch.distributed.run does): contigs
    are cut and dealt to the ranks, every rank scans its share, the tables are gathered to rank 0,
    and rank 0 alone writes -- the reference's bytes and stdout, as on one GPU.  Here with the control
    sockets as transport and the oracle as hit provider; max_piece forces cuts inside contigs (pieces
    of 40 / 100 characters)."""
    from conftest import read_golden_csv
    assert _spawn(_cli_worker, world, probe, max_piece, str(tmp_path), ()) == [0] * world
    with open(tmp_path / "rank0" / "out.csv", "rb") as f:
        assert f.read() == read_golden_csv(probe)
    assert (tmp_path / "rank0" / "stdout.txt").read_text() == manifest["cases"][probe]["stdout"]
    for r in range(1, world):  # the other ranks wrote nothing and printed nothing
        assert sorted(os.listdir(tmp_path / ("rank%d" % r))) == ["stdout.txt"]
        assert (tmp_path / ("rank%d" % r) / "stdout.txt").read_text() == ""


def test_cli_multi_process_offtarget_equals_single_process(tmp_path, monkeypatch, oracle, manifest):
    """--offtarget through the sharded flow (pieces of 100 characters over 3 ranks) writes the same
    bytes as one process."""
    from conftest import GOLDEN, OracleBackend, run_cli
    assert _spawn(_cli_worker, 3, "mixed", 100, str(tmp_path), ("--offtarget",)) == [0, 0, 0]
    single = tmp_path / "single"
    single.mkdir()
    want, _ = run_cli(single, monkeypatch, os.path.join(GOLDEN, "probe_mixed.fa"), OracleBackend(oracle), manifest["seed"],
                      extra=("--offtarget",))
    with open(tmp_path / "rank0" / "out.csv", "rb") as f:
        assert f.read() == want
    assert b"offtarget_seed_mm3\r\n" in want.split(b"\r\n", 1)[0] + b"\r\n"


def test_cli_multi_process_annotate_equals_single_process_and_brute_force(tmp_path, monkeypatch, oracle, manifest):
    """--annotate through the sharded flow (pieces of 100 characters with their halos over 3 ranks; every rank joins its
    own pieces with the track of THOSE pieces and the label-set ids travel with the tables): the bytes of one process,
    and the `features` column of every row equals the brute-force loop over the GFF (oracle/annotate_oracle.py)."""
    import csv
    import io
    from conftest import GOLDEN, OracleBackend, run_cli
    from oracle import annotate_oracle
    gff = tmp_path / "mixed.gff"
    gff.write_text("##gff-version 3\n"
                   "mix\tsrc\tgene\t40\t410\t.\t+\t.\tID=g1;Name=L1\n"      # spans four cuts
                   "mix\tsrc\tCDS\t95\t105\t.\t+\t0\tID=g1.cds1\n"        # straddles the first cut
                   "mix\tsrc\tCDS\t200\t200\t.\t+\t0\tID=g1.cds2\n"       # one base, on a cut
                   "mix\tsrc\tgene\t400\t1123\t.\t-\t.\tID=g2\n"
                   "mix\tsrc\tCDS\t1100\t2000\t.\t-\t0\tParent=g2\n"      # runs past the contig's end
                   "tail\tsrc\tgene\t1\t60\t.\t+\t.\tID=t1\n")
    info = tmp_path / "info.txt"
    info.write_text("1\tL1\tx\ty\t\t\t\t\t\t\tAT1G1.1\tsym\ta defline\n")
    extra = ("-g", str(gff), "-p", str(info), "--annotate")
    assert _spawn(_cli_worker, 3, "mixed", 100, str(tmp_path), extra) == [0, 0, 0]
    single = tmp_path / "single"
    single.mkdir()
    want, _ = run_cli(single, monkeypatch, os.path.join(GOLDEN, "probe_mixed.fa"), OracleBackend(oracle), manifest["seed"], extra=extra)
    with open(tmp_path / "rank0" / "out.csv", "rb") as f:
        got = f.read()
    assert got == want
    rows = list(csv.reader(io.StringIO(got.decode("latin-1"), newline="")))[1:]
    brute = annotate_oracle.features_of_rows([[c if i != 10 else "" for i, c in enumerate(r)] for r in rows],
                                             lambda chrom: chrom.strip("[(',"), 1, str(gff), annotate_oracle.parse_info(str(info)))
    assert [r[10] for r in rows if len(r) == 12] == [w for r, w in zip(rows, brute) if len(r) == 12]
    labels = set(r[10] for r in rows if len(r) == 12)
    assert "gene:g1|AT1G1.1|a defline;CDS:g1.cds1" in labels and "gene:g1|AT1G1.1|a defline;gene:g2" in labels
    assert "gene:g2;CDS:g2" in labels and "gene:t1" in labels and "" in labels


def _dying_worker(rank, world, port, out_dir):
    group = _group(rank, world, port)
    if rank == 1:
        os._exit(7)  # dies without a word, its peers sitting in a collective
    try:
        group.all_gather("waiting for rank 1 forever")
        msg = "returned"
    except Exception as e:
        msg = "raised %s" % type(e).__name__
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write(msg)


def _aborting_worker(rank, world, port, out_dir):
    group = _group(rank, world, port)
    if rank == 2:
        group.abort("out of memory on this GPU")
    import time
    time.sleep(30)  # stands for a blocking RCCL call: only the watchdog thread can end it
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("survived")


def test_a_dead_or_aborting_rank_takes_every_rank_down_quickly(tmp_path):
    """No hang: when a rank dies, or calls Group.abort, every other rank exits non-zero within seconds --
    also out of a call that never returns (a blocked collective)."""
    import time
    t0 = time.time()
    codes = _spawn(_dying_worker, 3, str(tmp_path))
    assert codes[1] == 7, codes
    for r in (0, 2):  # either the watchdog ended the process (3) or the collective raised; never a normal return
        f = tmp_path / ("rank%d.txt" % r)
        assert codes[r] == 3 or (codes[r] == 0 and f.read_text().startswith("raised")), (r, codes)
    t1 = time.time()
    codes = _spawn(_aborting_worker, 3, str(tmp_path))
    assert codes == [3, 3, 3], codes
    assert time.time() - t0 < 25 and time.time() - t1 < 15
    assert not any(f.startswith("rank") and open(tmp_path / f).read() == "survived" for f in os.listdir(tmp_path))


def test_rendezvous_refuses_strangers(tmp_path):
    """Rank 0 admits only clients that present the run's token, survives garbage on its port, and never builds
    anything but plain data from what it reads off a socket."""
    import json
    import pickle
    import struct
    import threading
    import time as _time
    from cropsr_amd import rendezvous as rz

    with pytest.raises(pickle.UnpicklingError):
        rz._loads(pickle.dumps(os.system))  # a global that is not on the list
    a = np.arange(5, dtype=np.uint64)
    back = rz._loads(pickle.dumps((a, {"k": [1, 2.5, None, b"x"]})))
    assert np.array_equal(back[0], a) and back[1] == {"k": [1, 2.5, None, b"x"]}

    path = str(tmp_path / "rdzv.json")
    groups = {}

    def hub():
        groups[0] = rz.Group(0, 2, 0, None, path)

    t = threading.Thread(target=hub, daemon=True)
    t.start()
    for _ in range(200):
        if os.path.exists(path):
            break
        _time.sleep(0.02)
    info = json.load(open(path))
    assert oct(os.stat(path).st_mode & 0o777) == "0o600" and len(info["token"]) == 32
    # garbage, then a well-formed hello with the wrong token: both are dropped, the hub keeps listening
    s = socket.create_connection((info["host"], info["port"]))
    s.sendall(struct.pack("<Q", 5) + b"hello")
    s.close()
    s = socket.create_connection((info["host"], info["port"]))
    blob = pickle.dumps({"rank": 1, "world": 2, "token": "0" * 32})
    s.sendall(struct.pack("<Q", len(blob)) + blob)
    # hellos whose fields have the wrong TYPES -- without the token (a rank of 'x' used to raise TypeError out
    # of the accept loop and kill rank 0's rendezvous) and with it: judged, dropped, nothing raised
    for hello in ({"world": 2, "rank": "x"}, {"world": 2, "rank": [1]}, {"world": "2", "rank": 1, "token": info["token"]},
                  {"world": 2, "rank": True, "token": info["token"]}, {"world": 2, "rank": 1.0, "token": info["token"]},
                  {"world": 2, "rank": 1, "token": None}, ["rank", 1], None, {"world": 2, "rank": 7, "token": info["token"]}):
        c = socket.create_connection((info["host"], info["port"]))
        blob = pickle.dumps(hello)
        c.sendall(struct.pack("<Q", len(blob)) + blob)
        c.close()
    _time.sleep(0.3)
    assert t.is_alive() and 0 not in groups  # still waiting for the real rank 1
    groups[1] = rz.Group(1, 2, 1, None, path)
    t.join(10)
    assert 0 in groups
    res = {}
    th = threading.Thread(target=lambda: res.setdefault(0, groups[0].all_gather("a")), daemon=True)
    th.start()
    assert groups[1].all_gather("b") == ["a", "b"]
    th.join(10)
    assert res[0] == ["a", "b"]
    tc = threading.Thread(target=groups[0].close, daemon=True)
    tc.start()
    groups[1].close()
    tc.join(10)
    s.close()
