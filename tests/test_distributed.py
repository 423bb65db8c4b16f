"""The N > 1 path on CPU: contig partitioning + the gatherv of per-rank hit tables
(cropsr_amd.parallel) with torch.distributed's gloo backend, world_size 2 and 3.
Per-rank tables come from the oracle here (no GPU in this suite); on the GPU the
same code moves the HIP tables over RCCL."""
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


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from cropsr_amd import parallel
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    contigs = _make_contigs()
    owner = parallel.partition_contigs([len(c) for c in contigs], world)
    mine = [k for k, o in enumerate(owner) if o == rank]
    # a rank-local "arena": contigs at 64-aligned offsets separated like the device arena
    layout, off = [], 64
    cols = {c: [] for c in parallel.COLUMNS}
    for k in mine:
        h = oracle.scan_score(contigs[k])
        layout.append((k, off, len(contigs[k])))
        cols["pos_plus"].append(h["pos_plus"] + np.uint32(off))
        cols["score_plus"].append(h["score_plus"])
        cols["pos_minus"].append(h["pos_minus"] + np.uint32(off))
        cols["score_minus"].append(h["score_minus"])
        off += ((len(contigs[k]) + 63) // 64 + 1) * 64
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.empty(0, dtype=dt)
    tables = {"pos_plus": torch.from_numpy(cat(cols["pos_plus"], np.uint32).view(np.int32)),
              "score_plus": torch.from_numpy(cat(cols["score_plus"], np.float64)),
              "pos_minus": torch.from_numpy(cat(cols["pos_minus"], np.uint32).view(np.int32)),
              "score_minus": torch.from_numpy(cat(cols["score_minus"], np.float64))}
    layouts = [None] * world
    dist.all_gather_object(layouts, layout)
    gather = parallel.TableGather(dst=0)
    for _ in range(2):  # twice: the receive buffers are reused
        got = gather(tables)
    if rank == 0:
        merged = parallel.merge_gathered(
            [{c: t[c].numpy() for c in parallel.COLUMNS} for t in got], owner, layouts)
        ok = len(merged) == len(contigs)
        for k, c in enumerate(contigs):
            want = oracle.scan_score(c)
            for col in parallel.COLUMNS:
                w = want[col]
                g = merged[k][col]
                ok = ok and g.shape == w.shape and bool((g.view(np.uint8) == w.view(np.uint8)).all())
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gatherv_over_gloo(world, tmp_path, oracle):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert open(out).read() == "ok"


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


def _cli_worker(rank, world, port, probe, max_piece, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), CROPSR_DIST_BACKEND="gloo", CROPSR_DIST_MAX_PIECE=str(max_piece))
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
                                          "-o", os.path.join(work, "out.csv"), "--cas9", "--seed", str(seed)])
    buf = io.StringIO()
    cli.run(args, backend=OracleBackend(oracle), out=buf)
    with open(os.path.join(work, "stdout.txt"), "w") as f:
        f.write(buf.getvalue())


@pytest.mark.parametrize("probe,world,max_piece", [("multi", 2, 0), ("mixed", 3, 100), ("tiny", 2, 40)])
def test_cli_under_torch_distributed_equals_reference(probe, world, max_piece, manifest, tmp_path):
    """python -m cropsr_amd launched as `world` processes (what torch.distributed.run does): contigs
    are cut and dealt to the ranks, every rank scans its share, the tables are gathered to rank 0,
    and rank 0 alone writes -- the reference's bytes and stdout, as on one GPU.  Here with gloo and the
    oracle as hit provider; max_piece forces cuts inside contigs (pieces of 40 / 100 characters)."""
    import torch.multiprocessing as mp
    from conftest import read_golden_csv
    port = _free_port()
    mp.spawn(_cli_worker, args=(world, port, probe, max_piece, str(tmp_path)), nprocs=world, join=True)
    with open(tmp_path / "rank0" / "out.csv", "rb") as f:
        assert f.read() == read_golden_csv(probe)
    assert (tmp_path / "rank0" / "stdout.txt").read_text() == manifest["cases"][probe]["stdout"]
    for r in range(1, world):  # the other ranks wrote nothing and printed nothing
        assert sorted(os.listdir(tmp_path / ("rank%d" % r))) == ["stdout.txt"]
        assert (tmp_path / ("rank%d" % r) / "stdout.txt").read_text() == ""
