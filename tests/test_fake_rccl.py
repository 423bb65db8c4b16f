"""The N > 1 RCCL branches of the library, executed and protocol-checked on a one-GPU box.

crp_node.cpp (ONE process, ncclCommInitAll, one grouped send/recv over N communicators, the grouped histogram all-reduce)
and crp_comm.cpp (one process per GPU, ncclCommInitRank, all-gather + grouped send/recv) are written against RCCL; RCCL
refuses two ranks on one device, so on the one-GPU boxes those branches used to run with world 1 only.  Here every test
runs its workload in CHILD processes whose LD_LIBRARY_PATH starts with a loop-back double of librccl.so.1
(tests/native/fake_rccl.cpp, tests/fake_rccl.py): the product's own dlopen finds the double, which accepts duplicate
devices, moves the bytes itself and -- unlike the real library -- CHECKS the protocol: a send without its receive, a
receive posted with another byte count, a collective somebody stays out of fail the group with ncclInvalidUsage instead
of hanging.  What the reference does at this point is one loop in one process (CROPSR.py:409); the tables that come back
must be that loop's, i.e. the oracle's, bit for bit.

Also here: every RCCL wait of the node handle is bounded (a bootstrap that never returns, a group whose copies never
complete), by making the double hang on purpose.
"""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

RCCL_NAME = "RCCL (in-library, one process)"


def _write(out, obj):
    with open(out, "w") as f:
        json.dump(obj, f)


# ------------------------------------------------------------------ children (run under the double)
def _child_logical_devices(out, world):
    """test_node.test_node_logical_devices_vs_oracle's genomes and options with CRP_NODE_TRANSPORT=rccl."""
    import fake_rccl
    import test_node as tn
    from cropsr_amd import node as nd
    from oracle import oracle
    rng = np.random.default_rng(100 + world)
    genomes = [
        [300_000, 5, 0, 70_000, 9_000, 123_457, 64, 1, 40_000],
        [1_500_000],
        [2_000, 3_000] + [int(v) for v in rng.integers(1, 6_000, 60)] + [400_000],
        [10, 20, 30],
        [],
    ]
    total = 0
    with nd.Node([0] * world) as node:
        for g, lengths in enumerate(genomes):
            contigs = tn._genome(rng, lengths)
            node.load(contigs)
            for l, kw in ((20, {}), (20, {"pos16": False}), (20, {"pre": True}), (20, {"root": world - 1}), (23, {}),
                          (7, {"pos16": False, "root": world // 2})):
                hits = node.scan(l, **kw)
                total += tn._check_against_oracle(hits, contigs, oracle, l, (world, g, l, kw), pre=kw.get("pre", False))
                st = node.gather_stats()
                assert st["transport"] == RCCL_NAME and st["note"] == "", st
                n_rows = hits.n_plus + hits.n_minus
                if n_rows > 1000:
                    assert st["bytes_to_root"] > 0
    stats = fake_rccl.in_process_stats()
    _write(out, {"hits": total, "stats": stats})


def _child_offtarget_annotation(out, world, gff_path):
    """The two opt-in steps through the node on RCCL: the 64 MiB histogram all-reduce on N communicators in one group, and
    the off-target / label-set columns inside the grouped send/recv."""
    import fake_rccl
    import test_node as tn
    from cropsr_amd import annotate, node as nd
    from oracle import annotate_oracle, oracle
    rng = np.random.default_rng(500 + world)
    lengths = [300_000, 5, 70_000, 123_457, 40_000]
    contigs = tn._genome(rng, lengths)
    block = rng.choice(tn.ALPHA, 40_000).tobytes()
    contigs[0] = contigs[0][:100_000] + block + contigs[0][140_000:]
    contigs[3] = contigs[3][:60_000] + block + contigs[3][100_000:]
    contigs[4] = b"'" + block + b"')]"
    rows = ["##gff-version 3"]
    for k, n in enumerate(lengths):
        for g in range(max(1, n // 20_000)):
            a = 1 + g * 20_000 + int(rng.integers(0, 5_000))
            b = min(n, a + int(rng.integers(300, 12_000)))
            if b > a:
                rows.append("c%d\tsrc\tgene\t%d\t%d\t.\t+\t.\tID=g%d_%d" % (k, a, b, k, g))
    with open(gff_path, "w") as f:
        f.write("\n".join(rows) + "\n")
    ann = annotate.Annotation(gff_path)
    req = annotate.Request(ann, ["c%d" % k for k in range(len(contigs))], 1)
    want_ot = oracle.offtarget_genome(contigs, 20)
    n_feat = 0
    with nd.Node([0] * world) as node:
        node.load(contigs)
        for kw in ({}, {"pos16": False, "root": world - 1}):
            hits = node.scan(20, offtarget=True, annotation=req, **kw)
            assert node.gather_stats()["transport"] == RCCL_NAME
            for k, c in enumerate(contigs):
                got = hits.contig(k)
                want = oracle.scan_score(c, 20)
                for key in ("pos_plus", "score_plus", "pos_minus", "score_minus"):
                    assert (tn.bits(got[key]) == tn.bits(want[key])).all(), (world, k, key)
                assert (got["ot_plus"] == want_ot[k]["ot_plus"]).all() and (got["ot_minus"] == want_ot[k]["ot_minus"]).all(), (world, k)
                fp, fm = annotate_oracle.host_join(ann, "c%d" % k, 0, 1, got, 20, len(c))
                assert (got["feat_plus"] == fp).all() and (got["feat_minus"] == fm).all(), (world, k)
                n_feat += int((fp != annotate.NO_FEATURE).sum() + (fm != annotate.NO_FEATURE).sum())
    ann.close()
    _write(out, {"n_feat": n_feat, "stats": fake_rccl.in_process_stats()})


def _child_tair10(out):
    """The TAIR10-like genome on four logical devices over RCCL == the N = 1 tables, SHA-256 per contig."""
    import bench_workload as bw
    import fake_rccl
    from cropsr_amd import Engine, node as nd

    def digest(h):
        d = hashlib.sha256()
        for key in ("pos_plus", "score_plus", "pos_minus", "score_minus"):
            d.update(np.ascontiguousarray(h[key]).tobytes())
        return d.hexdigest()

    wl = bw.tair10_like()
    strings = [wl.contig_string(k) for k in range(len(wl.specs))]
    with Engine(0) as eng:
        arena = eng.arena(strings)
        one = arena.scan_score(20, want_pre=False)
        n1 = [digest(one.contig(k)) for k in range(len(strings))]
        n1_hits = one.n_plus + one.n_minus
        arena.close()
    with nd.Node([0, 0, 0, 0]) as node:
        node.load(strings)
        got = {}
        for pos16 in (True, False):
            hits = node.scan(20, pos16=pos16)
            got[pos16] = [digest(hits.contig(k)) for k in range(len(strings))]
            st = node.gather_stats()
            assert st["transport"] == RCCL_NAME, st
    assert got[True] == n1 and got[False] == n1
    _write(out, {"hits": n1_hits, "bytes_to_root": st["bytes_to_root"], "stats": fake_rccl.in_process_stats()})


def _child_fuzz(out, trials, seed):
    """test_node.test_node_randomised_genomes_vs_oracle over RCCL (no host gather: that one crosses no link)."""
    import fake_rccl
    import test_node as tn
    from cropsr_amd import node as nd
    from oracle import oracle
    rng = np.random.default_rng(seed)
    alphabets = [b"ACGT", b"ACGTacgtN", b"GGCC", b"ACGTUZuzN')],", b"GGGGGGCCCCCCAT"]
    anchors = [0, 1, 30, 64, 127, 128, 129, 255, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 16384, 65535, 65536, 65537, 131072, 200000]
    nodes = {}
    total = cuts = 0
    try:
        for trial in range(trials):
            world = int(rng.integers(2, 8))
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
            hits = node.scan(l, root=int(rng.integers(0, world)), pre=pre, pos16=bool(rng.random() < 0.7))
            assert node.gather_stats()["transport"] == RCCL_NAME
            total += tn._check_against_oracle(hits, contigs, oracle, l, (trial, world, l, pre), pre=pre)
    finally:
        for node in nodes.values():
            node.close()
    _write(out, {"hits": total, "cuts": cuts, "stats": fake_rccl.in_process_stats()})


def _child_arenas(out, world, words):
    """Several arenas per device (a tiny per-arena limit) with the exchange on RCCL: a device posts one run of sends per
    arena, the root the matching receives, in the same order."""
    import fake_rccl
    import test_node as tn
    from cropsr_amd import node as nd
    from oracle import oracle
    rng = np.random.default_rng(77 + world)
    contigs = tn._genome(rng, [260_000, 5, 0, 70_000, 123_457, 64, 40_000])
    want_ot = oracle.offtarget_genome(contigs, 20)
    total = 0
    with nd.Node([0] * world) as node:
        node.set_option(arena_words=words)
        node.load(contigs)
        n_arenas = [node.n_arenas(k) for k in range(world)]
        assert max(n_arenas) > 1
        for l, kw in ((20, {}), (20, {"pos16": False, "root": world - 1}), (23, {"pre": True})):
            hits = node.scan(l, **kw)
            assert node.gather_stats()["transport"] == RCCL_NAME
            total += tn._check_against_oracle(hits, contigs, oracle, l, (world, words, l, kw), pre=kw.get("pre", False))
        hits = node.scan(20, offtarget=True)
        for k in range(len(contigs)):
            got = hits.contig(k)
            assert (got["ot_plus"] == want_ot[k]["ot_plus"]).all() and (got["ot_minus"] == want_ot[k]["ot_minus"]).all(), k
    _write(out, {"hits": total, "n_arenas": n_arenas, "stats": fake_rccl.in_process_stats()})


def _child_failure(out, world, mode):
    """One failure injected (miscount: the root posts one receive 8 bytes short; hang_init / hang_group: the double never
    returns / never completes): what the calls return, how long they took, what the node says, and -- where the node is
    allowed to go on without RCCL -- that the tables are still the oracle's."""
    import fake_rccl
    import test_node as tn
    from cropsr_amd import _native as nat, node as nd
    from oracle import oracle
    rng = np.random.default_rng(9)
    contigs = tn._genome(rng, [250_000, 3_000, 120_000])
    res = {"mode": mode}
    with nd.Node([0] * world) as node:
        node.load(contigs)
        t0 = time.perf_counter()
        try:
            if mode.endswith("offtarget"):
                hits = node.scan(20, offtarget=True)
            else:
                hits = node.scan(20)
            res["status"] = 0
        except nat.CropsrHipError as e:
            hits = None
            res["status"] = e.status
            res["message"] = str(e)
        res["seconds"] = time.perf_counter() - t0
        res["comm_stuck"] = nd.comm_stuck()
        if hits is not None:
            st = node.gather_stats()
            res["transport"], res["note"] = st["transport"], st["note"]
            res["hits"] = tn._check_against_oracle(hits, contigs, oracle, 20, mode)
            if mode.endswith("offtarget"):
                want_ot = oracle.offtarget_genome(contigs, 20)
                for k in range(len(contigs)):
                    got = hits.contig(k)
                    assert (got["ot_plus"] == want_ot[k]["ot_plus"]).all() and (got["ot_minus"] == want_ot[k]["ot_minus"]).all(), k
            # and the node stays usable: the next scan needs no RCCL any more
            again = node.scan(20, pos16=False)
            tn._check_against_oracle(again, contigs, oracle, 20, mode + " (again)")
            res["transport_again"] = node.gather_stats()["transport"]
    res["stats"] = fake_rccl.in_process_stats()
    _write(out, res)


# ------------------------------------------------------------------ the node handle (one process) on the double
@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 4, 7, 8])
def test_node_logical_devices_on_rccl_double(world):
    """ncclCommInitAll over `world` logical devices, then every gather as ONE group of sends and receives: packed and raw
    positions, the pre-sigmoid column, another root -- the oracle's rows, every send matched by a receive of the same size."""
    import fake_rccl
    with fake_rccl.Session(CRP_NODE_TRANSPORT="rccl") as s:
        p, r = s.run_child("_child_logical_devices", world)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        assert r["hits"] > 100_000
        st = r["stats"]
        assert st["pairs"] > 50 and st["mismatches"] == 0 and st["inits"] == 1 and st["p2p_bytes"] > 1_000_000, st


@pytest.mark.gpu
@pytest.mark.parametrize("world", [3, 4])
def test_node_offtarget_and_annotation_on_rccl_double(world, tmp_path):
    """crp_node_offtarget's all-reduce (N communicators of one process in one group, 64 MiB each) and the two extra columns
    in the grouped exchange: counts == the oracle's genome-wide enumeration, ids == the oracle's numpy join."""
    import fake_rccl
    with fake_rccl.Session(CRP_NODE_TRANSPORT="rccl") as s:
        p, r = s.run_child("_child_offtarget_annotation", world, str(tmp_path / "node.gff"))
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        assert r["n_feat"] > 5_000
        st = r["stats"]
        assert st["collectives"] == 2 and st["pairs"] > 20 and st["mismatches"] == 0, st


@pytest.mark.gpu
@pytest.mark.parametrize("world,words", [(2, 600), (3, 137)])
def test_node_several_arenas_per_device_on_rccl_double(world, words):
    import fake_rccl
    with fake_rccl.Session(CRP_NODE_TRANSPORT="rccl") as s:
        p, r = s.run_child("_child_arenas", world, words)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        n_peer_arenas = sum(r["n_arenas"]) - r["n_arenas"][0]
        assert r["hits"] > 100_000 and r["stats"]["mismatches"] == 0 and r["stats"]["pairs"] > 4 * n_peer_arenas, r


@pytest.mark.gpu
@pytest.mark.slow
def test_node_tair10_like_four_devices_on_rccl_double():
    """the TAIR10-like genome (7.7 M hits, every chromosome straddles a share) on {0, 0, 0, 0} with the
    exchange on RCCL == the N = 1 tables by SHA-256 per contig, packed and raw positions."""
    import fake_rccl
    with fake_rccl.Session(CRP_NODE_TRANSPORT="rccl") as s:
        p, r = s.run_child("_child_tair10", timeout=1200)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        assert r["hits"] > 7_000_000 and r["bytes_to_root"] > 60_000_000 and r["stats"]["mismatches"] == 0 and r["stats"]["pairs"] >= 18


@pytest.mark.gpu
def test_node_randomised_genomes_on_rccl_double():
    from conftest import fuzz_settings
    import fake_rccl
    trials, seed, _ = fuzz_settings(40, 20261006)
    with fake_rccl.Session(CRP_NODE_TRANSPORT="rccl") as s:
        p, r = s.run_child("_child_fuzz", trials, seed, timeout=1800)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        assert r["hits"] > 20000 * trials // 40 and r["cuts"] > trials // 4 and r["stats"]["mismatches"] == 0 and r["stats"]["pairs"] > trials


@pytest.mark.gpu
def test_node_count_mismatch_is_an_error_not_a_hang():
    """The root posts one receive 8 bytes short (CRP_TEST_NODE_MISCOUNT): the double refuses the group, crp_node_gather
    returns CRP_ERR_COMM with the reason -- within seconds, nothing waits for bytes that never come."""
    import fake_rccl
    from cropsr_amd import _native as nat
    with fake_rccl.Session(CRP_NODE_TRANSPORT="rccl", CRP_TEST_NODE_MISCOUNT="1") as s:
        p, r = s.run_child("_child_failure", 3, "miscount", timeout=300)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        assert r["status"] == nat.CRP_ERR_COMM and "bytes" in r["message"] and "invalid usage" in r["message"], r
        assert r["seconds"] < 30 and r["stats"]["mismatches"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["rccl", "try"])
def test_node_bootstrap_that_never_returns_is_bounded(transport):
    """ncclCommInitAll never returns (FAKE_RCCL_HANG=init): the helper thread is given CRP_NODE_COMM_INIT_TIMEOUT_S; asked
    for by name (rccl) the gather fails with CRP_ERR_COMM naming the bootstrap, otherwise (try) the same call goes on as
    device-to-device copies with the oracle's tables; either way within the bound, and crp_node_comm_stuck says a thread was
    left behind."""
    import fake_rccl
    from cropsr_amd import _native as nat
    with fake_rccl.Session(CRP_NODE_TRANSPORT=transport, FAKE_RCCL_HANG="init", FAKE_RCCL_HANG_MAX_S="40",
                           CRP_NODE_COMM_INIT_TIMEOUT_S="2") as s:
        p, r = s.run_child("_child_failure", 3, "hang_init", timeout=300)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        assert r["seconds"] < 15 and r["comm_stuck"] == 1, r
        if transport == "rccl":
            assert r["status"] == nat.CRP_ERR_COMM and "did not return within 2 s" in r["message"], r
        else:
            assert r["status"] == 0 and r["transport"] == "device-to-device copies" and "did not return within 2 s" in r["note"], r
            assert r["hits"] > 30_000 and r["transport_again"] == "device-to-device copies"


@pytest.mark.gpu
@pytest.mark.parametrize("transport,mode", [("rccl", "hang_group"), ("try", "hang_group"), ("try", "hang_group_offtarget")])
def test_node_collective_that_never_completes_is_bounded(transport, mode):
    """The copies of a group never complete (FAKE_RCCL_HANG=group: every stream of the group is held until ncclCommAbort):
    the event the node polls runs into CRP_NODE_COLLECTIVE_TIMEOUT_S, the communicators are aborted, and the call either
    fails with CRP_ERR_COMM naming the stage (rccl) or starts over on the device-to-device transport (try) -- for the gather
    and for the histogram all-reduce of the off-target step alike -- with the oracle's tables."""
    import fake_rccl
    from cropsr_amd import _native as nat
    with fake_rccl.Session(CRP_NODE_TRANSPORT=transport, FAKE_RCCL_HANG="group", FAKE_RCCL_HANG_MAX_S="60",
                           CRP_NODE_COLLECTIVE_TIMEOUT_S="1.5") as s:
        p, r = s.run_child("_child_failure", 4, mode, timeout=300)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        assert r["seconds"] < 30 and r["stats"]["hangs"] >= 1 and r["stats"]["aborts"] >= 4, r
        if transport == "rccl":
            assert r["status"] == nat.CRP_ERR_COMM and "did not complete within 1.5 s" in r["message"] and "send/recv" in r["message"], r
        else:
            stage = "all-reduce" if mode.endswith("offtarget") else "send/recv"
            assert r["status"] == 0 and r["transport"] == "device-to-device copies" and stage in r["note"], r
            assert r["hits"] > 30_000 and r["transport_again"] == "device-to-device copies"


@pytest.mark.gpu
def test_cli_devices_on_a_hung_rccl_exits_like_the_reference(manifest, tmp_path):
    """`python -m cropsr_amd --devices 0,0,0` when RCCL's bootstrap never returns: asked for by name the program ends with a
    message and a non-zero status within the bound (the reference's failure style: any error ends the run, CROPSR.py has no
    recovery); in the default spirit (try) it writes the reference's CSV over device-to-device copies.  Neither run waits for
    the thread that is still inside the bootstrap."""
    import fake_rccl
    from conftest import golden_fasta_path, read_golden_csv
    fa = golden_fasta_path("sample", tmp_path)
    common = ["-f", fa, "-g", os.path.join(GOLDEN, "sample_head.gff"), "--cas9", "--seed", str(manifest["seed"]), "--devices", "0,0,0"]
    for transport in ("rccl", "try"):
        with fake_rccl.Session(CRP_NODE_TRANSPORT=transport, FAKE_RCCL_HANG="init", FAKE_RCCL_HANG_MAX_S="60",
                               CRP_NODE_COMM_INIT_TIMEOUT_S="2") as s:
            d = tmp_path / transport
            d.mkdir()
            out = d / "out.csv"
            t0 = time.perf_counter()
            p = subprocess.run([sys.executable, "-m", "cropsr_amd", "-o", str(out)] + common, capture_output=True, text=True,
                               timeout=300, cwd=str(d), env=s.env())
            took = time.perf_counter() - t0
            assert took < 45, took
            if transport == "rccl":
                assert p.returncode != 0 and "did not return within 2 s" in p.stderr, p.stderr[-2000:]
            else:
                assert p.returncode == 0, p.stderr[-2000:]
                assert out.read_bytes() == read_golden_csv("sample")


# ------------------------------------------------------------------ one process per GPU (crp_comm.cpp) on the double
@pytest.mark.gpu
@pytest.mark.parametrize("nproc,extra", [(2, ()), (3, ("--offtarget", "--annotate")), (2, ("--score-finalize", "host")), (4, ("--offtarget",))])
def test_cli_multi_process_on_rccl_double(nproc, extra, manifest, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node N -m cropsr_amd ...` with every rank on GPU 0 and the exchange on
    RCCL (no CROPSR_GATHER=host): ncclCommInitRank over the mailbox, crp_gather_hits' two all-gather rounds and its grouped
    send/recv with every column (packed positions, scores or pre-sigmoid sums, off-target counts, label-set ids), the 64 MiB
    histogram all-reduce -- the bytes of the one-process run, and the root's communicator saw matched pairs only."""
    import fake_rccl
    gff = tmp_path / "mixed.gff"
    gff.write_text("##gff-version 3\nmix\tsrc\tgene\t40\t410\t.\t+\t.\tID=g1;Name=L1\nmix\tsrc\tCDS\t95\t105\t.\t+\t0\tID=g1.cds1\n"
                   "mix\tsrc\tgene\t400\t1123\t.\t-\t.\tID=g2\ntail\tsrc\tgene\t1\t60\t.\t+\t.\tID=t1\n")
    common = ["-f", os.path.join(GOLDEN, "probe_mixed.fa"), "-g", str(gff), "--cas9", "--seed", str(manifest["seed"]), "--device", "0"] + list(extra)
    with fake_rccl.Session(CROPSR_DIST_MAX_PIECE="100") as s:
        out_csv = tmp_path / "out.csv"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(29551 + nproc), "-m", "cropsr_amd", "-o", str(out_csv)] + common
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(tmp_path), env=s.env())
        assert p.returncode == 0, p.stderr[-4000:]
        assert "host transport" not in p.stderr  # (the communicator was created: nothing fell back to the sockets)
        stats = s.stats()
        assert len(stats) == nproc and sorted(x["rank"] for x in stats) == list(range(nproc)), stats
        root = [x for x in stats if x["rank"] == 0][0]
        assert root["pairs"] >= 2 * (nproc - 1) and root["collectives"] >= 2 and all(x["mismatches"] == 0 for x in stats), stats
        one = tmp_path / "one.csv"
        env1 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CROPSR_LAUNCHED", "CROPSR_GATHER")}
        q = subprocess.run([sys.executable, "-m", "cropsr_amd", "-o", str(one)] + common, capture_output=True, text=True,
                           timeout=600, cwd=str(tmp_path), env=dict(env1, PYTHONPATH=ROOT))
        assert q.returncode == 0, q.stderr[-2000:]
        assert out_csv.read_bytes() == one.read_bytes() and out_csv.stat().st_size > 10_000


@pytest.mark.gpu
def test_multi_process_count_mismatch_takes_the_run_down_quickly(manifest, tmp_path):
    """The last rank sends one row too few of its score column (CRP_TEST_GATHER_FAIL=3): the root's receive sees the wrong byte
    count, crp_gather_hits returns CRP_ERR_COMM there, and the abort channel ends every rank -- seconds, not a hang."""
    import fake_rccl
    common = ["-f", os.path.join(GOLDEN, "probe_mixed.fa"), "-g", os.path.join(GOLDEN, "sample_head.gff"), "--cas9", "--seed",
              str(manifest["seed"]), "--device", "0"]
    with fake_rccl.Session(CROPSR_DIST_MAX_PIECE="100", CRP_TEST_GATHER_FAIL="3", FAKE_RCCL_TIMEOUT_S="30") as s:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29561", "-m", "cropsr_amd", "-o", str(tmp_path / "out.csv")] + common
        t0 = time.perf_counter()
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=s.env())
        assert p.returncode != 0 and time.perf_counter() - t0 < 120
        assert "the receive expects" in p.stderr or "invalid usage" in p.stderr, p.stderr[-3000:]


@pytest.mark.gpu
def test_bench_strong_block_on_rccl_double():
    """bench.py's process-per-GPU line with three ranks sharing GPU 0 and RCCL forced (CROPSR_BENCH_FORCE_RCCL=1): fences and
    sums on ncclAllReduce, the weak gatherv and the strong block's gatherv on crp_gather_hits -- digest_ok, RCCL named."""
    import fake_rccl
    with fake_rccl.Session(CROPSR_BENCH_FORCE_RCCL="1", CRP_NODE_TRANSPORT="try") as s:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
               "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "3", "--share-gpu0", "--scale", "0.02", "--steps", "2",
               "--warmup", "1", "--offtarget-steps", "1"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=s.env())
        assert p.returncode == 0, p.stderr[-4000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 3 and "rccl_error" not in d and d["gatherv_ok"] is True
        assert d["strong"]["digest_ok"] is True and d["strong"]["gatherv_transport"].startswith("RCCL")
        # rank 0's closing node block: a child process drives the same three (logical) devices through the node handle --
        # ncclCommInitAll + one grouped send/recv in ONE process (CRP_NODE_TRANSPORT=try: RCCL although GPU 0 is listed thrice)
        nb = d["single_process_node"]
        assert "error" not in nb and nb["digest_ok"] is True and nb["gatherv"]["transport"].startswith("RCCL"), nb
        stats = s.stats()
        ranks = [x for x in stats if not x["in_process"]]
        node = [x for x in stats if x["in_process"]]
        assert len(ranks) == 3 and all(x["mismatches"] == 0 for x in stats) and max(x["pairs"] for x in ranks) > 4, stats
        assert len(node) == 3 and node[0]["pairs"] > 4, stats
