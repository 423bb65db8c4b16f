"""--annotate: the opt-in GFF / Phytozome join.  No reference oracle exists (the
reference drops the parsed GFF, CROPSR.py:375, and writes '' into `features`, :466-468): the product's
sweep + binary search is checked against the brute-force restatement in oracle/annotate_oracle.py, and
the default CSV is asserted untouched.  The join itself is host code; the `-m gpu` tests at the end run it on
hit tables that come out of the HIP engine."""
import csv
import gzip
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN, OracleBackend, golden_fasta_path, read_golden_csv, run_cli


@pytest.fixture()
def sample_gff(tmp_path):
    p = tmp_path / "sample_genome.gff"
    with gzip.open(os.path.join(GOLDEN, "sample_genome.gff.gz"), "rb") as f:
        p.write_bytes(f.read())
    return str(p)


def _run(tmp_path, monkeypatch, oracle, manifest, fasta, gff, extra):
    import io as _io
    from cropsr_amd import cli
    monkeypatch.chdir(tmp_path)
    out_csv = str(tmp_path / "out.csv")
    args = cli.build_parser().parse_args(["-f", fasta, "-g", gff, "-o", out_csv, "--cas9", "--seed", str(manifest["seed"])] + list(extra))
    cli.run(args, backend=OracleBackend(oracle), out=_io.StringIO())
    with open(out_csv, "rb") as f:
        return f.read()


def test_contig_names():
    from cropsr_amd import annotate
    assert annotate.contig_name("[('Chr01',") == "Chr01"
    assert annotate.contig_name("('c2',") == "c2"
    assert annotate.contig_name(">chrX") == "chrX"


@pytest.mark.parametrize("writer", ["native", "python"])
def test_sample_genome_features_equal_brute_force(writer, oracle, manifest, tmp_path, monkeypatch, sample_gff):
    """The reference's own sample pair (yeast chr I + its GFF, 100 genes / 97 CDS): every one of the 17 314
    rows gets the labels the brute-force join gives it; every other byte of the CSV is the reference's."""
    from oracle import annotate_oracle
    fa = golden_fasta_path("sample", tmp_path)
    got = _run(tmp_path, monkeypatch, oracle, manifest, fa, sample_gff, ("--annotate", "--csv-writer", writer))
    ref = read_golden_csv("sample")
    a = list(csv.reader(io.StringIO(got.decode("latin-1"), newline="")))
    b = list(csv.reader(io.StringIO(ref.decode("latin-1"), newline="")))
    assert len(a) == len(b) and a[0] == b[0]
    want = annotate_oracle.features_of_rows(b[1:], lambda chrom: chrom.strip("(',"), 1, sample_gff)
    n_annotated = 0
    for ra, rb, w in zip(a[1:], b[1:], want):
        assert ra[:10] == rb[:10] and ra[11:] == rb[11:]
        assert ra[10] == w
        n_annotated += bool(w)
    assert 8000 < n_annotated < 17314  # ~70 % of yeast chr I is genic
    assert any(";" in w for w in want) and any(w.startswith("gene:gene-YAL068C;CDS:cds-NP_009332.1") for w in want)
    # without the flag: the reference's bytes
    d = tmp_path / "plain"
    d.mkdir()
    assert _run(d, monkeypatch, oracle, manifest, fa, sample_gff, ()) == ref


def test_overlapping_features_phytozome_names_and_odd_gff(oracle, manifest, tmp_path, monkeypatch):
    """Nested and overlapping genes, CDS without ID, a seqid that is not in the FASTA, a contig without
    features, unformatted FASTA (dec = 0), Phytozome names (-p), and labels that need csv quoting."""
    from oracle import annotate_oracle
    from cropsr_amd import annotate
    rng = np.random.default_rng(12)
    seqs = {"c1": rng.choice(list("ACGT"), 3000), "c2": rng.choice(list("ACGTacgtN"), 1200), "c3": rng.choice(list("ACGT"), 500)}
    fa = tmp_path / "g.fa"
    with open(fa, "w") as f:
        for k, v in seqs.items():
            body = "".join(v)
            f.write(">%s\n" % k + "\n".join(body[i:i + 60] for i in range(0, len(body), 60)) + "\n")
    gff = tmp_path / "g.gff"
    lines = ["##gff-version 3",
             "c1\tphytozome\tgene\t100\t900\t.\t+\t.\tID=G1.v1;Name=G1",
             "c1\tphytozome\tmRNA\t100\t900\t.\t+\t.\tID=G1.1;Parent=G1.v1",
             "c1\tphytozome\tCDS\t150\t400\t.\t+\t0\tID=G1.1.CDS.1;Parent=G1.1",
             "c1\tphytozome\tCDS\t600\t880\t.\t+\t0\tParent=G1.1",
             "c1\tphytozome\tgene\t850\t1500\t.\t-\t.\tID=G2.v1;Name=G2",
             "c1\tphytozome\tgene\t300\t350\t.\t-\t.\tName=nested,with \"quote\"",
             "c1\tphytozome\tCDS\t2990\t3000\t.\t-\t0\tID=edge",
             "c2\tphytozome\tgene\t1\t1200\t.\t+\t.\tID=whole",
             "nowhere\tphytozome\tgene\t1\t100\t.\t+\t.\tID=ghost",
             "c1\tphytozome\tgene\tx\t10\t.\t+\t.\tID=badcoords"]
    gff.write_text("\n".join(lines) + "\n")
    info = tmp_path / "annotation_info.txt"
    info.write_text("#pacId\tlocusName\ttranscriptName\tpeptideName\tPfam\tPanther\tKOG\tKEGG/ec\tKO\tGO\tBest-hit-arabi-name\tarabi-symbol\tarabi-defline\n"
                    "1\tG1\tG1.1\tG1.1.p\t\t\t\t\t\t\tAT1G01010.1\tNAC001\tNAC domain containing protein 1\n"
                    "2\tG2\tG2.1\tG2.1.p\t\t\t\t\t\t\t\t\t\n")
    got = _run(tmp_path, monkeypatch, oracle, manifest, str(fa), str(gff), ("--annotate", "-p", str(info), "--each-contig-once"))
    plain_dir = tmp_path / "plain"
    plain_dir.mkdir()
    plain = _run(plain_dir, monkeypatch, oracle, manifest, str(fa), str(gff), ("--each-contig-once",))
    a = list(csv.reader(io.StringIO(got.decode("latin-1"), newline="")))
    b = list(csv.reader(io.StringIO(plain.decode("latin-1"), newline="")))
    pinfo = annotate.parse_annotation_info(str(info))
    assert pinfo == {"G1": ("AT1G01010.1", "NAC domain containing protein 1"), "G2": ("", "")}
    want = annotate_oracle.features_of_rows(b[1:], lambda chrom: chrom.strip("(',"), 1, str(gff), pinfo)
    assert len(a) == len(b)
    for ra, rb, w in zip(a[1:], b[1:], want):
        assert [x for i, x in enumerate(ra) if i != 10 or len(ra) != 12] == [x for i, x in enumerate(rb) if i != 10 or len(rb) != 12]
        if len(ra) == 12:
            assert ra[10] == w
    labels = set(w for w in want if w)
    assert "gene:G1.v1|AT1G01010.1|NAC domain containing protein 1;CDS:G1.1.CDS.1" in labels
    assert any("gene:nested,with \"quote\"" in w for w in labels)      # needs quoting in the CSV: round-trips
    assert any(w.startswith("gene:G1.v1|") and "gene:G2.v1" in w for w in labels)  # overlap 850..900
    assert "CDS:G1.1" in " ".join(labels)                                  # CDS without ID: its Parent
    assert all("ghost" not in w and "badcoords" not in w for w in labels)
    c3_rows = [ra for ra in a[1:] if "c3" in ra[4]]
    assert c3_rows and all(r[10] == "" for r in c3_rows if len(r) == 12)
    # two-line FASTA without a final newline is scanned unformatted: coordinates are 0-based there (dec = 0)
    fa2 = tmp_path / "two.fa"
    fa2.write_text(">c1\n" + "".join(seqs["c1"]))
    d2 = tmp_path / "two"
    d2.mkdir()
    got2 = _run(d2, monkeypatch, oracle, manifest, str(fa2), str(gff), ("--annotate",))
    rows2 = list(csv.reader(io.StringIO(got2.decode("latin-1"), newline="")))[1:]
    want2 = annotate_oracle.features_of_rows([[c if i != 10 else "" for i, c in enumerate(r)] for r in rows2],
                                             lambda chrom: chrom, 0, str(gff))
    assert [r[10] for r in rows2 if len(r) == 12] == [w for r, w in zip(rows2, want2) if len(r) == 12]
    assert any(w for w in want2)


def test_random_interval_sets_against_brute_force():
    """Seeded fuzz of the sweep + binary search: random gene / CDS intervals (nested, abutting, duplicated, single-base,
    out of range), random hit positions on both strands, guide lengths 17..25, dec 0 and 1 -- against a direct loop."""
    from cropsr_amd import annotate
    import tempfile
    rng = np.random.default_rng(99)
    for trial in range(40):
        n = int(rng.integers(200, 5000))
        feats = []
        for k in range(int(rng.integers(0, 40))):
            a = int(rng.integers(1, n + 50))
            b = a + int(rng.choice([0, 1, 5, 50, 500, n]))
            t = "gene" if rng.random() < 0.5 else "CDS"
            feats.append(("c", t, a, b, "ID=f%d" % (k if rng.random() < 0.8 else 0)))
        with tempfile.NamedTemporaryFile("w", suffix=".gff", delete=False) as f:
            f.write("##gff-version 3\n")
            for sid, t, a, b, attrs in feats:
                f.write("%s\tsrc\t%s\t%d\t%d\t.\t+\t.\t%s\n" % (sid, t, a, b, attrs))
            f.write("other\tsrc\tgene\t1\t100000\t.\t+\t.\tID=elsewhere\n")
            path = f.name
        try:
            ann = annotate.Annotation(path)
            l = int(rng.integers(17, 26))
            dec = int(rng.integers(0, 2))
            ip = np.sort(rng.choice(np.arange(l + 5, n), size=min(60, n - l - 5), replace=False)).astype(np.uint32)
            jm = np.sort(rng.choice(np.arange(2, n), size=min(60, n - 2), replace=False)).astype(np.uint32)
            strings, idx = ann.for_contig("('c'," if dec else ">c", dict(pos_plus=ip, pos_minus=jm), l, dec, n)
            cuts = [int(i) - 3 for i in ip] + [int(j) for j in jm]
            full = [min(int(i) + 5, n) - (int(i) - l - 5) == 30 for i in ip] + [min(int(j) + 3 + l + 5, n) - (int(j) - 2) == 30 for j in jm]
            for k, (cut, ok) in enumerate(zip(cuts, full)):
                want = []
                if ok:
                    x = cut - dec + 1
                    for sid, t, a, b, attrs in feats:
                        lab = "%s:%s" % (t, attrs[3:])
                        if a <= x <= b and lab not in want:
                            want.append(lab)
                got = "" if idx[k] == annotate.NO_FEATURE else strings[int(idx[k])]
                assert got == ";".join(want), (trial, k, cut, got, want)
        finally:
            os.unlink(path)


# ------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_annotate_sample_genome(manifest, tmp_path, monkeypatch, sample_gff):
    """The reference's sample pair through the PRODUCT path -- EngineBackend, hit tables from libcropsr_hip.so -- with
    --annotate: the `features` column equals the brute-force join (oracle/annotate_oracle.py) for all 17 314 rows and every
    other byte is the reference's; the same run without the flag is md5_libm, the reference's own bytes
    (CROPSR.py:77-95 parses the GFF, :375 drops it, :466-468 write '')."""
    import hashlib
    from oracle import annotate_oracle
    fa = golden_fasta_path("sample", tmp_path)
    got, _ = run_cli(tmp_path, monkeypatch, fa, None, manifest["seed"], extra=("-g", sample_gff, "--annotate"))
    ref = read_golden_csv("sample")
    a = list(csv.reader(io.StringIO(got.decode("latin-1"), newline="")))
    b = list(csv.reader(io.StringIO(ref.decode("latin-1"), newline="")))
    assert len(a) == len(b) == 17315 and a[0] == b[0]
    want = annotate_oracle.features_of_rows(b[1:], lambda chrom: chrom.strip("(',"), 1, sample_gff)
    for ra, rb, w in zip(a[1:], b[1:], want):
        assert ra[:10] == rb[:10] and ra[11:] == rb[11:] and ra[10] == w
    assert sum(bool(w) for w in want) > 8000
    d = tmp_path / "plain"
    d.mkdir()
    plain, _ = run_cli(d, monkeypatch, fa, None, manifest["seed"], extra=("-g", sample_gff))
    assert plain == ref and hashlib.md5(plain).hexdigest() == manifest["cases"]["sample"]["md5_libm"]


@pytest.mark.gpu
def test_gpu_annotate_with_offtarget_two_processes_equal_one(manifest, tmp_path, sample_gff):
    """`--annotate --offtarget` as two processes (contigs cut into 20 kb pieces dealt to two ranks that share the one
    GPU here, host transport) writes the bytes of the one-process run: the join sees the same stitched tables."""
    import subprocess
    import sys
    from conftest import ROOT
    fa = golden_fasta_path("sample", tmp_path)
    common = ["-f", fa, "-g", sample_gff, "--cas9", "--seed", str(manifest["seed"]), "--device", "0", "--annotate", "--offtarget"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CROPSR_LAUNCHED")}
    env["PYTHONPATH"] = ROOT
    one, two = tmp_path / "one.csv", tmp_path / "two.csv"
    p = subprocess.run([sys.executable, "-m", "cropsr_amd", "-o", str(one)] + common, capture_output=True, text=True,
                       timeout=600, cwd=str(tmp_path), env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    q = subprocess.run([sys.executable, "-m", "cropsr_amd", "--gpus", "2", "-o", str(two)] + common, capture_output=True, text=True,
                       timeout=600, cwd=str(tmp_path), env=dict(env, CROPSR_GATHER="host", CROPSR_DIST_MAX_PIECE="20000"))
    assert q.returncode == 0, q.stderr[-2000:]
    assert one.read_bytes() == two.read_bytes()
    rows = list(csv.reader(io.StringIO(one.read_bytes().decode("latin-1"), newline="")))
    assert len(rows) == 17315 and len(rows[0]) == 16 and sum(bool(r[10]) for r in rows[1:] if len(r) == 16) > 8000
