import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: whole-genome checks that take minutes (still part of -m gpu)")
    # a fresh checkout has no binaries: build them once (hipcc cross-compiles without a GPU)
    so = os.path.join(ROOT, "cropsr_amd", "libcropsr_hip.so")
    orc = os.path.join(ROOT, "oracle", "liborc.so")
    if not (os.path.exists(so) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); compiled on first use."""
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def sample_fasta_text():
    with gzip.open(os.path.join(GOLDEN, "sample_genome.fa.gz"), "rt") as f:
        return f.read()


PROBES = ["multi", "twoline", "spaces", "edges", "mixed", "rightend", "tiny", "dupname", "crlf"]
# probes the real reference was also run on with a non-default guide length: (probe, -l value)
LENGTH_CASES = [("mixed", 23), ("rightend", 17), ("multi", 25), ("tiny", 21), ("edges", 24),
                # beyond the engine's native 1..50 (the reference takes any integer, CROPSR.py:38-40), and the range's ends
                ("mixed", 0), ("multi", -3), ("tiny", -12), ("multi", 51), ("rightend", 64), ("mixed", 100),
                ("rightend", 35), ("rightend", 36), ("mixed", 50), ("tiny", 1)]


VERBOSE_CASES = ["multi", "mixed"]


def normalize_verbose(text):
    """-v output with the run-specific parts masked: file paths and the host's CPU count."""
    import re
    text = re.sub(r"\S*/\S+", "<PATH>", text)
    return re.sub(r"(Number of available CPUs:\s+)\d+", r"\1<N>", text)


def read_golden_csv(name, guide_len=None):
    if guide_len is not None:
        with open(os.path.join(GOLDEN, "probe_%s.l%d.libm.csv" % (name, guide_len)), "rb") as f:
            return f.read()
    if name == "sample":
        with gzip.open(os.path.join(GOLDEN, "sample_libm.csv.gz"), "rb") as f:
            return f.read()
    with open(os.path.join(GOLDEN, "probe_%s.libm.csv" % name), "rb") as f:
        return f.read()


def golden_fasta_path(name, tmp_path):
    if name == "sample":
        p = tmp_path / "sample_genome.fa"
        with gzip.open(os.path.join(GOLDEN, "sample_genome.fa.gz"), "rb") as f:
            p.write_bytes(f.read())
        return str(p)
    return os.path.join(GOLDEN, "probe_%s.fa" % name)


class OracleResident:
    """parallel.sharded_scan's Resident on the CPU oracle: one host "arena" with the texts at
    64-aligned offsets separated like the device arena."""

    def __init__(self, orc, texts, l):
        import numpy as np
        self.orc, self.l = orc, l
        self.texts, self.layout, self.hits = [bytes(t) for t in texts], [], []
        off = 64
        cols = {c: [] for c in ("pos_plus", "score_plus", "pos_minus", "score_minus")}
        for t in self.texts:
            h = orc.scan_score(t, l)
            self.hits.append(h)
            self.layout.append((0, off, len(t)))
            cols["pos_plus"].append(h["pos_plus"] + np.uint32(off))
            cols["pos_minus"].append(h["pos_minus"] + np.uint32(off))
            cols["score_plus"].append(h["score_plus"])
            cols["score_minus"].append(h["score_minus"])
            off += ((len(t) + 63) // 64 + 1) * 64
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.empty(0, dtype=dt)
        self.cols = {"pos_plus": cat(cols["pos_plus"], np.uint32), "score_plus": cat(cols["score_plus"], np.float64),
                     "pos_minus": cat(cols["pos_minus"], np.uint32), "score_minus": cat(cols["score_minus"], np.float64)}

    def offtarget(self, group, own_by_arena):
        """The seed scan by the oracle's histogram + enumeration method, sites of all ranks."""
        import numpy as np
        orc = self.orc
        own = np.asarray(own_by_arena[0] if own_by_arena else [], dtype=np.int64).reshape(-1, 2)
        seeds = {"plus": [], "minus": []}
        for t, h, (_, off, _ln) in zip(self.texts, self.hits, self.layout):
            for strand, minus in (("plus", False), ("minus", True)):
                codes = orc.seed_codes(t, h["pos_" + strand], minus, self.l)
                p = h["pos_" + strand].astype(np.int64) + off
                mine = np.zeros(p.size, dtype=bool)
                for b, e in own:
                    mine |= (p >= b) & (p < e)
                codes[~mine] = orc.NOT_A_SITE  # a neighbour's halo: not this rank's site
                seeds[strand].append(codes)
        cat = lambda xs: np.concatenate(xs) if xs else np.empty(0, np.uint32)
        sp, sm = cat(seeds["plus"]), cat(seeds["minus"])
        hist = orc.offtarget_hist([sp, sm])
        idx = np.flatnonzero(hist)
        total = np.zeros_like(hist)
        for i, v in group.all_gather((idx, hist[idx])):
            np.add.at(total, i, v)
        self.cols["ot_plus"], self.cols["ot_minus"] = orc.offtarget_enum(sp, total), orc.offtarget_enum(sm, total)

    def annotate(self, request):
        """The join by the oracle's numpy statement (oracle/annotate_oracle.host_join), text by text."""
        import numpy as np
        from oracle import annotate_oracle
        fp, fm = [], []
        for k, (t, h) in enumerate(zip(self.texts, self.hits)):
            a, b = annotate_oracle.host_join(request.annotation, request.names[k], request.starts[k], request.dec, h, self.l, len(t))
            fp.append(a)
            fm.append(b)
        cat = lambda xs: np.concatenate(xs) if xs else np.empty(0, np.uint32)
        self.cols["feat_plus"], self.cols["feat_minus"] = cat(fp), cat(fm)

    def gather(self, group, dst, offtarget, features=False):
        from cropsr_amd import parallel
        return parallel.gather_host(group, [self.cols], dst, offtarget, features)

    def release(self):
        pass


class OracleBackend:
    """cli backend built on the CPU oracle, for pinning the host logic without a GPU."""

    def __init__(self, orc, finalize="gpu"):
        self.orc = orc
        self.finalize = finalize

    def _sigmoid(self, pre):
        from cropsr_amd import cli
        return cli.host_sigmoid(pre)

    def scan(self, strings, l, offtarget=False, annotation=None):
        texts = [s.encode("ascii", "replace") if isinstance(s, str) else bytes(s) for s in strings]
        out = [self.orc.scan_score(t, l) for t in texts]
        if annotation is not None:  # in place of the GPU look-up: the oracle's numpy statement of it
            from oracle import annotate_oracle
            for k, (t, h) in enumerate(zip(texts, out)):
                h["feat_plus"], h["feat_minus"] = annotate_oracle.host_join(
                    annotation.annotation, annotation.names[k], annotation.starts[k], annotation.dec, h, l, len(t))
        if self.finalize == "host":
            for h in out:
                h["score_plus"], h["score_minus"] = self._sigmoid(h["pre_plus"]), self._sigmoid(h["pre_minus"])
        if offtarget:
            for h, ot in zip(out, self.orc.offtarget_genome(texts, l)):
                h["ot_plus"], h["ot_minus"] = ot["ot_plus"], ot["ot_minus"]
        return out

    def scan_resident(self, texts, l, offtarget=False):
        return OracleResident(self.orc, texts, l)

    def rescore(self, rows_u8, order):
        pre, score = self.orc.score30_order(rows_u8, order)
        return self._sigmoid(pre) if self.finalize == "host" else score


def oracle_scan_provider(orc):
    return OracleBackend(orc)


def run_cli(tmp_path, monkeypatch, fasta_path, backend, seed, extra=()):
    """Run cropsr_amd.cli in tmp_path; returns (csv bytes, stdout text)."""
    import io
    from cropsr_amd import cli
    monkeypatch.chdir(tmp_path)
    gff = os.path.join(GOLDEN, "sample_head.gff")
    out_csv = str(tmp_path / "out.csv")
    argv = ["-f", fasta_path, "-g", gff, "-o", out_csv, "--cas9", "--seed", str(seed)] + list(extra)
    args = cli.build_parser().parse_args(argv)
    buf = io.StringIO()
    cli.run(args, backend=backend, out=buf)
    with open(out_csv, "rb") as f:
        return f.read(), buf.getvalue()


def fuzz_settings(default_trials, base_seed):
    """(trials, seed, tick) of a seeded fuzz test.  CROPSR_FUZZ_TRIALS: more trials for a soak on the GPU box;
    CROPSR_FUZZ_SEED: an offset to the test's seed (a second soak covers new ground); CROPSR_FUZZ_PROGRESS: a file
    that gets a line every 100 trials (a long soak must keep writing, or the GPU pool takes it for hung)."""
    trials = int(os.environ.get("CROPSR_FUZZ_TRIALS", str(default_trials)))
    seed = base_seed + int(os.environ.get("CROPSR_FUZZ_SEED", "0"))
    path = os.environ.get("CROPSR_FUZZ_PROGRESS")

    def tick(name, trial):
        if path and trial % 100 == 99:
            with open(path, "a") as f:
                f.write("%s: %d trials\n" % (name, trial + 1))
    return trials, seed, tick
