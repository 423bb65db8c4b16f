"""CPU-only tests of the host side: the C-ABI library loads and exports what
include/cropsr_hip.h declares, host packing, the FASTA contig table, the row /
chunk logic.  No GPU compute is attempted here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, OracleBackend, golden_fasta_path, read_golden_csv, run_cli

from cropsr_amd import _native as nat
from cropsr_amd import fasta, rows


# ------------------------------------------------------------------ C ABI
def _declared_functions():
    text = open(os.path.join(ROOT, "include", "cropsr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = nat.lib()
    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(nat.SIGNATURES) == declared
    assert L.crp_abi_version() == nat.ABI_VERSION == 6
    assert b"no CPU fallback" in L.crp_strerror(nat.CRP_ERR_NO_DEVICE)


def test_library_is_gfx950_code_object():
    out = subprocess.run(["strings", "-n", "6", nat.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product must fail loudly, not compute on the CPU."""
    from cropsr_amd import Engine, CropsrHipError
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    with pytest.raises(CropsrHipError) as e:
        Engine(0)
    assert e.value.status == nat.CRP_ERR_NO_DEVICE


def test_product_and_bench_import_no_pytorch():
    """north_star: "no PyTorch" -- neither the package nor bench.py mentions torch as a module (the launcher
    `python -m torch.distributed.run` appears in command lines and comments only)."""
    import re
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "bench_workload.py")]
    for base, _, names in os.walk(os.path.join(ROOT, "cropsr_amd")):
        files += [os.path.join(base, f) for f in names if f.endswith(".py")]
    for f in files:
        assert not re.search(r"^\s*(import|from)\s+torch\b", open(f).read(), flags=re.M), f


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under cropsr_amd/ may touch it."""
    for base, _, files in os.walk(os.path.join(ROOT, "cropsr_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) :
                src = open(os.path.join(base, f)).read()
                assert "liborc" not in src and "from oracle" not in src and "import oracle" not in src, f


# ---------------------------------------------------------------- packing
def _py_planes(b):
    """Independent statement of the plane encoding (include/cropsr_hip.h)."""
    code = {"A": 0, "T": 1, "C": 2, "G": 3}
    n_words = (len(b) + 63) // 64
    hi = [0] * n_words
    lo = [0] * n_words
    up = [0] * n_words
    ac = [0] * n_words
    for k in range(n_words * 64):
        w, bit = divmod(k, 64)
        if k >= len(b):
            hi[w] |= 1 << bit
            lo[w] |= 1 << bit
            continue
        ch = chr(b[k])
        if ch == "U":
            ch = "A"
        if ch == "Z":
            hi[w] |= 1 << bit
            up[w] |= 1 << bit
            continue
        if ch.upper() in code and ch != "u" and ch != "z":
            c = code[ch.upper()]
            hi[w] |= (c >> 1) << bit
            lo[w] |= (c & 1) << bit
            ac[w] |= 1 << bit
            if ch.isupper():
                up[w] |= 1 << bit
    return [np.array(x, dtype=np.uint64) for x in (hi, lo, up, ac)]


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000])
def test_pack_ascii_matches_definition(n):
    from cropsr_amd import pack_ascii
    rng = np.random.default_rng(n)
    b = rng.choice(np.frombuffer(b"ACGTacgtNnUuZzRY')],-*", dtype=np.uint8), n).tobytes()
    got = pack_ascii(b)
    want = _py_planes(b)
    for g, w in zip(got, want):
        assert (g == w).all()


def test_pack_ascii_threads_equal_serial():
    from cropsr_amd import pack_ascii
    rng = np.random.default_rng(1)
    b = rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), 64 * 5000 + 17).tobytes()
    for a, c in zip(pack_ascii(b, 1), pack_ascii(b, 4)):
        assert (a == c).all()


def test_arena_word_arithmetic():
    L = nat.lib()
    assert L.crp_arena_words_for(0) == 1
    assert L.crp_arena_words_for(64) == 2
    assert L.crp_arena_words_for(65) == 3
    assert L.crp_arena_words_total(10) == 11


def test_weight_definition_matches_reference_constants():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "gen_score_terms", os.path.join(ROOT, "cropsr_amd", "csrc", "gen_score_terms.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    _, (w1, w2), consts = gen.generate(os.path.join(ROOT, "cropsr_amd", "csrc", "doench_weights.def"))
    w = np.load(os.path.join(GOLDEN, "weights.npz"))
    assert (np.array(w1) == w["first"]).all() and (np.array(w2) == w["second"]).all()
    assert consts["intersect"] == w["consts"][0] and consts["low_gc"] == w["consts"][1]


# ------------------------------------------------------------ contig table
def test_contig_table_formatted_path():
    t = fasta.contig_table(">c1\nACGT\nAC\n>c2\nGG\n")
    assert list(t.items()) == [("[('c1',", "'ACGTAC'),"), ("('c2',", "'GG')]")]


def test_contig_table_unformatted_path():
    t = fasta.contig_table(">c1\nACGT\n>c2\nGG")
    assert list(t.items()) == [(">c1", "ACGT"), (">c2", "GG")]


def test_contig_table_header_with_blanks_shifts_pairs():
    t = fasta.contig_table(">c1 some description GGCC here\nACGT\n")
    assert list(t.items()) == [("[('c1", "some"), ("description", "GGCC"), ("here',", "'ACGT')]")]


def test_contig_table_odd_tokens_and_duplicates():
    t = fasta.contig_table(">a b\nACGT\n")  # 3 tokens: last key gets ""
    assert list(t.items()) == [("[('a", "b',"), ("'ACGT')]", "")]
    t = fasta.contig_table(">d\nAAAA\n>x\nCC\n>d\nGGGG\n")
    assert list(t.keys()) == ["[('d',", "('x',", "('d',"]  # first key differs by its '['
    t = fasta.contig_table(">x\nAAAA\n>d\nCC\n>d\nGGGG\n")
    assert list(t.items()) == [("[('x',", "'AAAA'),"), ("('d',", "'GGGG')]")]


def test_contig_table_record_without_newline():
    t = fasta.contig_table(">lonely")  # 2*1 != 0+1 -> re-formatted, 1-tuple record
    assert list(t.items()) == [("[('lonely',)]", "")]
    t = fasta.contig_table(">a\nAC\n>lonely")
    assert list(t.items()) == [("[('a',", "'AC'),"), ("('lonely',)]", "")]


# ------------------------------------------------------- chunking / rows
def _reference_walk(size, chunk):
    """Literal walk of the flush conditions of CROPSR.py:451-474."""
    plan, count, counter = [], 0, 0
    for i in range(size):
        count += 1
        if (count == chunk and i < size - 1) or (count < chunk and i == size - 1):
            plan.append((count * counter, count))
            count = 0
            counter += 1
    return plan


@pytest.mark.parametrize("chunk", [1, 2, 3, 7, 10])
def test_flush_plan_equals_literal_walk(chunk):
    for size in range(0, 64):
        assert rows.flush_plan(size, chunk) == _reference_walk(size, chunk), (size, chunk)


def test_flush_plan_documented_cases():
    # 1 124 618 hits: the last chunk re-reads rows 124618.. instead of 1000000..
    assert rows.flush_plan(1124618) == [(0, 1000000), (124618, 124618)]
    # an exact multiple loses its last chunk
    assert rows.flush_plan(1000000) == []
    assert rows.flush_plan(2000000) == [(0, 1000000)]


def test_string_transforms_match_oracle(oracle):
    """rows.plus_text / minus_text against the oracle's literal replace-chain restatement,
    including the exotic U / Z / lower-case / decoration characters."""
    rng = np.random.default_rng(4)
    alpha = np.frombuffer(b"ACGTacgtNUZuz')],RY", dtype=np.uint8)
    for trial in range(20):
        s = rng.choice(alpha, 120).tobytes()
        st = s.decode()
        for pos in range(0, 120, 7):
            if pos - 25 >= 0:
                assert rows.plus_text(st, pos - 25, pos + 5) == oracle.long_sequence(s, pos, False)
                assert rows.plus_text(st, pos - 20, pos) == oracle.short_sequence(s, pos, False)
            if pos - 2 >= 0:
                assert rows.minus_text(st, pos - 2, pos + 28) == oracle.long_sequence(s, pos, True)
                assert rows.minus_text(st, pos + 3, pos + 23) == oracle.short_sequence(s, pos, True)


def test_small_chunk_walk_end_to_end(oracle, tmp_path, monkeypatch):
    """The >1e6-row behaviours (wrong final chunk, lost chunk, backwards ids) on a
    small input by shrinking the chunk size: product rows vs a literal restatement
    of the reference's list slicing."""
    from conftest import OracleBackend
    monkeypatch.setattr(rows, "CHUNK", 16)
    rng = np.random.default_rng(9)
    s = "'" + rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 700).tobytes().decode() + "')]"
    be = OracleBackend(oracle)
    hits = be.scan([s], 20)[0]
    block = rows.ContigRows("[('q',", s, hits, 20)
    ds = rows.Dataset()
    ds.append(block)
    size = len(ds)
    assert size > 48
    np.random.seed(3)
    ids = rows.make_ids(size)
    all_rows = [block.row(k, None) for k in range(size)]
    for index_range, count in rows.flush_plan(size):
        lesser = all_rows[index_range:index_range + count]
        got = ds.rows(index_range, count, ids, index_range, be.rescore)
        assert len(got) == len(lesser)
        id_list = ids.tolist()
        for index, (g, w) in enumerate(zip(got, lesser)):
            assert g[0] == id_list[index_range - index - 1]
            assert g[1:9] == w[1:9]


def test_cli_error_behaviour_of_the_reference(oracle, tmp_path, monkeypatch):
    """CROPSR.py:335-336 exits with a message when --cas9 is missing, before anything is written;
    a missing -g fails in open(None) after time.txt was created (:371, :375, :77-79); a missing
    FASTA raises FileNotFoundError from :58."""
    import io
    from conftest import GOLDEN, oracle_scan_provider
    from cropsr_amd import cli
    monkeypatch.chdir(tmp_path)
    fa = os.path.join(GOLDEN, "probe_tiny.fa")
    gff = os.path.join(GOLDEN, "sample_head.gff")
    out_csv = str(tmp_path / "o.csv")
    be = oracle_scan_provider(oracle)

    args = cli.build_parser().parse_args(["-f", fa, "-g", gff, "-o", out_csv])
    with pytest.raises(SystemExit) as e:
        cli.run(args, backend=be, out=io.StringIO())
    assert e.value.code == "Please select at least one CRISPR system: Cas9"
    assert not os.path.exists(out_csv) and not os.path.exists(tmp_path / "time.txt")

    args = cli.build_parser().parse_args(["-f", fa, "-o", out_csv, "--cas9"])
    with pytest.raises(TypeError):
        cli.run(args, backend=be, out=io.StringIO())
    assert os.path.exists(tmp_path / "time.txt") and not os.path.exists(out_csv)

    args = cli.build_parser().parse_args(["-f", str(tmp_path / "nope.fa"), "-g", gff, "-o", out_csv, "--cas9"])
    with pytest.raises(FileNotFoundError):
        cli.run(args, backend=be, out=io.StringIO())
    # -f is required (CROPSR.py:24-26).  With metavar='' on every option, like the reference, Python 3.10's
    # argparse trips over its own usage formatter while reporting that (AssertionError instead of exit 2)
    with pytest.raises((SystemExit, AssertionError)):
        cli.build_parser().parse_args(["-g", gff, "--cas9"])


def test_generated_scorer_sources_are_current():
    """score_terms.inc / dense_weights.inc / exp_table.inc in the tree are what gen_score_terms.py
    generates from doench_weights.def today (deterministic generator, gather constants included)."""
    import importlib.util
    csrc = os.path.join(ROOT, "cropsr_amd", "csrc")
    spec = importlib.util.spec_from_file_location("gen_score_terms", os.path.join(csrc, "gen_score_terms.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text, _, _ = gen.generate(os.path.join(csrc, "doench_weights.def"))
    with open(os.path.join(csrc, "score_terms.inc")) as f:
        assert f.read() == text


def test_score_tables_equal_sequential_sums():
    """Every entry of the chain-prefix tables (crp_score.h, CRP_SCORE_TAB_DATA) is the chain's start
    value plus the weights of its set bits added in chain order with one rounding each, found at
    the index the kernel computes; checked by re-deriving index and value from the emitted macro."""
    import re
    csrc = os.path.join(ROOT, "cropsr_amd", "csrc")
    text = open(os.path.join(csrc, "score_terms.inc")).read()
    data = [float.fromhex(x) for x in re.search(r"#define CRP_SCORE_TAB_DATA \{ \\\n(.*?)\n    \}", text, re.S).group(1)
            .replace("\\", "").replace(",", " ").split()]
    assert len(data) == int(re.search(r"#define CRP_SCORE_TAB_N (\d+)", text).group(1))
    init = {m.group(1): float.fromhex(m.group(2)) for m in re.finditer(r"#define CRP_PAM_INIT_(\w\w) (\S+)", text)}
    weights = {}
    for line in open(os.path.join(csrc, "doench_weights.def")):
        t = line.split("#", 1)[0].split()
        if t and t[0] == "FIRST":
            weights["%s%02d" % (t[1], int(t[2]))] = float(t[3])
        elif t and t[0] == "SECOND":
            weights["%s%s%02d" % (t[1], t[2], int(t[3]))] = float(t[4])
    body = text[text.index("#define CRP_SCORE_BODY_PAM_TABLES"):]
    looked_up = 0
    for m in re.finditer(r"(\w\w) = crp_tab_at\(score_tab, (\d+), \((.*?) >> (\d+)\) & 0x([0-9a-f]+)u\); /\* (\d+) terms: (.*?) \*/", body):
        chain, base, mul, sh, msk, k, names = m.group(1), int(m.group(2)), m.group(3), int(m.group(4)), int(m.group(5), 16), int(m.group(6)), m.group(7).split()
        assert len(names) == k and msk == ((1 << k) - 1) << 3
        magic = int(re.search(r", 0x([0-9a-f]+)u\)$", mul).group(1), 16)
        assert mul.startswith("__umul24(")
        lo = int(re.search(r">> (\d+)\), 0x", mul).group(1)) if re.search(r">> (\d+)\), 0x", mul) else 0
        # source-word bit of every term: single-base terms sit at their position, pair terms at the position of
        # their first base (+ the shift of their part)
        expr = mul[len("__umul24("):mul.rindex(", 0x")]
        bit = {}
        if chain[0] == "f":
            for name in names:
                bit[name] = int(name[1:]) - 1
        else:
            for part in re.finditer(r"\(\(m(\w)\) & n\w & 0x([0-9a-f]+)u\)( << (\d+))?", expr):
                b1, sel, shift = part.group(1), int(part.group(2), 16), int(part.group(4) or 0)
                for name in names:
                    if name[0] == b1 and (sel >> (int(name[2:]) - 1)) & 1:
                        bit[name] = int(name[2:]) - 1 + shift
        assert len(bit) == k and len(set(bit.values())) == k
        seen = set()
        for pattern in range(1 << k):
            word = sum(1 << (bit[names[i]] - lo) for i in range(k) if (pattern >> i) & 1)
            off = (((word * magic) & 0xFFFFFF) >> sh) & msk
            assert off % 8 == 0 and off not in seen
            seen.add(off)
            v = init[chain]
            for i in range(k):
                if (pattern >> i) & 1:
                    v = v + weights[names[i]]
            assert data[base // 8 + off // 8] == v, (chain, pattern)
        looked_up += k
    assert looked_up >= 40


def test_public_header_is_plain_c(tmp_path):
    """include/cropsr_hip.h is the C ABI: it must compile as C99 (no C++, no HIP or torch types)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "cropsr_hip.h"\nint main(void) { return crp_abi_version() == 0; }\n')
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def host_exp_flavour(oracle):
    """'libm' when this host's np.exp equals glibc's exp (no AVX-512 dispatch), else 'avx512'."""
    x = np.random.default_rng(5).uniform(-9.3, 17.3, 200000)
    return "libm" if (np.exp(x).view(np.uint64) == oracle.exp(x).view(np.uint64)).all() else "avx512"


def test_score_finalize_host_reproduces_this_hosts_reference_bytes(oracle, manifest, tmp_path, monkeypatch):
    """--score-finalize=host (SURVEY.md section 7 hard part 1; CROPSR.py:312-313): the pre-sigmoid sum comes
    from the scan, `1/(1+np.exp(.))` is applied by THIS host's numpy -- so the CSV is the one the
    unmodified reference prints on this host: md5_avx512 where numpy dispatches its AVX-512 exp (this
    development container), md5_libm elsewhere.  The default (gpu) always gives md5_libm."""
    import hashlib
    fa = golden_fasta_path("sample", tmp_path)
    flavour = host_exp_flavour(oracle)
    got, _ = run_cli(tmp_path, monkeypatch, fa, OracleBackend(oracle, finalize="host"), manifest["seed"],
                     extra=("--score-finalize", "host"))
    assert hashlib.md5(got).hexdigest() == manifest["cases"]["sample"]["md5_" + flavour], flavour
    d = tmp_path / "default"
    d.mkdir()
    got, _ = run_cli(d, monkeypatch, fa, OracleBackend(oracle), manifest["seed"])
    assert hashlib.md5(got).hexdigest() == manifest["cases"]["sample"]["md5_libm"]


@pytest.mark.parametrize("name", ["tiny", "multi", "mixed"])
def test_score_finalize_host_on_probes(name, oracle, manifest, tmp_path, monkeypatch):
    """On a host whose np.exp is libm's, host finalisation gives the libm goldens byte for byte (tail rows
    of every written chunk included: they are re-scored from `pre` too); on an AVX-512 host the rows may
    differ from the libm golden only in the score column, by <= 2 ulp."""
    import csv as _csv
    import io
    got, _ = run_cli(tmp_path, monkeypatch, golden_fasta_path(name, tmp_path), OracleBackend(oracle, finalize="host"),
                     manifest["seed"], extra=("--score-finalize", "host"))
    want = read_golden_csv(name)
    if host_exp_flavour(oracle) == "libm":
        assert got == want
        return
    a = list(_csv.reader(io.StringIO(got.decode("latin-1"), newline="")))
    b = list(_csv.reader(io.StringIO(want.decode("latin-1"), newline="")))
    assert len(a) == len(b)
    for ra, rb in zip(a, b):
        assert len(ra) == len(rb)
        if len(ra) == 12 and ra != rb:
            assert ra[:9] == rb[:9] and ra[10:] == rb[10:]
            ulp = abs(int(np.float64(ra[9]).view(np.int64)) - int(np.float64(rb[9]).view(np.int64)))
            assert ulp <= 2
        else:
            assert ra == rb


def test_parallel_fasta_read_equals_the_plain_read(tmp_path, monkeypatch):
    """fasta.read_text_bytes on a "large" file (several threads, os.preadv into one buffer) hands over the same bytes as
    the single read -- also with \\r\\n / \\r line ends, and also when a piece border falls inside a \\r\\n pair --, and
    table_from_bytes / count_byte take what it returns."""
    from cropsr_amd import fasta
    rng = np.random.default_rng(3)
    body = b"".join(b">c%d some text\n" % k + rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), 5000).tobytes() + b"\n"
                    for k in range(40))
    for name, data in (("unix.fa", body), ("dos.fa", body.replace(b"\n", b"\r\n")), ("mac.fa", body.replace(b"\n", b"\r"))):
        path = tmp_path / name
        path.write_bytes(data)
        plain = fasta.read_text_bytes(str(path))
        assert isinstance(plain, bytes) and plain == body
        monkeypatch.setattr(fasta, "PARALLEL_READ_MIN", 1000)
        monkeypatch.setattr(fasta, "READ_PIECE", 4097)  # many pieces; borders inside \r\n pairs
        got = fasta.read_text_bytes(str(path), n_threads=4)
        monkeypatch.undo()
        assert bytes(got) == body, name
        assert fasta.count_byte(got, b">") == body.count(b">") and fasta.count_byte(got, b"\n") == body.count(b"\n")
        assert [(k, bytes(v)) for k, v in fasta.table_from_bytes(got)] == \
            [(k, bytes(v)) for k, v in fasta.table_from_bytes(body)]
