"""Host-side native code under sanitizers (CPU build only; the GPU pool has no ASan).
The formatter, the FASTA loader, the annotation builder and the node's cut are compiled from source with g++, each with a
small driver."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(flags, tmp_path, driver, source, defines=(), args=()):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / driver)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + flags, "-fno-sanitize-recover=all", "-pthread",
           *defines, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", driver + ".cpp"),
           os.path.join(ROOT, "cropsr_amd", "csrc", source), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
    if run.returncode != 0 and "FATAL: ThreadSanitizer: unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow here")
    assert run.returncode == 0 and run.stdout.strip().endswith("OK"), run.stdout + run.stderr


@pytest.mark.parametrize("flags", ["address,undefined", "thread"])
def test_formatter_under_sanitizers(flags, tmp_path):
    _build_and_run(flags, tmp_path, "format_driver", "crp_format.cpp", args=[str(tmp_path / "out.bin")])


@pytest.mark.parametrize("flags,piece", [("address,undefined", 1), ("address,undefined", 64), ("thread", 7)])
def test_fasta_loader_fuzz_under_sanitizers(flags, piece, tmp_path):
    """3000 random inputs with 1-, 7- and 64-byte pieces against a serial restatement."""
    _build_and_run(flags, tmp_path, "fasta_driver", "crp_fasta.cpp", defines=["-DCRP_FASTA_CHUNK_BYTES=%d" % piece])


def test_annotation_builder_fuzz_under_sanitizers(tmp_path):
    """400 random GFF files (soups and well-formed ones with overlapping / nested / degenerate rows, odd attributes, CRLF,
    annotation_info) through crp_annotation_build / _track under ASan + UBSan: the label set of every coordinate and of every
    sampled arena position equals a direct loop over the rows."""
    _build_and_run("address,undefined", tmp_path, "annotation_driver", "crp_annotation.cpp")


def test_node_cut_fuzz_under_sanitizers(tmp_path):
    """20 000 random contig-length lists over 1..17 devices through crp_plan_shares under ASan + UBSan: coverage, order, at
    most world - 1 cuts, and the one-run-per-table property crp_node_gather rests on."""
    _build_and_run("address,undefined", tmp_path, "plan_driver", "crp_plan.cpp", defines=["-I", os.path.join(ROOT, "cropsr_amd", "csrc")])
