"""Builds and wires the loop-back test double of librccl.so.1 (tests/native/fake_rccl.cpp).  TEST INFRASTRUCTURE.

The product loads RCCL with dlopen("librccl.so.1") (cropsr_amd/csrc/crp_comm.cpp).  A child test process whose
LD_LIBRARY_PATH starts with the double's directory gets the double instead -- the product is untouched.  The double
accepts duplicate devices and checks the protocol (matched send/recv pairs, byte counts, everybody in a collective), which
is what lets the N > 1 RCCL branches of crp_node.cpp and crp_comm.cpp run, and be checked, on a one-GPU box."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "fake_rccl.cpp")
OUT_DIR = os.path.join(ROOT, "tests", "native", "_fake_rccl")
OUT = os.path.join(OUT_DIR, "librccl.so.1")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build():
    """hipcc cross-compiles for gfx950 without a GPU; rebuilt when the source is newer.  Returns the directory."""
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        os.makedirs(OUT_DIR, exist_ok=True)
        tmp = OUT + ".tmp%d" % os.getpid()
        subprocess.run([HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "--offload-arch=gfx950", SRC,
                        "-Wl,-soname,librccl.so.1", "-o", tmp], check=True)
        os.replace(tmp, OUT)
    return OUT_DIR


class Session:
    """One test's use of the double: a mailbox directory of its own in /dev/shm (removed afterwards, whatever the ranks left
    in it), a stats file every communicator appends a line to when it is destroyed, and the environment for the children."""

    def __init__(self, **knobs):
        self.lib_dir = build()
        self.shm = tempfile.mkdtemp(prefix="fakerccl-", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        self.stats_file = os.path.join(self.shm, "stats.jsonl")
        self.knobs = {k: str(v) for k, v in knobs.items()}

    def env(self, **extra):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CROPSR_LAUNCHED", "CROPSR_GATHER")}
        env["LD_LIBRARY_PATH"] = self.lib_dir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
        env["FAKE_RCCL_SHM_DIR"] = self.shm
        env["FAKE_RCCL_STATS_FILE"] = self.stats_file
        env["PYTHONPATH"] = ROOT + os.pathsep + os.path.join(ROOT, "tests")
        env.update(self.knobs)
        env.update({k: str(v) for k, v in extra.items()})
        return env

    def stats(self):
        if not os.path.exists(self.stats_file):
            return []
        with open(self.stats_file) as f:
            return [json.loads(line) for line in f if line.strip()]

    def close(self):
        shutil.rmtree(self.shm, ignore_errors=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def run_child(self, func, *args, timeout=900, **env_extra):
        """`tests/test_fake_rccl.py:<func>(*args)` in a child process under the double; returns (CompletedProcess, result)
        where result is what the child wrote to its result file (JSON) or None."""
        out = os.path.join(self.shm, "result-%s.json" % func)
        code = "import test_fake_rccl as t; t.%s(%s)" % (func, ", ".join(repr(a) for a in (out,) + args))
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=self.env(**env_extra),
                           cwd=os.path.join(ROOT, "tests"))
        result = None
        if os.path.exists(out):
            with open(out) as f:
                result = json.load(f)
        return p, result


def in_process_stats():
    """fake_rccl_stats() of the double loaded in THIS process (a child calls it after its node is closed)."""
    import ctypes
    lib = ctypes.CDLL("librccl.so.1")
    lib.fake_rccl_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    v = (ctypes.c_uint64 * 8)()
    lib.fake_rccl_stats(v)
    keys = ("pairs", "p2p_bytes", "collectives", "groups", "mismatches", "inits", "hangs", "aborts")
    return dict(zip(keys, (int(x) for x in v)))
