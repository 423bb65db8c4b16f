"""cropsr_amd.launch: `--gpus N` without an external launcher starts its own ranks.

No GPU here: the ranks are small Python programs that meet over cropsr_amd.rendezvous exactly as the
ranks of bench.py do; what is checked is the launcher's contract -- fresh processes with RANK / WORLD_SIZE /
LOCAL_RANK / a private rendezvous key, rank 0's stdout passed through, exit status 0 only if every rank's
was, a failing rank's peers not left behind.  The GPU-side counterpart (bench.py --gpus 2 --share-gpu0
with no launcher) is in tests/test_gpu_parity.py."""
import os
import subprocess
import sys
import textwrap
import time

from conftest import ROOT

from cropsr_amd import launch


def _script(tmp_path, body):
    p = tmp_path / "rank_prog.py"
    p.write_text("import os, sys, json\nsys.path.insert(0, %r)\n" % ROOT + textwrap.dedent(body))
    return str(p)


def _run(tmp_path, body, world, **kw):
    """Run the launcher itself in a child (it inherits stdout: capture it there)."""
    prog = _script(tmp_path, body)
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        from cropsr_amd import launch
        sys.exit(launch.spawn_ranks([sys.executable, %r], %d, **%r))
        """ % (ROOT, prog, world, kw)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", launch.ENV_MARK)}
    env["CROPSR_RDZV_DIR"] = str(tmp_path)
    t0 = time.time()
    r = subprocess.run([sys.executable, str(driver)], env=env, capture_output=True, text=True, timeout=120)
    return r, time.time() - t0


def test_wanted_only_without_a_launcher():
    assert launch.wanted(2, {}) and launch.wanted(8, {"WORLD_SIZE": "1"})
    assert not launch.wanted(1, {})
    assert not launch.wanted(2, {"WORLD_SIZE": "2", "RANK": "1"})       # torch.distributed.run set the group up
    assert not launch.wanted(2, {launch.ENV_MARK: "1"})                  # a rank we started ourselves
    env = launch.rank_env(3, 4, "k", base={})
    assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"]) == ("3", "3", "4", "127.0.0.1")
    assert env["CROPSR_RDZV_KEY"] == "k" and env[launch.ENV_MARK] == "1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_ranks_meet_and_rank0_line_passes_through(tmp_path):
    body = """
        from cropsr_amd import rendezvous
        g = rendezvous.Group.from_env()
        ranks = g.all_gather((g.rank, g.local_rank, os.getpid()))
        total = g.allreduce([g.rank + 1], "sum")[0]
        if g.rank == 0:
            print(json.dumps({"n_gpus": g.world, "ranks": [r[:2] for r in ranks], "sum": total,
                              "pids_distinct": len(set(r[2] for r in ranks)) == g.world}), flush=True)
        g.close()
    """
    r, _ = _run(tmp_path, body, 3)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1  # ONE line, from rank 0
    import json
    line = json.loads(lines[0])
    assert line == {"n_gpus": 3, "ranks": [[0, 0], [1, 1], [2, 2]], "sum": 6.0, "pids_distinct": True}
    assert not [f for f in os.listdir(tmp_path) if f.startswith("cropsr_rdzv_")]  # rendezvous file removed


def test_a_failing_rank_decides_the_exit_status_and_takes_its_peers_along(tmp_path):
    # rank 1 dies after the rendezvous; rank 0 and 2 sit in a collective that can never complete: the abort
    # channel (their connection to the dead rank's hub entry closes) ends them, the launcher reports 7
    body = """
        from cropsr_amd import rendezvous
        g = rendezvous.Group.from_env()
        g.barrier()
        if g.rank == 1:
            os._exit(7)
        g.barrier()
        print("not reached", flush=True)
    """
    r, dt = _run(tmp_path, body, 3, grace_s=30.0)
    assert r.returncode == 7, (r.returncode, r.stderr)
    assert "not reached" not in r.stdout
    assert dt < 25, "the peers left through the abort channel, not through the launcher's grace period"


def test_peers_that_do_not_leave_are_stopped_after_the_grace_period(tmp_path):
    body = """
        import time
        if os.environ["RANK"] == "0":
            sys.exit(5)
        time.sleep(600)
    """
    r, dt = _run(tmp_path, body, 2, grace_s=1.0)
    assert r.returncode == 5 and dt < 30


def test_timeout(tmp_path):
    r, dt = _run(tmp_path, "import time\ntime.sleep(600)\n", 2, timeout_s=1.0)
    assert r.returncode == 124 and dt < 30


def _pids_alive(pids):
    out = []
    for pid in pids:
        try:
            with open("/proc/%d/stat" % pid) as f:
                if f.read().rsplit(")", 1)[1].split()[0] != "Z":
                    out.append(pid)
        except OSError:
            pass
    return out


def _launcher_with_sleeping_ranks(tmp_path, world=3):
    """A launcher process whose ranks write their pids and sleep; returns (Popen of the launcher, rank pids)."""
    prog = _script(tmp_path, """
        import time
        open(os.path.join(%r, "pid%%s" %% os.environ["RANK"]), "w").write(str(os.getpid()))
        time.sleep(600)
    """ % str(tmp_path))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        from cropsr_amd import launch
        sys.exit(launch.spawn_ranks([sys.executable, %r], %d))
        """ % (ROOT, prog, world)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", launch.ENV_MARK)}
    p = subprocess.Popen([sys.executable, str(driver)], env=env, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while time.time() - t0 < 60 and not all(os.path.exists(tmp_path / ("pid%d" % r)) and (tmp_path / ("pid%d" % r)).read_text()
                                            for r in range(world)):
        time.sleep(0.05)
    pids = [int((tmp_path / ("pid%d" % r)).read_text()) for r in range(world)]
    assert len(_pids_alive(pids)) == world
    return p, pids


def test_a_launcher_told_to_stop_takes_its_ranks_along(tmp_path):
    """SIGTERM to the launcher (a driver's step time-out, timeout(1)) must not orphan the ranks: the handler
    raises out of the wait, the ranks are terminated by pid, the status is 128 + 15."""
    import signal
    p, pids = _launcher_with_sleeping_ranks(tmp_path)
    p.send_signal(signal.SIGTERM)
    err = p.communicate(timeout=60)[1]
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err)
    assert "stopping the ranks" in err
    t0 = time.time()
    while _pids_alive(pids) and time.time() - t0 < 10:
        time.sleep(0.05)
    assert not _pids_alive(pids)


def test_a_launcher_killed_outright_takes_its_ranks_along(tmp_path):
    """SIGKILL cannot be handled: the ranks were started with PR_SET_PDEATHSIG and get SIGTERM from the kernel."""
    import signal
    p, pids = _launcher_with_sleeping_ranks(tmp_path, world=2)
    p.send_signal(signal.SIGKILL)
    p.communicate(timeout=60)
    assert p.returncode == -signal.SIGKILL
    t0 = time.time()
    while _pids_alive(pids) and time.time() - t0 < 10:
        time.sleep(0.05)
    assert not _pids_alive(pids)


def test_bench_parent_does_not_touch_the_gpu_library(tmp_path):
    """`python bench.py --gpus 2` in a process without a launcher takes the spawn branch BEFORE importing the
    engine: with a stub in place of the ranks' interpreter-side work the parent just relays the status.  (The
    ranks themselves need a GPU; here they fail at crp_init with 'no usable HIP device', which is what the
    parent must report -- non-zero, no hang.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", launch.ENV_MARK)}
    env["CROPSR_RDZV_DIR"] = str(tmp_path)
    env["CROPSR_RDZV_CONNECT_TIMEOUT"] = "20"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.002", "--steps", "1",
                        "--share-gpu0", "--cpu-sample-bases", "0", "--launch-timeout", "100"],
                       env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode != 0
    assert "no usable HIP device" in r.stderr and "--gpus 2 but WORLD_SIZE=1" not in r.stderr
    assert "[cropsr_amd.launch] rank" in r.stderr
