"""Start the ranks of a one-node multi-GPU run without an external launcher.

`python bench.py --gpus N` (and `python -m cropsr_amd --gpus N ...`) must work by
themselves: the reference is one process (its only hint of parallelism is the dead
cropsr_functions.py:256-273), so nothing upstream provides a launcher, and
`python -m torch.distributed.run` is only one way to export RANK / WORLD_SIZE /
LOCAL_RANK.  spawn_ranks() is the other: the calling process -- which must not have
touched HIP or RCCL yet, and never does -- starts N FRESH child processes of the same
command line (subprocess, no exec of a process that holds a GPU), each with

    RANK, LOCAL_RANK = 0..N-1      WORLD_SIZE = N      MASTER_ADDR = 127.0.0.1
    CROPSR_RDZV_KEY  = a key unique to this launch (rendezvous.Group.from_env)
    CROPSR_LAUNCHED  = 1           (the child must not launch again)

and waits for them.  Standard output and error are inherited, so rank 0's one JSON
line (or the CLI's stdout) reaches the caller unchanged.  The parent's exit status is
0 only if every rank's is; the first rank that fails decides it, its peers get a grace
period to leave through the rendezvous abort channel and are then terminated -- by
their exact pids, never by a pattern.  A launcher that is itself told to stop (SIGTERM /
SIGHUP / SIGINT: a driver's step time-out, `timeout(1)`, a scheduler's cancel) stops its
ranks before it leaves with 128 + signal; should it be killed outright (SIGKILL), the
ranks get SIGTERM from the kernel (PR_SET_PDEATHSIG) -- no rank is ever left behind
holding a GPU.
"""
import ctypes
import os
import signal
import subprocess
import sys
import threading
import time

ENV_MARK = "CROPSR_LAUNCHED"
ABORTED_WITH_PEER = 3  # exit status of rendezvous.Group._die


def wanted(n_ranks, env=None):
    """True when this process should start the ranks itself: more than one rank is asked
    for and no launcher (torch.distributed.run, a previous spawn_ranks) has set up the group."""
    env = os.environ if env is None else env
    return n_ranks > 1 and int(env.get("WORLD_SIZE", "1")) <= 1 and env.get(ENV_MARK) != "1"


def rank_env(rank, world, key, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "CROPSR_RDZV_KEY": key, ENV_MARK: "1"})
    env.setdefault("MASTER_PORT", "0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this stack
    return env


class _Stopped(Exception):
    def __init__(self, signum):
        super().__init__(signum)
        self.signum = signum


def _die_with_parent(launcher_pid):
    """preexec_fn of a rank (between fork and exec, so before anything could touch a GPU): SIGTERM when the launcher dies --
    and if it died between the fork and this call (the signal would never come), leave at once.  libc's prctl is looked up
    HERE, in the parent: the forked child only calls it (a dlopen between fork and exec is not safe in a process that
    may have threads -- ADVICE r04)."""
    try:
        prctl = ctypes.CDLL(None, use_errno=True).prctl
    except Exception:  # not Linux: the signal handlers of spawn_ranks are all there is
        prctl = None
    sigterm = int(signal.SIGTERM)

    def arm():
        if prctl is None:
            return
        try:
            prctl(1, sigterm, 0, 0, 0)  # PR_SET_PDEATHSIG
        except Exception:
            return
        if os.getppid() != launcher_pid:
            os._exit(128 + sigterm)
    return arm


def spawn_ranks(argv, world, timeout_s=None, grace_s=20.0, poll_s=0.05, env=None):
    """Run `argv` as `world` processes (ranks 0..world-1) and wait.  Returns the exit
    status for the caller: 0 if all ranks returned 0, else the status of the first rank
    seen failing (a rank killed by a signal counts as 128 + signal); 124 on timeout;
    128 + signal when the launcher itself was told to stop (its ranks are stopped first)."""
    key = "self_%d_%s" % (os.getpid(), os.urandom(4).hex())
    procs = []
    # SIGTERM / SIGHUP / SIGINT while the ranks run: raise out of the wait so that the `finally` below stops them
    # (the default action would end this process at once and orphan N ranks inside a collective, GPUs held)
    old = {}
    if threading.current_thread() is threading.main_thread():
        def on_signal(signum, _frame):
            raise _Stopped(signum)
        for sig in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
            try:
                old[sig] = signal.signal(sig, on_signal)
            except (OSError, ValueError):
                pass
    try:
        for r in range(world):
            procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, key, env), preexec_fn=_die_with_parent(os.getpid())))
        deadline = None if timeout_s is None else time.monotonic() + timeout_s
        status, first_fail_at, seen = 0, None, set()
        while True:
            codes = [p.poll() for p in procs]
            for r, c in enumerate(codes):
                if c in (None, 0) or r in seen:
                    continue
                seen.add(r)
                c = c if c > 0 else 128 - c
                sys.stderr.write("[cropsr_amd.launch] rank %d exited with status %d\n" % (r, c))
                # status 3 is how a rank leaves when ANOTHER rank died or asked for an abort (rendezvous.Group._die):
                # the rank at fault decides the status, whichever of them the polling sees first
                if status == 0 or (status == ABORTED_WITH_PEER and c != ABORTED_WITH_PEER):
                    status = c
                if first_fail_at is None:
                    first_fail_at = time.monotonic()
            if all(c is not None for c in codes):
                return status
            now = time.monotonic()
            if deadline is not None and now > deadline and status == 0:
                status, first_fail_at = 124, now - grace_s  # no grace: the run is over time
                sys.stderr.write("[cropsr_amd.launch] ranks still running after %.0f s: stopping them\n" % timeout_s)
            if first_fail_at is not None and now - first_fail_at > grace_s:
                break  # the peers did not leave by themselves (abort channel): stop them below
            time.sleep(poll_s)
        return status
    except _Stopped as e:
        sys.stderr.write("[cropsr_amd.launch] signal %d: stopping the ranks\n" % e.signum)
        return 128 + e.signum
    finally:
        for sig in old:  # (a second signal must not interrupt the clean-up)
            signal.signal(sig, signal.SIG_IGN)
        _stop(procs)
        for sig, handler in old.items():
            signal.signal(sig, handler)


def _stop(procs):
    """Terminate whatever is still running (exact pids), then make sure it is gone."""
    alive = [p for p in procs if p.poll() is None]
    for p in alive:
        try:
            p.terminate()
        except OSError:
            pass
    t0 = time.monotonic()
    while alive and time.monotonic() - t0 < 5.0:
        alive = [p for p in alive if p.poll() is None]
        time.sleep(0.05)
    for p in alive:
        try:
            p.kill()
        except OSError:
            pass
    for p in procs:
        try:
            p.wait(timeout=5)
        except Exception:
            pass
