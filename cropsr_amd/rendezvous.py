"""Process group of a multi-GPU run, without PyTorch: one process per GPU on one node.

The reference is a single process (its only hint of parallelism is the dead
cropsr_functions.py:256-273), so nothing here mirrors reference code.  The DATA of
the path moves over xGMI inside libcropsr_hip.so (RCCL: crp_gather_hits,
crp_offtarget_reduce).  This module is the small CONTROL plane around it:

  * find the other ranks: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT
    as `python -m torch.distributed.run` (or any launcher) exports them.  Rank 0
    listens on an ephemeral TCP port of 127.0.0.1 and publishes it in a file
    keyed by the launcher (its pid and MASTER_PORT); the others poll that file.
    (MASTER_PORT itself belongs to the launcher's own store.)  CROPSR_RDZV_ENDPOINT=
    host:port names the listening socket explicitly instead (no file);
  * carry the RCCL unique id from rank 0 to the others (bcast);
  * exchange small Python objects (all_gather), agree on errors before a data
    collective (check), fence (barrier), sum/max a few numbers (allreduce);
  * send_array / recv_array: numpy arrays over the same sockets -- the host
    transport of the gatherv, used by the CPU tests and by rehearsals that put
    several ranks on one GPU, where RCCL cannot run.

Star topology through rank 0; every operation is a collective that all ranks
call in the same order.  Messages: 8-byte length + pickle (arrays: raw bytes),
read back with an unpickler that builds plain containers, numbers, strings and
numpy arrays only.  Rank 0 listens on 127.0.0.1 and admits only clients that
present the run's random token (published in the 0600 rendezvous file).

A second connection per rank is the ABORT channel, watched by a daemon thread: a
rank that dies (its sockets close) or calls Group.abort(msg) makes every other
rank print the reason and exit with status 3 -- also out of a blocking RCCL
call, which has no time-out of its own.  A collective that cannot complete ends
in a non-zero exit, never in a hang.  Group.close() is itself a barrier, so a
rank that merely finishes first does not look like one that died.
"""
import hmac
import json
import os
import pickle
import select
import socket
import struct
import sys
import tempfile
import threading
import time

import numpy as np

_CONNECT_TIMEOUT_S = float(os.environ.get("CROPSR_RDZV_CONNECT_TIMEOUT", "300"))  # finding the other ranks
_TIMEOUT_S = float(os.environ.get("CROPSR_RDZV_TIMEOUT", "1800"))  # a live rank may be slow; a dead one is caught by the abort channel


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)))
    sock.sendall(payload)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("peer closed the control connection")
        got += k
    return buf


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


# What may come out of a message: plain containers, numbers, strings, bytes -- and numpy arrays of them.
# Nothing else is ever sent, so nothing else is ever built from bytes read off a socket.
_ALLOWED_GLOBALS = {
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
}


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("control message names %s.%s: refused" % (module, name))


def _loads(data):
    import io
    return _Unpickler(io.BytesIO(bytes(data))).load()


LAST_GROUP = None  # the most recent multi-rank Group of this process (for top-level abort handlers)


class RankError(RuntimeError):
    """Raised on EVERY rank by Group.check when any rank reported an error."""


class Group:
    """world processes; rank 0 is the hub."""

    def __init__(self, rank, world, local_rank=0, endpoint=None, rdzv_file=None):
        self.rank, self.world, self.local_rank = int(rank), int(world), int(local_rank)
        self._peers = {}     # hub: rank -> socket
        self._hub = None     # others: socket to rank 0
        self._abort_peers = {}  # hub: rank -> abort-channel socket
        self._abort_hub = None  # others: abort-channel socket to rank 0
        self._closing = False
        self._listener = None
        self._file = None
        self.connect_s = 0.0  # how long this rank waited for the group to form (start-up skew included)
        t_start = time.time()
        # a run's clients prove they belong to it: rank 0 draws a token and publishes it in the rendezvous file
        # (mode 0600); with an explicit endpoint there is no file, and the token is CROPSR_RDZV_TOKEN (or empty)
        self._token = os.environ.get("CROPSR_RDZV_TOKEN", "")
        if self.world == 1:
            return
        if self.rank == 0:
            self._listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            self._listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            if endpoint:
                host, port = endpoint.rsplit(":", 1)
                if not self._token and host not in ("127.0.0.1", "localhost", "::1"):
                    raise ValueError("CROPSR_RDZV_ENDPOINT on a non-loopback address needs CROPSR_RDZV_TOKEN")
                self._listener.bind((host, int(port)))
            else:
                self._listener.bind(("127.0.0.1", 0))
            self._listener.listen(self.world)
            self._listener.settimeout(_CONNECT_TIMEOUT_S)
            if not endpoint:
                host, port = self._listener.getsockname()
                if not self._token:
                    self._token = os.urandom(16).hex()
                tmp = rdzv_file + ".%d.tmp" % os.getpid()
                try:
                    os.unlink(tmp)  # (a leftover of a crashed run with the same pid)
                except OSError:
                    pass
                # O_EXCL | O_NOFOLLOW: never write the token through a file or link someone else put there
                with open(os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600), "w") as f:
                    json.dump({"host": host, "port": port, "world": self.world, "pid": os.getpid(), "token": self._token}, f)
                os.replace(tmp, rdzv_file)  # atomic: a reader never sees half a file
                self._file = rdzv_file
            while len(self._peers) < self.world - 1 or len(self._abort_peers) < self.world - 1:
                conn, _ = self._listener.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(5.0)  # a client that says nothing must not hold up the others
                try:
                    hello = _loads(_recv_msg(conn))
                except Exception:
                    hello = None
                conn.settimeout(_TIMEOUT_S)
                # Whatever arrives on the port is judged before it is used: the token first (constant time),
                # then the types, and nothing a stranger sends may raise out of this loop.
                try:
                    ok = (isinstance(hello, dict) and isinstance(hello.get("token"), str)
                          and hmac.compare_digest(hello["token"].encode(), self._token.encode())
                          and type(hello.get("rank")) is int and type(hello.get("world")) is int
                          and hello["world"] == self.world and 0 < hello["rank"] < self.world)
                    table = (self._abort_peers if hello.get("abort") else self._peers) if ok else None
                    if ok and hello["rank"] in table:
                        ok = False
                except Exception:
                    ok = False
                if not ok:
                    conn.close()  # not one of ours: a stray client of another run, or garbage
                    continue
                table[hello["rank"]] = conn
            for r in sorted(self._peers):
                _send_msg(self._peers[r], pickle.dumps("welcome"))
            for sock in self._abort_peers.values():
                sock.settimeout(None)
        else:
            deadline = time.time() + _CONNECT_TIMEOUT_S
            while True:
                try:
                    if endpoint:
                        host, port = endpoint.rsplit(":", 1)
                    else:
                        with open(rdzv_file) as f:
                            info = json.load(f)
                        if info.get("world") != self.world:
                            raise OSError("stale rendezvous file")
                        host, port = info["host"], info["port"]
                        self._token = info.get("token", "")
                    s = socket.create_connection((host, int(port)), timeout=5)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(_TIMEOUT_S)
                    _send_msg(s, pickle.dumps({"rank": self.rank, "world": self.world, "token": self._token}))
                    a = socket.create_connection((host, int(port)), timeout=5)
                    a.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    a.settimeout(_TIMEOUT_S)
                    _send_msg(a, pickle.dumps({"rank": self.rank, "world": self.world, "abort": True, "token": self._token}))
                    if _loads(_recv_msg(s)) != "welcome":
                        raise OSError("unexpected greeting")
                    a.settimeout(None)
                    self._hub, self._abort_hub = s, a
                    break
                except (OSError, ValueError, ConnectionError):
                    if time.time() > deadline:
                        raise TimeoutError("rank %d: no rendezvous with rank 0 within %.0f s" % (self.rank, _CONNECT_TIMEOUT_S))
                    time.sleep(0.05)
        threading.Thread(target=self._watch, name="cropsr-abort-watch", daemon=True).start()
        self.connect_s = time.time() - t_start
        global LAST_GROUP
        LAST_GROUP = self

    # ------------------------------------------------------------- abort channel
    def _die(self, why):
        sys.stderr.write("[cropsr_amd rank %d] aborting: %s\n" % (self.rank, why))
        sys.stderr.flush()
        os._exit(3)

    def _watch(self):
        """Daemon thread: leave the process as soon as any rank has died or asked for an abort."""
        try:
            if self.rank == 0:
                socks = dict((s, r) for r, s in self._abort_peers.items())
                while socks and not self._closing:
                    ready, _, _ = select.select(list(socks), [], [], 0.5)
                    for s in ready:
                        try:
                            msg = _recv_msg(s)
                            why = "rank %d: %s" % (socks[s], _loads(msg))
                        except Exception:
                            why = "rank %d died (its connection closed)" % socks[s]
                        if self._closing:
                            return
                        for o in socks:
                            if o is not s:
                                try:
                                    _send_msg(o, pickle.dumps(why))
                                except OSError:
                                    pass
                        self._die(why)
            else:
                try:
                    why = _loads(_recv_msg(self._abort_hub))
                except Exception:
                    why = "rank 0 died (its connection closed)"
                if not self._closing:
                    self._die(why)
        except Exception:  # the sockets went away under a closing group
            return

    def abort(self, why):
        """Make every rank exit (status 3) with this reason; does not return."""
        why = str(why)
        try:
            if self.rank == 0:
                for s in self._abort_peers.values():
                    _send_msg(s, pickle.dumps("rank 0: " + why))
            elif self._abort_hub is not None:
                _send_msg(self._abort_hub, pickle.dumps(why))
        except OSError:
            pass
        self._die(why)

    # ------------------------------------------------------------------ set-up
    @classmethod
    def from_env(cls, env=None):
        """The group torch.distributed.run (or a compatible launcher) describes in the
        environment; None for a single process."""
        env = os.environ if env is None else env
        world = int(env.get("WORLD_SIZE", "1"))
        if world <= 1:
            return None
        rank = int(env.get("RANK", "0"))
        local = int(env.get("LOCAL_RANK", str(rank)))
        endpoint = env.get("CROPSR_RDZV_ENDPOINT")
        path = None
        if not endpoint:
            # all ranks are children of one launcher process: its pid and port name the run
            key = env.get("CROPSR_RDZV_KEY") or "%s_%s" % (os.getppid(), env.get("MASTER_PORT", "0"))
            path = os.path.join(env.get("CROPSR_RDZV_DIR") or tempfile.gettempdir(), "cropsr_rdzv_%s.json" % key)
        return cls(rank, world, local, endpoint, path)

    def close(self):
        """Collective: returns once every rank is closing (a rank that finishes first must not look dead)."""
        if self.world > 1 and not self._closing and (self._hub is not None or self._peers):
            self._closing = True
            try:
                self.barrier()
            except Exception:
                pass
        self._closing = True
        for s in list(self._peers.values()) + list(self._abort_peers.values()) + [self._hub, self._abort_hub, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._abort_peers, self._hub, self._abort_hub, self._listener = {}, {}, None, None, None
        if self._file:
            try:
                os.unlink(self._file)
            except OSError:
                pass
            self._file = None

    # ------------------------------------------------------------- collectives
    def all_gather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank.  A peer that is gone ends this rank the way the abort
        channel would (status 3, "another rank died"): an exception out of here would race the watcher thread for the
        exit status, and the launcher takes the status of the rank at fault, not of the ranks that noticed."""
        if self.world == 1:
            return [obj]
        r = 0
        try:
            if self.rank == 0:
                objs = [obj] + [None] * (self.world - 1)
                for r, s in self._peers.items():
                    objs[r] = _loads(_recv_msg(s))
                blob = pickle.dumps(objs)
                for r, s in self._peers.items():
                    _send_msg(s, blob)
                return objs
            _send_msg(self._hub, pickle.dumps(obj))
            return _loads(_recv_msg(self._hub))
        except OSError as e:
            self._die("lost the control connection to rank %d in a collective (%s)" % (r, e))

    def bcast(self, obj, src=0):
        return self.all_gather(obj if self.rank == src else None)[src]

    def barrier(self):
        self.all_gather(None)

    def allreduce(self, values, op="sum"):
        """Element-wise sum or max of a short list of numbers over all ranks."""
        rows = self.all_gather([float(v) for v in values])
        f = max if op == "max" else sum
        return [f(col) for col in zip(*rows)]

    def check(self, error=None):
        """Call on every rank BEFORE a data collective, with this rank's error (or None).  If any
        rank has one, every rank raises RankError with the same text -- no rank is left waiting
        in a collective its peers never enter."""
        errors = self.all_gather(None if error is None else str(error))
        bad = [(r, e) for r, e in enumerate(errors) if e is not None]
        if bad:
            raise RankError("; ".join("rank %d: %s" % (r, e) for r, e in bad))

    # -------------------------------------------------- arrays (host transport)
    def send_array(self, arr, dst=0):
        """rank != dst: one numpy array to dst (must be the hub, rank 0)."""
        if dst != 0 or self.rank == 0:
            raise ValueError("arrays travel to rank 0 only")
        a = np.ascontiguousarray(arr)
        try:
            _send_msg(self._hub, pickle.dumps((a.dtype.str, a.shape)))
            _send_msg(self._hub, memoryview(a).cast("B") if a.size else b"")
        except OSError as e:
            self._die("lost the control connection to rank 0 while sending a table (%s)" % e)

    def recv_array(self, src):
        """rank 0: the array rank `src` sent."""
        s = self._peers[src]
        try:
            dtype, shape = _loads(_recv_msg(s))
            raw = _recv_msg(s)
        except OSError as e:
            self._die("lost the control connection to rank %d while receiving a table (%s)" % (src, e))
        return np.frombuffer(raw, dtype=np.dtype(dtype)).reshape(shape)
