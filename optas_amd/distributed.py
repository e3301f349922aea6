"""Multi-GPU plumbing for one-process-per-GPU runs (bench.py, user scripts).  No torch: the RCCL communicator lives inside
liboptas_hip (``oh_comm_*``, include/optas_hip.h).

MPC instances are independent, so the data path needs no collective at all: each rank solves its own shard.  The only exchange is ONE
broadcast of the kinematic constants (``oh_chain``, 2952 bytes) from rank 0 over RCCL/xGMI.  What Python does is the rendezvous: rank 0
asks the library for an RCCL unique id and hands the 128 bytes to the other ranks; then every rank calls ``oh_comm_init``.  The launcher
contract is the usual one: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT in the environment (``python -m torch.distributed.run ...``
sets them; the workers themselves never load it).

Two carriers for the 128 bytes (``OPTAS_RDZV=file|tcp``, default ``file``), both SINGLE-NODE by design (the 8 GPUs of one node):

* ``file`` -- a record in a directory only this user can enter (``$OPTAS_RDZV_DIR`` or ``<tmp>/optas_amd_<uid>``, mode 0700).  The record
  is ``id (128 B) | pid of rank 0 | start time of that process``: a reader accepts it only while that very process is alive
  (``/proc/<pid>/stat``), so the file of an earlier job that died before cleaning up -- same shell, same port -- is never taken for this
  one's, however the name collides.  Rank 0 removes whatever sits at the path and creates its record exclusively (O_EXCL) under a temporary
  name before renaming it into place.
* ``tcp`` -- for launchers whose workers do not share a temporary directory: rank 0 listens on ``MASTER_ADDR:(MASTER_PORT + 1 + k)`` (the
  launcher's own store owns MASTER_PORT), each other rank connects, sends the launch tag and receives the id.

A rank that waits longer than ``timeout`` says what it was waiting for; if rank 0 dies *after* publishing, the others are inside
``ncclCommInitRank`` and RCCL's own timeout applies (``NCCL_DEBUG=WARN`` is exported by default so that its diagnosis reaches stderr).
"""
from __future__ import annotations

import ctypes as C
import os
import socket
import struct
import tempfile
import time
from typing import Callable, Optional, Tuple

from . import _lib

_REC = struct.Struct("<qq")  # pid, start time (clock ticks since boot) of the publishing process


def shard(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous partition [lo, hi) of n_total instances for this rank."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def launch_tag() -> str:
    """Names one launch: the launcher's port, its run id and its process id (every worker of one launch has the same parent)."""
    tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
    secret = os.environ.get("OPTAS_RDZV_SECRET")  # optional: a random string the launcher hands to every rank makes the tag unguessable (TCP carrier)
    return f"{tag}_{secret}" if secret else tag


def rendezvous_dir() -> str:
    d = os.environ.get("OPTAS_RDZV_DIR")
    if d:
        return d
    return os.path.join(tempfile.gettempdir(), f"optas_amd_{os.getuid()}")


def rendezvous_path(tag: Optional[str] = None) -> str:
    """File the unique id travels through (see the module docstring for why a colliding name is harmless)."""
    return os.path.join(rendezvous_dir(), f"optas_amd_rdzv_{launch_tag() if tag is None else tag}.id")


def _proc_start(pid: int) -> Optional[int]:
    """Start time of a live process (field 22 of /proc/<pid>/stat), None if it does not exist."""
    try:
        with open(f"/proc/{pid}/stat", "rb") as fh:
            stat = fh.read().decode("ascii", "replace")
        return int(stat[stat.rindex(")") + 2 :].split()[19])
    except (OSError, ValueError, IndexError):
        return None


def _private_dir(path: str) -> None:
    d = os.path.dirname(path) or "."
    try:
        os.makedirs(d, mode=0o700, exist_ok=True)
    except OSError as e:
        raise RuntimeError(f"rendezvous directory {d} cannot be created ({e}); set OPTAS_RDZV_DIR or OPTAS_RDZV=tcp") from e
    _check_private(d)


def _check_private(d: str) -> None:
    """Both sides of the file carrier: the directory must belong to this user and be writable by nobody else -- in a shared directory
    another user could publish a record of a live process of theirs and redirect the RCCL id.  (The system tmp is refused for the same reason:
    the default is a 0700 sub-directory of it.)"""
    st = os.stat(d)
    if st.st_uid != os.getuid():
        raise RuntimeError(f"rendezvous directory {d} belongs to another user; set OPTAS_RDZV_DIR to a directory of your own, or OPTAS_RDZV=tcp")
    if st.st_mode & 0o022:
        raise RuntimeError(f"rendezvous directory {d} is group- or world-writable; use a private directory (chmod 700), or OPTAS_RDZV=tcp")


def exchange_unique_id(rank: int, world: int, make_id: Callable[[], bytes], path: Optional[str] = None, timeout: float = 300.0) -> bytes:
    """File carrier.  Rank 0 creates the id and publishes ``id | pid | start time`` atomically; the others wait for a record whose
    publisher is alive."""
    path = rendezvous_path() if path is None else path
    if rank == 0:
        uid = make_id()
        assert len(uid) == _lib.OH_COMM_ID_BYTES
        _private_dir(path)
        me = os.getpid()
        rec = uid + _REC.pack(me, _proc_start(me) or 0)
        tmp = f"{path}.{me}.tmp"
        for stale in (tmp, path):  # whatever an earlier job left behind goes first
            try:
                os.unlink(stale)
            except FileNotFoundError:
                pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)  # never through somebody else's pre-created file
        with os.fdopen(fd, "wb") as fh:
            fh.write(rec)
        os.replace(tmp, path)
        return uid
    t0 = time.monotonic()
    seen_stale = False
    want = _lib.OH_COMM_ID_BYTES + _REC.size
    checked = False
    while True:
        try:
            if not checked and os.path.isdir(os.path.dirname(path) or "."):  # once, as soon as the directory exists (not on every poll)
                _check_private(os.path.dirname(path) or ".")
                checked = True
            with open(path, "rb") as fh:
                if os.fstat(fh.fileno()).st_uid != os.getuid():
                    raise RuntimeError(f"rendezvous record {path} belongs to another user")
                rec = fh.read()
            if len(rec) == want:
                pid, start = _REC.unpack(rec[_lib.OH_COMM_ID_BYTES :])
                if _proc_start(pid) == start and start != 0:
                    return rec[: _lib.OH_COMM_ID_BYTES]
                seen_stale = True  # the record of a process that is gone: an earlier job's, keep waiting for this one's
        except FileNotFoundError:
            pass
        if time.monotonic() - t0 > timeout:
            why = "only the record of a dead process (an earlier job?)" if seen_stale else "no record"
            raise TimeoutError(f"rank {rank}: {why} at {path} after {timeout:.0f} s -- is rank 0 running, and does it share this directory AND this "
                               f"PID namespace (the liveness test reads /proc/<pid> of the publisher: ranks in different containers never match)? "
                               f"OPTAS_RDZV=tcp exchanges the id over MASTER_ADDR instead")
        time.sleep(0.01)


def tcp_port(k: int = 0) -> int:
    return int(os.environ.get("OPTAS_RDZV_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1 + k))


def exchange_unique_id_tcp(rank: int, world: int, make_id: Callable[[], bytes], addr: Optional[str] = None, port: Optional[int] = None,
                           timeout: float = 300.0, tag: Optional[str] = None) -> bytes:
    """TCP carrier: rank 0 serves the id to world - 1 clients that present this launch's tag (a stranger on the port is turned away)."""
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1") if addr is None else addr
    tagb = (launch_tag() if tag is None else tag).encode()[:64].ljust(64, b"\0")
    if rank == 0:
        uid = make_id()
        assert len(uid) == _lib.OH_COMM_ID_BYTES
        srv, err = None, None
        for k in range(16 if port is None else 1):  # the first free port above the launcher's
            try:
                srv = socket.create_server((addr, tcp_port(k) if port is None else port), reuse_port=False)
                break
            except OSError as e:
                err = e
        if srv is None:
            raise RuntimeError(f"rank 0: cannot listen on {addr}:{tcp_port() if port is None else port}.. for the id exchange ({err})")
        srv.settimeout(1.0)
        served, t0 = 0, time.monotonic()
        with srv:
            while served < world - 1:
                if time.monotonic() - t0 > timeout:
                    raise TimeoutError(f"rank 0: {world - 1 - served} of {world - 1} ranks never asked for the RCCL id within {timeout:.0f} s")
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                with conn:
                    conn.settimeout(5.0)
                    try:
                        got = conn.recv(64, socket.MSG_WAITALL)
                        if got == tagb:
                            conn.sendall(uid)
                            served += 1
                        else:
                            conn.sendall(b"\0")  # not of this launch
                    except OSError:
                        pass
        return uid
    t0 = time.monotonic()
    k = 0
    while True:
        try:
            with socket.create_connection((addr, tcp_port(k) if port is None else port), timeout=2.0) as c:
                c.sendall(tagb)
                uid = b""
                while len(uid) < _lib.OH_COMM_ID_BYTES:
                    part = c.recv(_lib.OH_COMM_ID_BYTES - len(uid))
                    if not part:
                        break
                    uid += part
                if len(uid) == _lib.OH_COMM_ID_BYTES:
                    return uid
        except OSError:
            pass
        k = (k + 1) % 16 if port is None else 0  # rank 0 may have had to move up a port
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: nobody serves the RCCL id of launch {launch_tag()} at {addr}:{tcp_port()}.. after {timeout:.0f} s (is rank 0 running?)")
        time.sleep(0.02)


def exchange(rank: int, world: int, make_id: Callable[[], bytes], path: Optional[str] = None, timeout: float = 300.0) -> bytes:
    """The carrier OPTAS_RDZV selects."""
    mode = os.environ.get("OPTAS_RDZV", "file")
    if mode == "tcp":
        return exchange_unique_id_tcp(rank, world, make_id, timeout=timeout)
    if mode != "file":
        raise ValueError(f"OPTAS_RDZV={mode!r}: expected 'file' or 'tcp'")
    return exchange_unique_id(rank, world, make_id, path, timeout)


class Communicator:
    """The process's RCCL communicator inside liboptas_hip."""

    def __init__(self, rank: int, world: int, local_rank: int, path: Optional[str] = None):
        os.environ.setdefault("NCCL_DEBUG", "WARN")  # RCCL's own diagnosis (a peer that died, a link that is down) reaches stderr
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")  # ... and not stdout, where harnesses print their one result line
        lib = _lib.load()
        self.rank, self.world, self._path = rank, world, (rendezvous_path() if path is None else path)
        _lib.check(lib.oh_set_device(local_rank), "oh_set_device")

        def make_id() -> bytes:
            buf = C.create_string_buffer(_lib.OH_COMM_ID_BYTES)
            _lib.check(lib.oh_comm_unique_id(buf), "oh_comm_unique_id")
            return buf.raw

        uid = exchange(rank, world, make_id, self._path)
        _lib.check(lib.oh_comm_init(rank, world, C.create_string_buffer(uid, _lib.OH_COMM_ID_BYTES)), "oh_comm_init")
        self.barrier()
        if rank == 0:  # everyone holds a communicator: the file has done its job
            try:
                os.remove(self._path)
            except OSError:
                pass

    def info(self) -> Tuple[int, int]:
        """(rank, world) as RCCL reports them (ncclCommUserRank / ncclCommCount)."""
        r, w = C.c_int(-1), C.c_int(0)
        _lib.check(_lib.load().oh_comm_info(C.byref(r), C.byref(w)), "oh_comm_info")
        return r.value, w.value

    def broadcast_constants(self, handle, root: int = 0) -> None:
        """One ncclBroadcast of the oh_chain block; non-root ranks then hold the root's constants (validated before the handle adopts them)."""
        _lib.check(_lib.load().oh_comm_broadcast_constants(handle, int(root)), "oh_comm_broadcast_constants")

    def barrier(self) -> None:
        _lib.check(_lib.load().oh_comm_barrier(), "oh_comm_barrier")

    def max_over_ranks(self, value: float) -> float:
        v = C.c_double(float(value))
        _lib.check(_lib.load().oh_comm_allreduce_max(C.byref(v)), "oh_comm_allreduce_max")
        return v.value

    def sum_over_ranks(self, value: float) -> float:
        v = C.c_double(float(value))
        _lib.check(_lib.load().oh_comm_allreduce_sum(C.byref(v)), "oh_comm_allreduce_sum")
        return v.value

    def allgather(self, d_send, d_recv, nbytes: int) -> None:
        """ncclAllGather of `nbytes` bytes per rank between device buffers (optas_amd._lib.DeviceBuffer): results of the shards, after the solves."""
        _lib.check(_lib.load().oh_comm_allgather(d_send.ptr, d_recv.ptr, C.c_size_t(int(nbytes))), "oh_comm_allgather")

    def destroy(self) -> None:
        _lib.check(_lib.load().oh_comm_destroy(), "oh_comm_destroy")


def init_from_env() -> Optional[Communicator]:
    """Communicator for a launcher-started process (RANK / LOCAL_RANK / WORLD_SIZE in the environment); None for a plain single process."""
    if "RANK" not in os.environ:
        return None
    return Communicator(int(os.environ["RANK"]), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))
