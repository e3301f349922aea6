"""Multi-GPU plumbing for one-process-per-GPU runs (bench.py, user scripts).  No torch: the RCCL communicator lives inside
liboptas_hip (``oh_comm_*``, include/optas_hip.h).

MPC instances are independent, so the data path needs no collective at all: each rank solves its own shard.  The only exchange is ONE
broadcast of the kinematic constants (``oh_chain``, 2952 bytes) from rank 0 over RCCL/xGMI.  What Python does is the rendezvous: rank 0
asks the library for an RCCL unique id and hands the 128 bytes to the other ranks through a file (one node: the launcher's workers share
a filesystem); then every rank calls ``oh_comm_init``.  The launcher contract is the usual one: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR,
MASTER_PORT in the environment (``python -m torch.distributed.run ...`` sets them; the workers themselves never load it).
"""
from __future__ import annotations

import ctypes as C
import os
import tempfile
import time
from typing import Callable, Optional, Tuple

from . import _lib


def shard(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous partition [lo, hi) of n_total instances for this rank."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rendezvous_path(tag: Optional[str] = None) -> str:
    """File the unique id travels through.  The name is unique per launch: the launcher's port plus the launcher's process id (every
    worker of one launch has the same parent), so a stale file of an earlier job can never be read."""
    d = os.environ.get("OPTAS_RDZV_DIR", tempfile.gettempdir())
    if tag is None:
        tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
    return os.path.join(d, f"optas_amd_rdzv_{tag}.id")


def exchange_unique_id(rank: int, world: int, make_id: Callable[[], bytes], path: Optional[str] = None, timeout: float = 300.0) -> bytes:
    """Rank 0 creates the id and publishes it atomically (write + rename); the others wait for the file."""
    path = rendezvous_path() if path is None else path
    if rank == 0:
        uid = make_id()
        assert len(uid) == _lib.OH_COMM_ID_BYTES
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as fh:
            fh.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.monotonic()
    while True:
        try:
            with open(path, "rb") as fh:
                uid = fh.read()
            if len(uid) == _lib.OH_COMM_ID_BYTES:
                return uid
        except FileNotFoundError:
            pass
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no RCCL unique id at {path} after {timeout:.0f} s (is rank 0 running?)")
        time.sleep(0.01)


class Communicator:
    """The process's RCCL communicator inside liboptas_hip."""

    def __init__(self, rank: int, world: int, local_rank: int, path: Optional[str] = None):
        lib = _lib.load()
        self.rank, self.world, self._path = rank, world, (rendezvous_path() if path is None else path)
        _lib.check(lib.oh_set_device(local_rank), "oh_set_device")

        def make_id() -> bytes:
            buf = C.create_string_buffer(_lib.OH_COMM_ID_BYTES)
            _lib.check(lib.oh_comm_unique_id(buf), "oh_comm_unique_id")
            return buf.raw

        uid = exchange_unique_id(rank, world, make_id, self._path)
        _lib.check(lib.oh_comm_init(rank, world, C.create_string_buffer(uid, _lib.OH_COMM_ID_BYTES)), "oh_comm_init")
        self.barrier()
        if rank == 0:  # everyone holds a communicator: the file has done its job
            try:
                os.remove(self._path)
            except OSError:
                pass

    def broadcast_constants(self, handle, root: int = 0) -> None:
        """One ncclBroadcast of the oh_chain block, in place in the handle's device buffer; non-root ranks then hold the root's constants."""
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

    def destroy(self) -> None:
        _lib.check(_lib.load().oh_comm_destroy(), "oh_comm_destroy")


def init_from_env() -> Optional[Communicator]:
    """Communicator for a launcher-started process (RANK / LOCAL_RANK / WORLD_SIZE in the environment); None for a plain single process."""
    if "RANK" not in os.environ:
        return None
    return Communicator(int(os.environ["RANK"]), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))
