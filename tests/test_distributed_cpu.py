"""World-size-2 run of the multi-GPU plumbing bench.py uses (optas_amd/distributed.py) on CPU: the rendezvous that carries the RCCL
unique id from rank 0 to the others (two real processes, a file in a temporary directory), instance sharding, per-rank inputs.  The
RCCL calls themselves live in liboptas_hip (oh_comm_*) and need GPUs: here they must fail loudly, never fall back."""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest


def _worker(rank, world, path, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import time

    import numpy as np

    import bench
    from optas_amd import _lib
    from optas_amd import distributed as oad

    if rank == 0:
        time.sleep(0.3)  # the others are already polling
    made = []

    def make_id():  # stands in for oh_comm_unique_id (needs a GPU): 128 bytes only rank 0 can know
        made.append(1)
        return bytes(np.random.default_rng(os.getpid()).integers(0, 256, _lib.OH_COMM_ID_BYTES, dtype=np.uint8))

    uid = oad.exchange_unique_id(rank, world, make_id, path=path, timeout=60.0)
    lo, hi = oad.shard(1001, world, rank)
    x0, qc = bench.make_inputs(8, rank)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), uid=np.frombuffer(uid, dtype=np.uint8), made=len(made), lo=lo, hi=hi, qc=qc)


def test_rendezvous_world_size_2(tmp_path):
    world = 2
    path = str(tmp_path / "rdzv.id")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, path, str(tmp_path))) for r in range(world)]
    for p in procs[::-1]:  # rank 1 first: it has to wait for the file
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(world)]
    assert r[0]["uid"].tobytes() == r[1]["uid"].tobytes() and len(r[0]["uid"]) == 128  # every rank holds rank 0's id, bit for bit
    assert int(r[0]["made"]) == 1 and int(r[1]["made"]) == 0  # only rank 0 creates one
    assert (int(r[0]["lo"]), int(r[0]["hi"]), int(r[1]["lo"]), int(r[1]["hi"])) == (0, 501, 501, 1001)
    assert not np.array_equal(r[0]["qc"], r[1]["qc"])  # ranks draw different instances
    assert not os.path.exists(path + ".tmp")


def test_rendezvous_path_is_unique_per_launch(monkeypatch, tmp_path):
    from optas_amd import distributed as oad

    monkeypatch.setenv("OPTAS_RDZV_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "29500")
    a = oad.rendezvous_path()
    monkeypatch.setenv("MASTER_PORT", "29501")
    b = oad.rendezvous_path()
    assert a != b and a.startswith(str(tmp_path)) and str(os.getppid()) in a
    with pytest.raises(TimeoutError):
        oad.exchange_unique_id(1, 2, lambda: b"", path=str(tmp_path / "never.id"), timeout=0.2)


def test_communicator_needs_a_gpu_and_says_so():
    """No CPU path: without a device the communicator entry points return an error code and a message."""
    from optas_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible: the failure path is not reachable")
    lib = _lib.load()
    uid = C.create_string_buffer(_lib.OH_COMM_ID_BYTES)
    assert lib.oh_comm_init(0, 2, uid) == _lib.OH_ERR_HIP and b"no HIP device" in lib.oh_last_error()
    v = C.c_double(1.0)
    assert lib.oh_comm_allreduce_max(C.byref(v)) == _lib.OH_ERR_STATE
    assert lib.oh_comm_barrier() == _lib.OH_ERR_STATE
    assert lib.oh_comm_init(2, 2, uid) == _lib.OH_ERR_INVALID


def test_no_torch_in_the_product_or_the_bench():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    files = [os.path.join(root, "bench.py")] + [os.path.join(root, "optas_amd", f) for f in os.listdir(os.path.join(root, "optas_amd")) if f.endswith(".py")]
    for f in files:
        src = open(f).read()
        assert "import torch" not in src and "from torch" not in src, f


def test_shard_partitions_exactly():
    from optas_amd.distributed import shard

    for n in (0, 1, 7, 1024, 1001):
        for w in (1, 2, 3, 8):
            parts = [shard(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
