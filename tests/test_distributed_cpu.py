"""World-size-2 gloo run of the multi-GPU plumbing bench.py uses (optas_amd/distributed.py): the one
broadcast of the kinematic constants, instance sharding, MAX/SUM reductions of the timing.  No GPU."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from conftest import KUKA_KIN


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kin, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np

    import bench
    from optas_amd import _lib
    from optas_amd import distributed as oad
    from optas_amd.models import RobotModel

    dist = oad.init_process_group("gloo")
    # only rank 0 knows the robot; the others receive the folded constants over the wire
    chain = RobotModel(urdf_filename=kin).kinematic_chain("end_effector_ball") if rank == 0 else _lib.oh_chain()
    buf, got = oad.broadcast_chain(chain, "cpu", src=0)
    lo, hi = oad.shard(1001, world, rank)
    x0, qc = bench.make_inputs(8, rank)
    tmax = oad.max_over_ranks(1.0 + rank, "cpu")
    tsum = oad.sum_over_ranks(float(hi - lo), "cpu")
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), chain=np.frombuffer(bytes(got), dtype=np.uint8), lo=lo, hi=hi, qc=qc, tmax=tmax, tsum=tsum,
             nbytes=buf.numel())
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world_size_2(tmp_path):
    import torch.multiprocessing as mp

    from optas_amd import _lib
    from optas_amd.models import RobotModel

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, KUKA_KIN, str(tmp_path)), nprocs=world, join=True)
    ref = bytes(RobotModel(urdf_filename=KUKA_KIN).kinematic_chain("end_effector_ball"))
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(world)]
    for k in range(world):
        assert r[k]["chain"].tobytes() == ref  # every rank holds rank 0's constants, bit for bit
        assert int(r[k]["nbytes"]) == C.sizeof(_lib.oh_chain) == 2952
        assert float(r[k]["tmax"]) == 2.0 and float(r[k]["tsum"]) == 1001.0
    assert (int(r[0]["lo"]), int(r[0]["hi"]), int(r[1]["lo"]), int(r[1]["hi"])) == (0, 501, 501, 1001)
    assert not np.array_equal(r[0]["qc"], r[1]["qc"])  # ranks draw different instances


def test_shard_partitions_exactly():
    from optas_amd.distributed import shard

    for n in (0, 1, 7, 1024, 1001):
        for w in (1, 2, 3, 8):
            parts = [shard(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
