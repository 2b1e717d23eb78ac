"""The N>1 host path on CPU: two gloo ranks shard the filters and reduce the timed seconds (max over ranks)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.shard(2048, world, rank)
    dist.barrier()
    secs = bench.reduce_times(1.0 + rank, world)   # rank 1 is the slow one
    # the run's one collective: {steps, seconds, bytes, max_rel_err} -> {sum, max, sum, max}
    rec = bench.reduce_record((hi - lo) * 10, 1.0 + rank, 1000.0 * (rank + 1), 1e-12 * (rank + 1), world)
    q.put((rank, lo, hi, secs, rec))
    dist.destroy_process_group()


def test_two_rank_shard_and_time_reduction():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:3] == (0, 1024) and res[1][1:3] == (1024, 2048)     # contiguous, disjoint, complete
    assert res[0][3] == 2.0 and res[1][3] == 2.0                        # max over ranks on every rank
    for r in res:                                                       # the four-field record, identical on every rank
        assert r[4] == {"steps": 20480.0, "seconds": 2.0, "bytes": 3000.0, "max_rel_err": 2e-12}


def test_single_rank_record_is_local():
    import bench
    assert bench.reduce_record(5, 0.5, 7.0, 1e-13, 1) == {"steps": 5.0, "seconds": 0.5, "bytes": 7.0, "max_rel_err": 1e-13}


def test_shard_partitions_exactly():
    import bench
    for total, world in [(1024, 1), (8192, 8), (1000, 3), (7, 8)]:
        parts = [bench.shard(total, world, r) for r in range(world)]
        assert parts[0][0] == 0 and parts[-1][1] == total
        for a, b in zip(parts, parts[1:]):
            assert a[1] == b[0]


def test_scene_is_deterministic_and_ordered():
    from vi_ekf_amd import scene
    a, b = scene.make_scene(3, 5, 2, seed=9), scene.make_scene(3, 5, 2, seed=9)
    for k in ("pix", "u", "z", "slot"):
        np.testing.assert_array_equal(a[k], b[k])
    assert (a["slot"][0] == np.arange(4, -1, -1)).all()        # reverse in-frame order (vi_ekf_meas.cpp:150-176)
    assert scene.algorithmic_bytes_per_step(50) == 446424
    assert scene.algorithmic_bytes_per_step(150) == 3490424
