"""N > 1 path on CPU: world-size-2 gloo processes exercise the sharding (denseflow_amd/shard.py) and
the aggregation bench.py performs (barrier, MAX over ranks).  No collective is on the data path, so
correctness means: shards are disjoint, cover every flow exactly once, each rank's frame range is
sufficient for its flows, and the aggregate equals total pairs / slowest rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from denseflow_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, step, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = shard.shard_pairs(n_frames, step, world, rank)
    # every rank "computes" a signature per flow from the two frame indices it would read
    flows = torch.full((max(n_frames - abs(step), 0),), -1, dtype=torch.int64)
    for i in range(sh.flow_begin, sh.flow_end):
        a = i if step > 0 else i - step
        b = i + step if step > 0 else i
        assert sh.frame_begin <= min(a, b) and max(a, b) < sh.frame_end, "shard lacks a frame it needs"
        flows[i] = a * 100003 + b
    # gather coverage: each flow index must be produced by exactly one rank
    cover = (flows >= 0).to(torch.int64)
    dist.all_reduce(cover, op=dist.ReduceOp.SUM)
    merged = flows.clone()
    dist.all_reduce(merged, op=dist.ReduceOp.MAX)
    # bench.py's timing reduction: barrier + MAX over ranks
    dist.barrier()
    t = torch.tensor([0.5 + 0.25 * rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([sh.n_flows], dtype=torch.int64)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.savez(os.path.join(out_dir, "result.npz"), cover=cover.numpy(), merged=merged.numpy(), t=t.numpy(),
                 n=n.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,step", [(300, 1), (17, -2), (5, 3), (2, 5)])
def test_pair_sharding_world2_gloo(tmp_path, n_frames, step):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_frames, step, str(tmp_path)), nprocs=world, join=True)
    r = np.load(tmp_path / "result.npz")
    m = max(n_frames - abs(step), 0)
    assert r["cover"].shape == (m,)
    assert np.all(r["cover"] == 1), "a flow was computed twice or not at all"
    expect = np.array([(i if step > 0 else i - step) * 100003 + (i + step if step > 0 else i) for i in range(m)])
    assert np.array_equal(r["merged"], expect)
    assert int(r["n"][0]) == m
    assert float(r["t"][0]) == 0.75  # MAX over ranks
    assert shard.aggregate_throughput([m // 2, m - m // 2], [0.5, 0.75]) == pytest.approx(m / 0.75)


def test_shard_properties_exhaustive():
    for world in (1, 2, 3, 4, 8):
        for n_frames in range(0, 40):
            for step in (1, 2, -1, -3):
                shards = [shard.shard_pairs(n_frames, step, world, r) for r in range(world)]
                m = max(n_frames - abs(step), 0)
                assert sum(s.n_flows for s in shards) == m
                pos = 0
                for s in shards:
                    assert s.flow_begin == pos
                    pos = s.flow_end
                    if s.n_flows:
                        assert s.n_frames == s.n_flows + abs(step) and s.frame_end <= n_frames
                sizes = [s.n_flows for s in shards]
                assert max(sizes) - min(sizes) <= 1


def test_video_sharding_round_robin():
    vids = [f"v{i}.mp4" for i in range(11)]
    parts = [shard.shard_videos(vids, 4, r) for r in range(4)]
    assert sorted(sum(parts, [])) == sorted(vids)
    assert parts[0] == ["v0.mp4", "v4.mp4", "v8.mp4"]
    with pytest.raises(ValueError):
        shard.shard_videos(vids, 2, 2)
