"""Partitioning of denseflow work across the GPUs of one node.

A flow depends only on its two frames and videos are independent, so the path shards with no
exchange step: every rank (one process, one GPU, one `FlowEngine`) works on its own part and the
only cross-rank traffic is the final throughput aggregation.  Two levels:

* ``shard_videos``  - a ``videolist.txt`` (BASELINE config 4): videos are dealt round-robin, which
  balances clips of similar length and keeps a per-video ``.done`` marker meaningful.
* ``shard_pairs``   - one long clip (configs 2/3/5): the M = max(N-|step|,0) flows are split into
  contiguous ranges; a rank needs the frames of its range plus ``|step|`` overlap frames, exactly
  like the reference's own batch padding (src/denseflow_gpu.cpp:204-207), and ``base_start`` keeps
  the output indices global.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence


@dataclass(frozen=True)
class PairShard:
    rank: int
    flow_begin: int   # first global flow index of this rank (the FlowBuffer's base_start)
    flow_end: int     # one past the last
    frame_begin: int  # first frame index the rank must load
    frame_end: int    # one past the last

    @property
    def n_flows(self) -> int:
        return self.flow_end - self.flow_begin

    @property
    def n_frames(self) -> int:
        return self.frame_end - self.frame_begin


def shard_pairs(n_frames: int, step: int, world: int, rank: int) -> PairShard:
    """Contiguous, balanced split of the flows of one clip.  Flow i uses frames i and i+|step|
    (whichever direction), so the frame range is [flow_begin, flow_end + |step|)."""
    if step == 0:
        raise ValueError("step must be non-zero for flow extraction")
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    a = abs(step)
    m = max(n_frames - a, 0)
    base, extra = divmod(m, world)
    begin = rank * base + min(rank, extra)
    end = begin + base + (1 if rank < extra else 0)
    if end == begin:
        return PairShard(rank, begin, end, begin, begin)
    return PairShard(rank, begin, end, begin, end + a)


def shard_videos(videos: Sequence, world: int, rank: int) -> List:
    """Round-robin by line of the list file."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return [v for i, v in enumerate(videos) if i % world == rank]


def aggregate_throughput(pairs_per_rank: Sequence[int], seconds_per_rank: Sequence[float]) -> float:
    """Whole-job frame-pairs/s: all pairs divided by the slowest rank's time (what bench.py reports)."""
    return float(sum(pairs_per_rank)) / max(max(seconds_per_rank), 1e-12)
