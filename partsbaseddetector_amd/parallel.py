"""Multi-GPU sharding of detect() — one process per GPU, no data-path collective.

The path shards two ways, both embarrassingly parallel:
  * frames  -> ranks        (configs[2]: batch of frames, frame f on rank f % world)
  * pyramid levels -> ranks (configs[3]: one large frame; a cost-balanced subset of levels per
                             rank via pbd_options.level_begin/level_end or an explicit LPT split)
The only exchange is the gather of the (tiny, fixed-capacity) candidate buffers to rank 0:
`torch.distributed.all_gather` — RCCL over xGMI when the backend is "nccl", gloo on CPU tests.
Payload is KB-scale, so it is latency- not bandwidth-bound; no all-reduce is ever needed.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

HEAD_WORDS = 4


def shard_frames(nframes: int, world: int, rank: int) -> List[int]:
    """Frame f -> rank f % world."""
    return [f for f in range(nframes) if f % world == rank]


def shard_levels_contiguous(cells: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Split levels 0..n-1 into `world` contiguous [begin, end) ranges with balanced cell counts
    (cost of a level is proportional to its cells).  Contiguous ranges map directly onto
    pbd_options.level_begin/level_end.  Level 0 alone can exceed 1/world of the work; ranges may
    then be empty for trailing ranks."""
    n = len(cells)
    total = float(sum(cells))
    out, b, acc = [], 0, 0.0
    for r in range(world):
        if r == world - 1:
            e = n
        else:
            target = total * (r + 1) / world
            e = b
            while e < n and acc + cells[e] / 2.0 <= target:  # take a level if its midpoint lies before the cut
                acc += cells[e]
                e += 1
        out.append((b, e))
        b = e
    return out


def shard_levels_lpt(cells: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time greedy bin packing of levels onto `world` ranks (cost of a level is
    proportional to its cells): levels in decreasing cost, each to the least loaded rank.
    Returns one sorted level list per rank for pbd_set_levels / Handle.set_levels; the makespan is within
    4/3 of optimal and never below max(level 0, total/world)."""
    loads = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for l in sorted(range(len(cells)), key=lambda i: (-cells[i], i)):
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(l)
        loads[r] += cells[l]
    return [sorted(x) for x in out]


def pack_candidates(cands, max_parts: int, capacity: int) -> np.ndarray:
    """(heads, boxes, locs) -> int32 [1 + capacity * (4 + 7*max_parts)] : count, then records."""
    heads, boxes, locs = cands
    n = len(heads)
    if n > capacity:     # the C ABI reports PBD_ERR_CAPACITY for the same condition: never drop detections silently
        raise OverflowError(f"{n} candidates do not fit the gather capacity {capacity}: pass capacity >= the handle's "
                            f"max_candidates")
    rec = HEAD_WORDS + 7 * max_parts
    buf = np.zeros(1 + capacity * rec, np.int32)
    buf[0] = n
    body = buf[1:].reshape(capacity, rec)
    if n:
        body[:n, 0] = heads["score"][:n].view(np.int32)
        body[:n, 1] = heads["component"][:n]
        body[:n, 2] = heads["level"][:n]
        body[:n, 3] = heads["nparts"][:n]
        body[:n, 4:4 + 4 * max_parts] = boxes[:n].reshape(n, -1)
        body[:n, 4 + 4 * max_parts:] = locs[:n].reshape(n, -1)
    return buf


def unpack_candidates(buf: np.ndarray, max_parts: int):
    from .capi import HEAD_DTYPE
    rec = HEAD_WORDS + 7 * max_parts
    n = int(buf[0])
    body = buf[1:].reshape(-1, rec)[:n]
    heads = np.zeros(n, HEAD_DTYPE)
    heads["score"] = body[:, 0].copy().view(np.float32)
    heads["component"], heads["level"], heads["nparts"] = body[:, 1], body[:, 2], body[:, 3]
    boxes = body[:, 4:4 + 4 * max_parts].reshape(n, max_parts, 4).copy()
    locs = body[:, 4 + 4 * max_parts:].reshape(n, max_parts, 3).copy()
    return heads, boxes, locs


def gather_candidates(cands, max_parts: int, capacity: int = 4096, device=None, dst=None):
    """Gather of every rank's candidates; returns a list (one entry per rank) of (heads, boxes, locs).
    dst=None: every rank gets the list (all_gather); dst=r: rank r only (the others return None) — what a detector
    host needs, and 1/world of the traffic.  Works with any initialised torch.distributed backend (nccl = RCCL over
    xGMI with device tensors, gloo on the host).  Two small collectives: the ranks' counts (every rank learns the
    longest list), then the records padded to that length — KB-scale payloads whatever `capacity` (the most a rank
    may hold; more raises, like PBD_ERR_CAPACITY) is."""
    import torch
    import torch.distributed as dist

    heads = cands[0]
    n = len(heads)
    if n > capacity:     # the C ABI reports PBD_ERR_CAPACITY for the same condition: never drop detections silently
        raise OverflowError(f"{n} candidates do not fit the gather capacity {capacity}: pass capacity >= the handle's max_candidates")
    world, rank = dist.get_world_size(), dist.get_rank()
    cnt = torch.tensor([n], dtype=torch.int32, device=device)
    counts = torch.empty(world, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(counts, cnt)
    counts = counts.cpu().tolist()
    nmax = max(max(counts), 1)
    buf = torch.from_numpy(pack_candidates(cands, max_parts, nmax))
    if device is not None:
        buf = buf.to(device)
    # one contiguous receive buffer (the per-rank blocks are views of it): ONE device-to-host copy afterwards
    big = torch.empty(world * buf.numel(), dtype=buf.dtype, device=buf.device) if (dst is None or rank == dst) else None
    outs = list(big.chunk(world)) if big is not None else None
    if dst is None:
        dist.all_gather(outs, buf)
    else:
        dist.gather(buf, outs, dst=dst)
        if outs is None:
            return None
    host = big.cpu().numpy().reshape(world, -1)
    return [unpack_candidates(host[r], max_parts) for r in range(world)]


def merge_candidates(per_rank):
    """Concatenate gathered candidates and put them in the order of a single-threaded reference run —
    (level, component, root row, root column), src/DynamicProgram.cpp:197-253 — which is also the order one
    handle produces; then Candidate::sort / NMS on the host as usual."""
    heads = np.concatenate([p[0] for p in per_rank])
    boxes = np.concatenate([p[1] for p in per_rank])
    locs = np.concatenate([p[2] for p in per_rank])
    if len(heads):
        order = np.lexsort((locs[:, 0, 0], locs[:, 0, 1], heads["component"], heads["level"]))
        heads, boxes, locs = heads[order], boxes[order], locs[order]
    return heads, boxes, locs
