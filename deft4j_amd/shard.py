"""Multi-GPU sharding of independent DEFLATE streams.

The reference iterates streams sequentially with no shared state (K/DeflateFilesContainer.java:22);
here stream i goes to one rank (one process per GPU), every rank runs the whole hot path on its
shard with no data-path collective, and the only exchange is the final gather of
(status, saved_bits, out_len) and of the variable-length outputs to rank 0 — RCCL over xGMI when the
process group is `nccl`, `gloo` in the CPU tests.
"""
import torch


def lpt_partition(sizes, world):
    """Longest-processing-time-first: streams sorted by compressed size, each to the least-loaded rank.
    Returns a list (per rank) of stream indices, each in increasing index order."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    load = [0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += sizes[i]
    return [sorted(s) for s in shards]


def optimise_sharded(streams, merge_blocks, make_batch, dist=None, device="cpu"):
    """Every rank calls this with the same `streams`.  `make_batch(list_of_bytes)` returns an object with
    run(merge) / result(i) / output(i) / close() (deft4j_amd.Batch).  Rank 0 gets
    (total_saved, outputs, per_stream_saved); other ranks get None."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    shards = lpt_partition([len(s) for s in streams], world)
    mine = shards[rank]
    meta = torch.zeros((len(streams), 3), dtype=torch.int64)
    outs = {}
    if mine:
        b = make_batch([streams[i] for i in mine]).run(merge_blocks)
        for k, i in enumerate(mine):
            r = b.result(k)
            o = b.output(k) if r["status"] >= 0 else streams[i]
            outs[i] = o
            meta[i, 0] = r["status"]
            meta[i, 1] = r["saved_bits"] if r["status"] >= 0 else 0
            meta[i, 2] = len(o)
        b.close()
    if dist is None or world == 1:
        return int(meta[:, 1].sum()), [outs[i] for i in range(len(streams))], meta[:, 1].tolist()
    meta = meta.to(device)
    dist.all_reduce(meta, op=dist.ReduceOp.SUM)          # each row is written by exactly one rank
    meta = meta.cpu()
    # variable-length outputs: pad every rank's concatenation to the longest and all-gather
    blob = b"".join(outs[i] for i in mine)
    lens = torch.tensor([len(blob)], dtype=torch.int64, device=device)
    dist.all_reduce(lens, op=dist.ReduceOp.MAX)
    cap = max(1, int(lens.item()))
    buf = torch.zeros(cap, dtype=torch.uint8)
    if blob:
        buf[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
    buf = buf.to(device)
    gathered = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    if rank != 0:
        return None
    result = [None] * len(streams)
    for r in range(world):
        data = gathered[r].cpu().numpy().tobytes()
        off = 0
        for i in shards[r]:
            n = int(meta[i, 2])
            result[i] = data[off:off + n]
            off += n
    return int(meta[:, 1].sum()), result, meta[:, 1].tolist()
