"""Multi-GPU sharding of independent DEFLATE streams.

The reference iterates streams sequentially with no shared state (K/DeflateFilesContainer.java:22); here stream i goes
to one rank (one process per GPU), every rank runs the whole hot path on its shard with no data-path collective, and the
only exchange is at the end: the per-stream (status, saved_bits, out_len) table is combined on every rank and the
variable-length outputs travel to rank 0 only — grouped point-to-point sends, so over xGMI rank 0 receives on its seven
links concurrently.  RCCL when the process group is `nccl`, `gloo` in the CPU tests.

A rank only ever materialises the streams of its own shard: callers pass the sizes of all streams (what the LPT
partition needs) and a `load(i)` function.
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


def _chunks(idx, sizes, budget):
    """split a shard into sub-batches of at most `budget` compressed bytes (device memory stays bounded)"""
    cur, tot = [], 0
    for i in idx:
        if cur and tot + sizes[i] > budget:
            yield cur
            cur, tot = [], 0
        cur.append(i)
        tot += sizes[i]
    if cur:
        yield cur


def optimise_sharded(sizes, load, merge_blocks, make_batch, dist=None, device="cpu", batch_bytes=64 << 20, run=None):
    """Every rank calls this with the same `sizes`; `load(i)` returns stream i's bytes and is only called for streams of
    the caller's shard.  `make_batch(list_of_bytes)` returns an object with run(merge) / result(i) / output(i) / close()
    (deft4j_amd.Batch); `run(batch)` overrides the default batch.run(merge_blocks) (e.g. a recompress mode).
    Rank 0 gets (total_saved, outputs, per_stream_saved) with outputs[i] = the new bytes or None (unchanged / did not
    parse: the caller keeps its original); other ranks get None.  saved = what DeflateStream.optimise saved on the
    stream plus, in a recompress run, what the grafted recompressed stream saved on top (M/CMDUtil.java:95-103)."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    n = len(sizes)
    shards = lpt_partition(list(sizes), world)
    mine = shards[rank]
    meta = torch.zeros((n, 3), dtype=torch.int64)   # status, saved_bits, out_len (0: keep the original)
    outs = {}
    for sub in _chunks(mine, sizes, batch_bytes):
        b = make_batch([load(i) for i in sub])
        if run is not None:
            run(b)
        else:
            b.run(merge_blocks)
        for k, i in enumerate(sub):
            r = b.result(k)
            meta[i, 0] = r["status"]
            if r["status"] == 0:
                o = b.output(k)
                outs[i] = o
                meta[i, 1] = r["saved_bits"]
                if run is not None and hasattr(b, "recompress_result"):
                    try:
                        grafted, rsaved = b.recompress_result(k)
                        if grafted:
                            meta[i, 1] += rsaved
                    except Exception:  # noqa: BLE001 — the batch was not run in a recompress mode
                        pass
                meta[i, 2] = len(o)
        b.close()
    if dist is None or world == 1:
        return int(meta[:, 1].sum()), [outs.get(i) for i in range(n)], meta[:, 1].tolist()
    # (status, saved_bits, out_len) of every stream on every rank: each row is written by exactly one rank
    meta = meta.to(device)
    dist.all_reduce(meta, op=dist.ReduceOp.SUM)
    meta = meta.cpu()
    # variable-length outputs -> rank 0 only, one message per rank, all in flight together
    if rank != 0:
        blob = b"".join(outs[i] for i in mine if i in outs)
        if blob:
            t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, 0)]):
                w.wait()
        return None
    bufs, ops = {}, []
    for r in range(1, world):
        nbytes = int(sum(int(meta[i, 2]) for i in shards[r]))
        if nbytes:
            bufs[r] = torch.empty(nbytes, dtype=torch.uint8, device=device)
            ops.append(dist.P2POp(dist.irecv, bufs[r], r))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    result = [outs.get(i) for i in range(n)]
    for r, buf in bufs.items():
        data = buf.cpu().numpy().tobytes()
        off = 0
        for i in shards[r]:
            ln = int(meta[i, 2])
            if ln:
                result[i] = data[off:off + ln]
                off += ln
    return int(meta[:, 1].sum()), result, meta[:, 1].tolist()
