#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline: input MB/s on raw-deflate `optimise` (mode NONE).

A step = one pass of the hot path (parse -> candidate search -> bit packing) over one
batch: config 2 of BASELINE.json, a 64 MiB synthetic repetitive-text buffer deflated
with zlib level 9 (== the reference's JavaCompressor) and optimised as ONE raw stream,
merge off (the headline row).  Inputs are resident in HBM when the timed
region starts.  With N GPUs every rank optimises its own independent stream (weak
scaling, no data-path collective; the final size gather is the only exchange).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def _gen_member(seed):
    import synth
    raw = synth.reptext(1 << 20, seed)
    return raw, synth.deflate9(raw)


def bench_config3(args, rank, world, dist, tdev, D, synth, torch):
    """BASELINE config 3: `--members` independent 1 MiB gzip members per GPU, mode CHEAP, merge on (the CLI default,
    M/Optimise.java:33-34).  A step = CMDUtil.optimise's per-stream work for the whole batch (M/CMDUtil.java:70-105):
    optimise every member, recompress its decoded bytes with the six zlib-family compressors (each output optimised),
    re-parse + optimise the winner, graft it where smaller, and recompute the gzip trailer (CRC-32 / ISIZE)."""
    import zlib
    from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
    members = args.members
    seeds = [0xD4F7 + rank * members + i for i in range(members)]
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        gen = list(ex.map(_gen_member, seeds, chunksize=8))
    raws = [g[0] for g in gen]
    streams = [g[1] for g in gen]
    n_in = sum(len(r) for r in raws)
    merge = True

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    total = args.warmup + args.steps

    def step(b):
        b.run_recompress(D.MODE_CHEAP, merge)
        return b.checksums(0)      # trailer values of every member are computed in one device pass

    for _ in range(args.warmup):
        b = D.Batch(streams)
        step(b)
        b.close()
    batches = [D.Batch(streams) for _ in range(args.steps)]   # uploads: inputs resident before timing
    barrier()
    t0 = time.perf_counter()
    for b in batches:
        step(b)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=tdev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    last = batches[-1]
    st = last.stats()
    outs, grafted, saved, rsaved = [], 0, 0, 0
    for i in range(members):
        r = last.result(i)
        g, rs = last.recompress_result(i)
        outs.append(last.output(i) if r["status"] == 0 else streams[i])
        grafted += g
        saved += r["saved_bits"]
        rsaved += rs
    crc0 = last.checksums(0)
    ok = all(zlib.decompress(o, -15) == r for o, r in zip(outs[:32], raws[:32])) and crc0[0] == (zlib.crc32(raws[0]) & 0xffffffff)
    for b in batches:
        b.close()
    if dist is not None:
        mine = torch.tensor([sum(len(s) for s in streams), sum(len(o) for o in outs), saved + rsaved], device=tdev, dtype=torch.int64)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)    # the path's only exchange: sizes to rank 0 (RCCL over xGMI)
    ms_step = elapsed * 1000.0 / args.steps
    value = (n_in * world / 1e6) / (elapsed / args.steps)
    launches = max(1, st["state_launches"])
    n_tok = st["state_tokens_per_round"] / launches
    n_u = st["state_bytes_per_round"] / launches
    alg = int(8 * n_tok + min(n_u, 48 * n_tok))
    dur_s = st["ms_state_kernels"] / 1000.0 / launches
    achieved = alg / dur_s / 1e9 if dur_s > 0 else 0.0
    line = {
        "metric": "input MB/s on gzip-member optimise, mode CHEAP", "value": round(value, 3), "unit": "MB/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%d x 1 MiB synthetic repetitive-text gzip members per GPU (zlib-9), mode=CHEAP, merge=on" % members,
                   "members": members, "grafted": grafted, "saved_bits": saved, "recompress_saved_bits": rsaved,
                   "compressor_outputs_searched": st["recompress_outputs"], "huffman_only_outputs_pruned_by_entropy_bound": st["recompress_outputs_pruned"],
                   "roundtrip_ok": bool(ok), "value_per_gpu": round(value / world, 3)},
        "phases_ms": {"optimise_originals": round(st["ms_total"], 2), "encode_and_search_outputs": round(st["ms_recompress_encode"], 2),
                      "encoder_front_end": round(st["ms_recompress_encode_front"], 2), "search_of_outputs": round(st["ms_recompress_encode_search"], 2),
                      "reoptimise_winners": round(st["ms_recompress_reoptimise"], 2),
                      "lz_sort_kernels": round(st["ms_lz_sort"], 2), "lz_parse_kernels": round(st["ms_lz_parse"], 2), "lz_emit_kernels": round(st["ms_lz_emit"], 2),
                      "lz_parse_passes": st["lz_parse_passes"], "lz_chunks_rerun": st["lz_chunks_rerun"]},
        "roofline": {"bound": "hbm", "kernel": "k_exec_state_ops", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "algorithmic_bytes_per_launch": int(alg),
                     "launches_per_step": launches, "avg_launch_ms": round(dur_s * 1000.0, 4),
                     # SURVEY.md §8d whole-path figure: C_in + U + C_out, plus U + C_out_k per compressor pass
                     "path_algorithmic_bytes": int(st["search_bytes_algorithmic"]),
                     "path_achieved_GBs": round(st["search_bytes_algorithmic"] / (elapsed / args.steps) / 1e9, 3)},
    }
    parity_failed = False
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_compose as OC
        k = min(4, members)
        t0 = time.perf_counter()
        OC.recompress(streams[0], merge)
        c1 = time.perf_counter() - t0
        with ThreadPoolExecutor(max_workers=k) as ex:
            want = list(ex.map(lambda a: OC.recompress(a, merge), streams[:k]))
        identical = all((outs[i] == (w["out"] if w["status"] == 0 else streams[i])) for i, w in enumerate(want))
        line["config"]["output_bytes_identical_to_oracle"] = bool(identical)
        line["config"]["parity_sample"] = "first %d members against the oracle orchestration (zlib-9 / jzlib oracles + optimiser oracle)" % k
        parity_failed = not identical
        line["cpu_baseline"] = {"value": round(len(raws[0]) / 1e6 / c1, 4), "unit": "MB/s", "cores": 1, "kind": "port",
                                "sample": "one member (1 MiB) through the same orchestration over the oracles, %.1f s of CPU" % c1}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if parity_failed or not ok:
        sys.exit("bench.py: the GPU output differs from the oracle's (or does not round-trip)")


def _gen_mixed(i):
    import synth
    return synth.mixed_stream(i)


def bench_config5(args, rank, world, dist, tdev, D, synth, torch):
    """BASELINE config 5: --total-mib of mixed PNG-IDAT-like / gzip-member streams, mode NONE, merge on, sharded over the
    GPUs at stream granularity (deft4j_amd/shard.py: LPT partition, no data-path collective, sizes combined on every rank
    and outputs sent to rank 0 only).  Strong scaling: the job is fixed, every rank generates and holds only its shard."""
    import zlib
    from concurrent.futures import ProcessPoolExecutor
    from deft4j_amd import shard
    specs, tot = [], 0
    while tot < (args.total_mib << 20):
        kind, n = synth.mixed_spec(len(specs))
        specs.append((kind, n))
        tot += n
    sizes = [n for _, n in specs]                   # the partition works on uncompressed sizes (known without generating)
    mine = shard.lpt_partition(sizes, world)[rank]
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        gen = dict(zip(mine, ex.map(_gen_mixed, mine, chunksize=2)))
    streams = {i: g[2] for i, g in gen.items()}

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def step():
        return shard.optimise_sharded(sizes, lambda i: streams[i], True, lambda ss: D.Batch(ss), dist=dist, device=tdev,
                                      batch_bytes=1 << 30)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=tdev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        total_saved, outs, _ = res
        ok = True
        for i in mine[:8]:                          # rank 0 holds the originals of its own shard only
            o = outs[i] if outs[i] is not None else streams[i]
            ok = ok and zlib.decompress(o, -15) == gen[i][1]
        value = (tot / 1e6) / (elapsed / args.steps)
        line = {"metric": "input MB/s on raw-deflate optimise (mode NONE), mixed streams sharded over the GPUs", "value": round(value, 3),
                "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1000.0 / args.steps, 3),
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": "%d MiB of mixed PNG-IDAT-like (1-16 MiB) and 1 MiB text streams, zlib-9, mode=NONE, merge=on, stream-sharded" % args.total_mib,
                           "streams": len(specs), "idat_streams": sum(1 for k, _ in specs if k == "idat"), "saved_bits": int(total_saved),
                           "changed_streams": sum(1 for o in outs if o is not None), "roundtrip_ok": bool(ok), "value_per_gpu": round(value / world, 3)},
                "roofline": None, "cpu_baseline": None}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mib", type=int, default=64, help="uncompressed MiB per stream (64 = BASELINE config 2)")
    ap.add_argument("--merge", type=int, default=0)
    ap.add_argument("--total-mib", type=int, default=8192, help="config5: uncompressed MiB of the whole job (strong scaling)")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config5"],
                    help="config2 (default, BASELINE.json's headline): one 64 MiB stream per GPU, mode NONE; "
                         "config3: --members x 1 MiB gzip members per GPU, mode CHEAP (recompress + graft); "
                         "config5: --total-mib of mixed PNG-IDAT-like / gzip streams sharded over the GPUs, mode NONE")
    ap.add_argument("--members", type=int, default=1024, help="config3: members per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for 1-GPU rehearsals)")
    ap.add_argument("--device", type=int, default=None, help="override the HIP device index (rehearsal: several ranks on one GPU)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Started by hand as `python bench.py --gpus N`: launch the N ranks (one process per GPU) the way the driver
        # does.  Nothing in THIS process has touched the GPU yet, and the ranks are children, not an exec.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dev = local if args.device is None else args.device
        torch.cuda.set_device(dev)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        world = dist.get_world_size()   # n_gpus is what the process group reports, not what the flag asked for

    import deft4j_amd as D
    import synth
    D.init(local if args.device is None else args.device)
    tdev = "cuda" if (dist is None or args.backend == "nccl") else "cpu"

    if args.workload == "config3":
        return bench_config3(args, rank, world, dist, tdev, D, synth, torch)
    if args.workload == "config5":
        return bench_config5(args, rank, world, dist, tdev, D, synth, torch)

    # --- workload: one independent stream per rank (per-file sharding) ---
    n = args.mib << 20
    t0 = time.time()
    raw = synth.reptext(n, 0xD4F7 + rank)
    stream = synth.deflate9(raw)
    gen_s = time.time() - t0

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    total = args.warmup + args.steps
    batches = [D.Batch([stream]) for _ in range(total)]  # uploads: inputs resident before timing
    for i in range(args.warmup):
        batches[i].run(bool(args.merge))
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        batches[i].run(bool(args.merge))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=tdev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    last = batches[-1]
    st = last.stats()
    res = last.result(0)
    # size-independent parity property at full size: the output is a valid stream of the same bytes
    import zlib
    out = last.output(0)
    ok = zlib.decompress(out, -15) == raw and res["status"] in (0, 1)
    sizes = [len(stream), len(out), res["saved_bits"]]
    if dist is not None:
        # the path's only exchange: gather (in_len, out_len, saved_bits) of every shard on rank 0 (RCCL over xGMI)
        mine = torch.tensor(sizes, device=tdev, dtype=torch.int64)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
    for b in batches:
        b.close()

    ms_step = elapsed * 1000.0 / args.steps
    value = (n * world / 1e6) / (elapsed / args.steps)  # whole-job MB/s of uncompressed input
    # Dominant kernel = k_exec_state_ops (profiles/: ~67 % of device time).  One launch runs one dependency level
    # of the search program over every active block: algorithmically one pass over the tokens plus the decoded
    # bytes the literal-cost evaluation touches, 8N + min(U, 48*refs) (SURVEY.md §8d "cost-eval"; refs <= N).
    launches = max(1, st["state_launches"])
    # a launch covers one block group (stream lane) at one level: mean tokens / decoded bytes per launch
    n_tok = st["state_tokens_per_round"] / launches
    n_u = st["state_bytes_per_round"] / launches
    alg = int(8 * n_tok + min(n_u, 48 * n_tok))
    dur_s = st["ms_state_kernels"] / 1000.0 / launches  # HIP events around each launch on the library's stream
    achieved = alg / dur_s / 1e9 if dur_s > 0 else 0.0
    dom = "k_exec_state_ops"
    kern = {dom: st["ms_state_kernels"]}
    # HBM traffic per launch of that kernel: FETCH_SIZE + WRITE_SIZE from the committed rocprofv3 --pmc passes of this
    # same command (separate passes; FETCH_SIZE left uncorrected — the accesses are not the wide coalesced
    # stream the guide's x2 applies to), newest profiles/*_pmc_fetch_write_summary.json
    traffic = None
    try:
        import glob
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_fetch_write_summary.json")))[-1]
        pm = json.load(open(f))
        traffic = int((pm["FETCH_SIZE"][dom]["per_dispatch_MB"] + pm["WRITE_SIZE"][dom]["per_dispatch_MB"]) * 1e6)
    except Exception:  # noqa: BLE001
        traffic = None
    try:
        baseline_metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:  # noqa: BLE001
        baseline_metric = None
    line = {
        "metric": "input MB/s on raw-deflate optimise (mode NONE)",
        "value": round(value, 3), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%d MiB synthetic repetitive text, zlib-9 raw deflate, one stream per GPU, mode=NONE, merge=%s"
                   % (args.mib, "on" if args.merge else "off"),
                   "stream_bytes": len(stream), "blocks": st["n_blocks"], "tokens": st["n_tokens"],
                   "saved_bits": res["saved_bits"], "roundtrip_ok": bool(ok),
                   # BASELINE.json words the metric per GPU and adds the output-bit delta against the Java reference: the
                   # value above is the whole job (== per GPU at N=1); the delta is MEASURED below on the cpu_baseline
                   # sample (GPU output vs the oracle's output for the same stream) and added to this object
                   "baseline_metric": baseline_metric, "value_per_gpu": round(value / world, 3)},
        "phases_ms": {k: round(st[k], 2) for k in ("ms_parse", "ms_optimise", "ms_merge", "ms_write", "ms_total")},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                     "algorithmic_bytes_per_launch": int(alg), "launches_per_step": launches,
                     "avg_launch_ms": round(dur_s * 1000.0, 4), "kernel_ms_per_step": round(kern[dom], 3),
                     "parse_kernels_ms": round(st["ms_parse_kernels"], 3), "search_kernels_ms": round(st["ms_search_kernels"], 3)},
    }
    parity_failed = False
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle (single-thread C++ restatement; the reference's mode NONE is single-threaded,
        # K/DeflateFilesContainer.java:22) on a bounded sample of the same workload.  Its output is also the parity
        # reference of this run: the GPU optimises the same sample stream and the two outputs are compared.
        import oracle_lib as O
        sample_n = min(16 << 20, n)
        sraw = raw[:sample_n]
        sstream = synth.deflate9(sraw)
        t0 = time.perf_counter()
        orc, oout, osaved, _, _ = O.optimise(sstream, bool(args.merge))
        cs = time.perf_counter() - t0
        gb = D.Batch([sstream]).run(bool(args.merge))
        gres = gb.result(0)
        gout = gb.output(0)
        gb.close()
        obits = O.size_bits(oout if orc == 0 else sstream)
        gbits = gres["size_bits_in"] - gres["saved_bits"]
        identical = gout == (oout if orc == 0 else sstream) and gres["status"] == orc and gres["saved_bits"] == (osaved if orc == 0 else 0)
        line["config"]["output_bit_delta_vs_reference"] = int(gbits - obits)
        line["config"]["output_bytes_identical_to_oracle"] = bool(identical)
        line["config"]["parity_sample"] = "first %d MiB of the timed input, re-deflated (zlib-9), merge=%s" % (sample_n >> 20, "on" if args.merge else "off")
        parity_failed = not identical
        line["cpu_baseline"] = {"value": round(sample_n / 1e6 / cs, 4), "unit": "MB/s", "cores": 1, "kind": "port",
                                "sample": "first %d MiB of the same generator (seed 0xD4F7), zlib-9 stream, %.1f s of CPU" % (sample_n >> 20, cs)}
        # all host cores, stream-parallel: one independent 2 MiB stream per core, the way the
        # reference would be run over many files; ctypes releases the GIL
        from concurrent.futures import ThreadPoolExecutor
        cores = min(os.cpu_count() or 1, 16)   # a one-GPU box's CPU share is 16 cores
        part = 2 << 20
        pieces = [synth.deflate9(raw[(k * part) % max(part, n - part):][:part]) for k in range(cores)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            list(ex.map(lambda a: O.optimise(a, bool(args.merge)), pieces))
        ca = time.perf_counter() - t0
        line["cpu_baseline"]["all_cores"] = {"value": round(cores * part / 1e6 / ca, 4), "unit": "MB/s", "cores": cores,
                                             "sample": "%d independent 2 MiB slices of the input, one oracle thread each, %.1f s" % (cores, ca)}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if parity_failed or not ok:
        sys.exit("bench.py: the GPU output differs from the oracle's (or does not round-trip)")


if __name__ == "__main__":
    main()
