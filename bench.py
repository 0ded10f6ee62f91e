#!/usr/bin/env python3
"""bench.py -- headline benchmark: haystack bytes scanned per second by the match-emitting hot path.

One "step" = one pass of runWithCase (am_run_batch: every match record, in the reference's fold
order, written to HBM) over one batch of synthetic UTF-8 haystacks that is already resident in HBM.
Default workload = BASELINE.json configs[2] (the configuration the metric is quoted on, it fits one
GPU): runLower / IgnoreCase, 100k lower-cased needles, 10240 x 1 MiB haystacks (10 GiB) per GPU.
Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL over xGMI); rank 0 flattens
the automaton and broadcasts the image, every rank scans its own shard of haystacks (weak scaling,
no collective on the data path), match counts are summed with an all-reduce at the end.

Prints ONE JSON line on rank 0 (see the contract in the task statement / DESIGN.md "Measurement").
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3_runLower_100k_10GiB")
    ap.add_argument("--hay-count", type=int, default=0, help="override the number of haystacks per GPU (smaller runs)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 general AC kernel, 2 suffix-filter kernel")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import alfred_margaret_amd as am
    from alfred_margaret_amd import dist as amdist
    from alfred_margaret_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libam has no CPU path")
    # AM_BENCH_SAME_DEVICE=1 + AM_BENCH_BACKEND=gloo: development aid to exercise the N-rank code path
    # (image broadcast, attach, sharding, all-reduce) on a box with a single GPU; never used for numbers
    same_device = os.environ.get("AM_BENCH_SAME_DEVICE") == "1"
    backend = os.environ.get("AM_BENCH_BACKEND", "nccl")
    dev_index = 0 if same_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"

    w = synth.WORKLOADS[args.workload]
    case = w["case"]
    n_hay = args.hay_count or w["n_hay"]
    hay_cells = w["hay_bytes"] // synth.CELL
    lib = am.api.libam()

    # ---- automaton: rank 0 builds + flattens, the image is broadcast over RCCL, others attach
    needles = synth.needles_for(args.workload)
    t0 = time.time()
    machine = None
    if rank == 0:
        machine = am.Automaton(needles)
        machine.set_kernel(args.kernel)
        handle = C.c_void_p(machine.device)
        nbytes = C.c_size_t(0)
        am.api.check(lib.am_automaton_image_size(handle, case, C.byref(nbytes)))
    build_s = time.time() - t0
    if world > 1:
        image = None
        if rank == 0:
            image = torch.empty(int(nbytes.value), dtype=torch.uint8, device=dev)
            am.api.check(lib.am_automaton_image_copy(handle, case, image.data_ptr(), image.numel()))
            torch.cuda.synchronize(dev)
        image = amdist.broadcast_image(image, dev, src=0)        # automaton over xGMI (RCCL broadcast)
        torch.cuda.synchronize(dev)
        if rank != 0:
            handle = C.c_void_p()
            am.api.check(lib.am_automaton_from_image(image.data_ptr(), image.numel(), C.byref(handle)))
            am.api.check(lib.am_automaton_set_kernel(handle, args.kernel))
    image_bytes = int(nbytes.value) if rank == 0 else 0

    # ---- this rank's shard of haystacks, generated in HBM (weak scaling: n_hay haystacks per GPU)
    n_cells = n_hay * hay_cells
    text, n_bytes = synth.haystacks_device(needles, w["mixed"], rank * n_cells, n_cells, dev)
    offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * w["hay_bytes"]
    batch = C.c_void_p()
    am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(batch)))

    def step():
        m = C.c_void_p()
        am.api.check(lib.am_run_batch(handle, case, batch, C.byref(m)))   # synchronises its stream before returning
        n = int(lib.am_matches_size(m))
        lib.am_matches_free(m)
        return n

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        n_records = step()
    am.api.check(lib.am_profile_reset())
    am.api.check(lib.am_profile_enable(1))
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_records = step()
    fence()
    elapsed = time.perf_counter() - t0
    am.api.check(lib.am_profile_enable(0))
    elapsed = amdist.allreduce_max(elapsed, dev)

    # values folded (= reference's countMatches) via the count-only entry point, summed over ranks
    total_values = C.c_uint64(0)
    t1 = time.perf_counter()
    am.api.check(lib.am_count_batch(handle, case, batch, None, C.byref(total_values)))
    count_only_s = time.perf_counter() - t1
    total_matches, total_records = amdist.allreduce_sum([int(total_values.value), n_records], dev)   # final gather of match counts

    kname = b"sf" if args.kernel != 1 else b"ac"
    ms, launches = C.c_double(0), C.c_uint64(0)
    am.api.check(lib.am_profile_read(kname, C.byref(ms), C.byref(launches)))
    if launches.value == 0:                            # automaton routed to the other kernel (e.g. empty needle)
        kname = b"ac" if kname == b"sf" else b"sf"
        am.api.check(lib.am_profile_read(kname, C.byref(ms), C.byref(launches)))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        gib = n_bytes * world / float(1 << 30)
        value = gib * args.steps / elapsed
        # dominant kernel: one launch per step scans the batch and writes every record.  Algorithmic
        # bytes per launch: 1 B per haystack byte + 16 B per record + 16 B per haystack.
        # (The general AC kernel runs a count launch and an emit launch per step: 8 B per record on average.)
        launches_n = max(int(launches.value), 1)
        avg_ms = ms.value / launches_n
        per_step = launches_n / float(args.steps)
        alg_bytes = n_bytes + (16.0 / per_step) * n_records + 16.0 * n_hay
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic per launch: not measurable from inside this process; taken from the committed PMC
        # profile of the same kernel + workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as
        # MI355X_MICROARCH.md prescribes), scaled to this launch's bytes.  null if there is no such profile.
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pt = json.load(f)
            if pt.get("workload") == args.workload and pt.get("kernel") == "k_" + kname.decode():
                traffic = int(pt["hbm_bytes_per_scanned_byte"] * n_bytes)
        except (OSError, ValueError, KeyError):
            traffic = None
        out = {
            "metric": "GiB/s haystack bytes scanned (match-emitting runLower, 100k-needle automaton)" if "cfg3" in args.workload
                      else "GiB/s haystack bytes scanned (match-emitting run)",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": args.workload, "n_needles": len(needles), "case": "IgnoreCase" if case else "CaseSensitive",
                       "haystacks_per_gpu": n_hay, "haystack_bytes": w["hay_bytes"], "bytes_per_gpu": n_bytes,
                       "parallelism": "haystack-sharded x%d" % world, "kernel": kname.decode(),
                       "automaton_image_bytes": image_bytes, "build_s": round(build_s, 2)},
            "matches_per_s": round(total_matches * args.steps / elapsed, 1),
            "matches_per_step": total_matches, "records_per_step": total_records,
            "count_only_gibps": round(n_bytes / float(1 << 30) / count_only_s, 3),
            "roofline": {"bound": "hbm", "kernel": "k_" + kname.decode(), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "avg_launch_ms": round(avg_ms, 4), "launches": int(launches.value), "alg_bytes_per_launch": int(alg_bytes)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, w, needles, case, hay_cells, handle, batch, lib)
        print(json.dumps(out), flush=True)

    lib.am_batch_destroy(batch)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, w, needles, case, hay_cells, handle, batch, lib):
    """The oracle (C restatement of the reference algorithm, 1 thread) on a bounded sample of the SAME
    workload: haystacks 0..k-1 of rank 0's shard, as many as fit the time budget.  Also a parity spot
    check: the GPU's per-haystack counts for the sample must equal the oracle's."""
    import numpy as np
    from alfred_margaret_amd import synth
    from oracle import oracle
    t0 = time.perf_counter()
    o = oracle.Machine(needles)
    build_s = time.perf_counter() - t0
    counts, scanned, spent, k = [], 0, 0.0, 0
    while spent < args.cpu_seconds and k < 64:
        hay = synth.haystacks_host(needles, w["mixed"], k * hay_cells, hay_cells)
        t1 = time.perf_counter()
        counts.append(o.count_matches(case, hay))
        spent += time.perf_counter() - t1
        scanned += hay.size
        k += 1
    n_hay = int(lib.am_batch_total_bytes(batch)) // w["hay_bytes"]
    gpu_counts = np.zeros(n_hay, np.uint64)
    import alfred_margaret_amd as am
    am.api.check(lib.am_count_batch(handle, case, batch, gpu_counts.ctypes.data, None))
    parity = [int(c) for c in gpu_counts[:k]] == counts
    if not parity:
        raise SystemExit("PARITY FAILURE: GPU counts differ from the oracle on the CPU-baseline sample")
    return {"value": round(scanned / float(1 << 30) / spent, 5), "unit": "GiB/s", "cores": 1, "kind": "port",
            "sample": "first %d haystacks (%d MiB) of the same workload, run only; oracle build %.2fs; GPU counts on the sample identical" % (k, scanned >> 20, build_s)}


if __name__ == "__main__":
    main()
