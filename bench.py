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
    ap.add_argument("--total-haystacks", type=int, default=0, help="strong-scaling workloads (cfg4): haystacks of the WHOLE job, block-sharded over the ranks (smaller runs)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 general AC kernel, 2 suffix-filter kernel")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample (rank 0, N=1)")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the all-host-cores leg of the CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive leg (am_count / am_run on pinned host slices)")
    ap.add_argument("--h2d-mib", type=int, default=2048, help="haystack bytes of the PCIe-inclusive leg")
    ap.add_argument("--no-parity", action="store_true", help="skip the fold-checksum parity gate (kernel vs kernel on every haystack, oracle on a sample)")
    ap.add_argument("--parity-oracle-mib", type=int, default=1024, help="haystack bytes of rank 0's shard the oracle re-scans for the parity gate")
    ap.add_argument("--plants", type=int, default=1, help="needles planted per 1-KiB cell of the synthetic haystacks (robustness sweep: 0, 1, 8, 64; BASELINE = 1)")
    ap.add_argument("--workloads", default=None, help="comma-separated BASELINE configurations measured AFTER the headline into the line's `workloads` object "
                    "(default: all of %s when the headline is the default cfg3 run on one GPU; 'none' switches them off)" % ", ".join(EXTRA_WORKLOADS))
    ap.add_argument("--workload-steps", type=int, default=5, help="timed steps of each entry of `workloads` (after 2 warm-up steps)")
    ap.add_argument("--workloads-oracle-mib", type=int, default=64, help="haystack bytes the oracle re-scans in each `workloads` entry's parity gate")
    ap.add_argument("--torch-collectives", action="store_true", help="N ranks: broadcast the automaton image and sum the counts with torch.distributed "
                    "instead of libam's own RCCL entry points (am_multi_*): cross-check of the product path")
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
    if "replacer" in args.workload:
        out = measure_replacer(args, args.workload, rank, world, dev)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if world == 1 and (args.gpus > 1 or os.environ.get("AM_BENCH_SINGLE_PROCESS") == "1"):     # the env var: the same path on a 1-GPU box (tests)
        return bench_single_process(args, w)          # one process drives all N GPUs through libam (am_multi_create)
    case = w["case"]
    # cfg4 (BASELINE configs[3]) is ONE batch of 1M haystacks block-sharded over the ranks (dist.shard_bounds): strong
    # scaling; every other workload gives each GPU its own n_hay haystacks: weak scaling
    strong = bool(w.get("sharded_total")) and not args.hay_count
    if strong:
        lo_hay, hi_hay = amdist.shard_bounds(args.total_haystacks or w["n_hay"], rank, world)
        n_hay, first_hay = hi_hay - lo_hay, lo_hay
    else:
        n_hay = args.hay_count or w["n_hay"]
        first_hay = rank * n_hay
    hay_cells = w["hay_bytes"] // synth.CELL
    lib = am.api.libam()

    # ---- automaton: rank 0 builds + flattens, the image is broadcast over RCCL, others attach
    needles = synth.needles_for(args.workload)
    t0 = time.time()
    machine = None
    build = None
    if rank == 0:
        machine, build = timed_build(needles, case)
        machine.set_kernel(args.kernel)
        handle = C.c_void_p(machine.device)
        nbytes = C.c_size_t(build["image_bytes"])
    build_s = time.time() - t0
    multi = None
    rccl_ms = {}
    if world > 1 and not args.torch_collectives:
        # the product's own multi-GPU entry points (include/am.h am_multi_*): RCCL communicator over the ranks, automaton image
        # broadcast over xGMI and the final all-reduce of counts inside libam; torch.distributed only carries the 128-byte id.
        # A failure here is an ERROR (round 2 fell back to torch.distributed silently: the driver could have timed the wrong path
        # without noticing); --torch-collectives selects the torch path explicitly, and the JSON line names what ran ("collectives").
        try:
            ident = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                buf = (C.c_uint8 * 128)()
                am.api.check(lib.am_multi_unique_id(buf))
                ident = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).clone()
            ident = ident.to(dev)
            dist.broadcast(ident, 0)
            ident_b = bytes(ident.cpu().numpy().tobytes())
            multi = C.c_void_p()
            with stdout_to_stderr():                       # RCCL prints a version banner on stdout
                am.api.check(lib.am_multi_create_rank(world, rank, (C.c_uint8 * 128).from_buffer_copy(ident_b), C.byref(multi)))
            autos = (C.c_void_p * 1)()
            if rank == 0:
                am.api.check(lib.am_automaton_image_size(handle, case, C.byref(nbytes)))      # flatten + upload outside the timed broadcast
            torch.cuda.synchronize(dev); dist.barrier()
            t_bc = time.perf_counter()
            am.api.check(lib.am_multi_broadcast_automaton(multi, handle if rank == 0 else None, case, 0, autos))
            rccl_ms["image_broadcast_ms"] = round((time.perf_counter() - t_bc) * 1e3, 3)      # size + flag round + blob over xGMI + attach + flag round
            handle = C.c_void_p(autos[0])
            am.api.check(lib.am_automaton_set_kernel(handle, args.kernel))
        except Exception as e:                              # noqa: BLE001
            raise SystemExit("rank %d: libam's multi-GPU path (am_multi_create_rank / am_multi_broadcast_automaton) failed: %s\n"
                             "No fallback is taken silently; re-run with --torch-collectives to time the torch.distributed path instead." % (rank, e))
    if world > 1 and multi is None:
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
    text, n_bytes = synth.haystacks_device(needles, w["mixed"], first_hay * hay_cells, n_cells, dev, plants=args.plants, natural=bool(w.get("natural")))
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
    if multi is not None:
        # the device-resident multi-GPU entry point: count on this rank's resident batch, ncclAllReduce of the totals inside libam
        autos1, batches1 = (C.c_void_p * 1)(handle), (C.c_void_p * 1)(batch)
        local_total, job_total = C.c_uint64(0), C.c_uint64(0)
        am.api.check(lib.am_multi_count_batch(multi, autos1, case, batches1, None, C.byref(local_total), C.byref(job_total)))
        count_only_s = time.perf_counter() - t1
        sums = np.array([n_records, n_bytes], dtype=np.uint64)
        t_ar = time.perf_counter()
        am.api.check(lib.am_multi_allreduce_sum(multi, sums.ctypes.data, 2))
        rccl_ms["allreduce_ms"] = round((time.perf_counter() - t_ar) * 1e3, 3)                # staging + ncclAllReduce of 3 words + read-back
        total_matches = int(job_total.value)
        total_records, amdist_total_bytes = (int(x) for x in sums)
    else:
        am.api.check(lib.am_count_batch(handle, case, batch, None, C.byref(total_values)))
        count_only_s = time.perf_counter() - t1
        total_matches, total_records, amdist_total_bytes = amdist.allreduce_sum([int(total_values.value), n_records, n_bytes], dev)

    parity = None
    if not args.no_parity:
        parity = parity_gate(args, w, needles, machine, handle, case, batch, text, n_hay, rank, world, dev, lib)

    kname = b"sf" if args.kernel != 1 else b"ac"
    ms, launches = C.c_double(0), C.c_uint64(0)
    am.api.check(lib.am_profile_read(kname, C.byref(ms), C.byref(launches)))
    if launches.value == 0:                            # automaton routed to the other kernel (e.g. empty needle)
        kname = b"ac" if kname == b"sf" else b"sf"
        am.api.check(lib.am_profile_read(kname, C.byref(ms), C.byref(launches)))
    dfa_ms, dfa_launches = C.c_double(0), C.c_uint64(0)
    am.api.check(lib.am_profile_read(b"dfa", C.byref(dfa_ms), C.byref(dfa_launches)))
    pair_ms = None
    if dfa_launches.value and args.kernel == 0:        # a dictionary on the table-walk route (csrc/am_dfa.hip): one k_dfa launch + one k_dfa_place launch per step -- the roofline is the pair's
        pl_ms, pl_n = C.c_double(0), C.c_uint64(0)
        am.api.check(lib.am_profile_read(b"dfa_place", C.byref(pl_ms), C.byref(pl_n)))
        pair_ms = {"k_dfa": round(dfa_ms.value / max(int(dfa_launches.value), 1), 4), "k_dfa_place": round(pl_ms.value / max(int(pl_n.value), 1), 4)}
        kname, ms, launches = b"dfa", C.c_double(dfa_ms.value + pl_ms.value), dfa_launches

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_bytes = amdist_total_bytes
        gib = total_bytes / float(1 << 30)
        value = gib * args.steps / elapsed
        # dominant kernel: one launch per step scans the batch and writes every record.  Algorithmic
        # bytes per launch: 1 B per haystack byte + 16 B per record + 16 B per haystack.
        # (The general AC kernel runs a count launch and an emit launch per step: 8 B per record on average.)
        launches_n = max(int(launches.value), 1)
        per_step = launches_n / float(args.steps)
        if kname in (b"sf", b"dfa"):
            # the suffix-filter route: one k_sf launch per step; algorithmic bytes per launch as above (the table-walk route: one k_dfa launch)
            avg_ms = ms.value / launches_n
            alg_bytes = n_bytes + 16.0 * n_records + 16.0 * n_hay
        else:
            avg_ms = ms.value / launches_n
            alg_bytes = n_bytes + (16.0 / per_step) * n_records + 16.0 * n_hay
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic per launch: not measurable from inside this process; taken from the committed PMC profile of the same kernel + workload
        traffic, traffic_source = pmc_traffic_entry(args.workload, "k_dfa + k_dfa_place" if pair_ms else "k_" + kname.decode(), n_bytes) if args.plants == 1 else (None, None)
        out = {
            "metric": "GiB/s haystack bytes scanned (match-emitting runLower, 100k-needle automaton)" if "cfg3" in args.workload
                      else "GiB/s haystack bytes scanned (match-emitting run)",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": args.workload + ("" if args.plants == 1 else " plants=%d" % args.plants), "n_needles": len(needles), "case": "IgnoreCase" if case else "CaseSensitive",
                       "haystacks_per_gpu": n_hay, "haystack_bytes": w["hay_bytes"], "bytes_per_gpu": n_bytes,
                       "parallelism": "haystack-sharded x%d%s" % (world, ", automaton broadcast + count all-reduce by libam (am_multi_*, RCCL)" if multi is not None else ""), "kernel": kname.decode(),
                       "automaton_image_bytes": image_bytes, "build_s": round(build_s, 2)},
            "build": build, "gpu_build_plus_run": build_plus_run(n_bytes, build, ms_per_step) if build else None,
            "matches_per_s": round(total_matches * args.steps / elapsed, 1),
            "matches_per_step": total_matches, "records_per_step": total_records,
            "count_only_gibps": round(n_bytes / float(1 << 30) / count_only_s, 3),
            "roofline": {"bound": "hbm", "kernel": "k_dfa + k_dfa_place" if pair_ms else "k_" + kname.decode(), **({"per_kernel_ms": pair_ms} if pair_ms else {}), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "avg_launch_ms": round(avg_ms, 4), "launches": int(launches.value), "alg_bytes_per_launch": int(alg_bytes)},
            "collectives": "none (1 GPU)" if world == 1 else ("libam-rccl" if multi is not None else "torch"),
        }
        if multi is not None:
            out["rccl"] = dict(rccl_ms, ranks=int(lib.am_multi_world_size(multi)), image_bytes=image_bytes,
                               data_path_collectives=0, what="am_multi_create_rank + am_multi_broadcast_automaton + am_multi_count_batch / am_multi_allreduce_sum (csrc/am_multi.cpp)")
        if rank == 0:
            # what the box gives: the kernels that count on two workgroups per CU (k_dfa, k_rp_lds, the small-filter k_sf) lose 10-45 % where a CU runs 16 wavefronts at a time
            waves, one_ms, two_ms = am.api.resident_waves()
            out["machine"] = {"resident_waves_per_cu": waves, "spin_16_per_cu_ms": one_ms, "spin_32_per_cu_ms": two_ms,
                              "what": "am_debug_resident_waves: a spinning kernel with 16 and with 32 wavefronts per CU; equal times = 32 resident (the architecture's number), twice = 16"}
        if parity is not None:
            out["parity"] = parity
        if world == 1 and not args.no_h2d:
            out["h2d_inclusive"] = h2d_inclusive(args, w, handle, case, text, n_hay, lib, settle=VRAM_WIPE_SETTLE_S if w.get("natural") else 0.0)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, w, needles, case, hay_cells, handle, batch, lib)
        if args.workloads is None:
            args.workloads = ",".join(EXTRA_WORKLOADS) if (world == 1 and args.workload == "cfg3_runLower_100k_10GiB" and not args.hay_count and args.plants == 1 and args.kernel == 0) else "none"
        if world == 1 and args.workloads != "none":
            lib.am_batch_destroy(batch); batch = None
            del text
            out["workloads"] = extra_workloads(args, dev, lib, machine, needles)
        print(json.dumps(out), flush=True)

    if batch is not None:
        lib.am_batch_destroy(batch)
    if multi is not None:
        lib.am_automaton_destroy(handle)
        lib.am_multi_destroy(multi)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


EXTRA_WORKLOADS = ("cfg2_runText_10k_1GiB", "cfg2_single_1GiB", "cfg4_100k_1M_haystacks", "natural_100k_10GiB", "cfg5_replacer_50k_1GiB")


def timed_build(needles, case):
    """`build` as the reference's benchmark times it -- together with the run (benchmark/haskell/app/Main.hs:62-64,73): (machine, split).  automaton_s = the host mirror of
    Automaton.build + am_automaton_create (which validates by flattening the CaseSensitive image; the IgnoreCase image is flattened on a thread of its own meanwhile);
    first_use_s = until the image of the workload's case mode lies in HBM (the rest of its flatten, the upload)."""
    import alfred_margaret_amd as am
    if not timed_build.warm:                      # the HIP runtime's start (context, first allocation, the library's streams: 0.1-0.15 s) is the process's, not the build's
        tiny = am.Automaton(["warm-up"]); n0 = C.c_size_t(0)
        am.api.check(am.api.libam().am_automaton_image_size(C.c_void_p(tiny.device), case, C.byref(n0)))
        del tiny
        timed_build.warm = True
    t0 = time.perf_counter()
    machine = am.Automaton(needles)
    t1 = time.perf_counter()
    nbytes = C.c_size_t(0)
    am.api.check(am.api.libam().am_automaton_image_size(C.c_void_p(machine.device), case, C.byref(nbytes)))
    t2 = time.perf_counter()
    return machine, {"automaton_s": round(t1 - t0, 3), "first_use_s": round(t2 - t1, 3), "total_s": round(t2 - t0, 3), "image_bytes": int(nbytes.value),
                     "what": "automaton_s: host build (mirror of Automaton.build) + am_automaton_create (validation = CaseSensitive flatten, IgnoreCase flatten on its own thread); "
                             "first_use_s: the case mode's image in HBM (rest of its flatten + upload); after a one-needle warm-up automaton (the HIP runtime's start is the process's)"}


timed_build.warm = False


def build_plus_run(n_bytes, build, ms_per_step):
    """One batch scanned by a freshly built automaton, every record written: GiB/s over build + one step (what cpu_baseline.value_build_plus_run is for the oracle)."""
    total = build["total_s"] + ms_per_step * 1e-3
    return {"value": round(n_bytes / float(1 << 30) / total, 2), "unit": "GiB/s", "seconds": round(total, 3), "what": "bytes of ONE batch / (build.total_s + ms_per_step)"}


def pmc_traffic_entry(workload, kernel, n_bytes):
    """HBM traffic of one launch from the committed PMC profile of the same kernel + workload (profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes), scaled to this launch's bytes; (None, None) without such a profile.
    Never a counter read in this run."""
    import alfred_margaret_amd as am
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f)
        e = pt.get("workloads", {}).get("cfg2_runText_10k_1GiB" if workload == "cfg2_single_1GiB" else workload)      # (one 1-GiB haystack: the same automaton over the same text cells)
        if e and e.get("kernel") == kernel and int(pt.get("image_version", -1)) == am.api.image_version():
            return int(e["hbm_bytes_per_scanned_byte"] * n_bytes), "profiles/pmc_traffic.json (%s): 2 x FETCH_SIZE + WRITE_SIZE of a %.1f-GiB launch of this workload, scaled; not read in this run" % (
                e.get("profile", "?"), e.get("launch_bytes", 0) / float(1 << 30))
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return None, None


def measure_scan_workload(args, name, dev, lib, machine=None, needles=None, n_hay=None, steps=5, warmup=2):
    """One BASELINE configuration other than the headline's, on this GPU, as an entry of the line's `workloads` object: the same step (am_run_batch:
    every record, sorted, in HBM, batch resident), the dominant kernel's average launch from HIP events, the roofline fraction by SURVEY 8d's
    algorithmic bytes, and its own parity block (parity_gate: k_sf == k_ac on every haystack, the oracle on a sample, full lists on 1 % of that)."""
    import copy
    import torch
    import alfred_margaret_amd as am
    from alfred_margaret_amd import synth
    w = synth.WORKLOADS[name]
    case = w["case"]
    t0 = time.time()
    if needles is None:
        needles = synth.needles_for(name)
    build = None
    if machine is None:
        machine, build = timed_build(needles, case)
    handle = C.c_void_p(machine.device)
    build_s = time.time() - t0
    n_hay = n_hay or w["n_hay"]
    hay_cells = w["hay_bytes"] // synth.CELL
    text, n_bytes = synth.haystacks_device(needles, w["mixed"], 0, n_hay * hay_cells, dev, natural=bool(w.get("natural")))
    offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * w["hay_bytes"]
    batch = C.c_void_p()
    am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(batch)))

    def step():
        m = C.c_void_p()
        am.api.check(lib.am_run_batch(handle, case, batch, C.byref(m)))
        n = int(lib.am_matches_size(m))
        lib.am_matches_free(m)
        return n

    try:
        for _ in range(warmup):
            n_records = step()
        am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            n_records = step()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        am.api.check(lib.am_profile_enable(0))
        ms, launches = C.c_double(0), C.c_uint64(0)
        am.api.check(lib.am_profile_read(b"dfa", C.byref(ms), C.byref(launches)))            # the table-walk route (dictionaries: natural text), else the suffix filter
        kernel = "k_dfa" if launches.value else "k_sf"
        pair_ms = None
        if launches.value:
            # the table-walk route writes its records with TWO kernels: the walk drops tokens, k_dfa_place turns them into the records.  The algorithmic bytes of a step
            # (text + records) are the pair's work, so the roofline is taken over the pair (over k_dfa alone the fraction flattered the route)
            ms2, l2 = C.c_double(0), C.c_uint64(0)
            am.api.check(lib.am_profile_read(b"dfa_place", C.byref(ms2), C.byref(l2)))
            pair_ms = {"k_dfa": round(ms.value / max(int(launches.value), 1), 4), "k_dfa_place": round(ms2.value / max(int(l2.value), 1), 4)}
            ms = C.c_double(ms.value + ms2.value)
            kernel = "k_dfa + k_dfa_place"
        if not launches.value:
            am.api.check(lib.am_profile_read(b"sf", C.byref(ms), C.byref(launches)))
        total = C.c_uint64(0)
        am.api.check(lib.am_count_batch(handle, case, batch, None, C.byref(total)))          # warm
        t1 = time.perf_counter()
        am.api.check(lib.am_count_batch(handle, case, batch, None, C.byref(total)))
        count_s = time.perf_counter() - t1
        other_route = None
        if kernel.startswith("k_dfa"):
            # the same step on the suffix-filter route (am_automaton_set_kernel(a, 2)), so that the line shows what the choice of route is worth
            am.api.check(lib.am_automaton_set_kernel(handle, 2))
            try:
                step()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(2):
                    n_sf = step()
                torch.cuda.synchronize(dev)
                sf_s = (time.perf_counter() - t1) / 2
                am.api.check(lib.am_count_batch(handle, case, batch, None, C.byref(total)))
                t1 = time.perf_counter()
                am.api.check(lib.am_count_batch(handle, case, batch, None, C.byref(total)))
                other_route = {"kernel": "k_sf", "value": round(n_bytes / float(1 << 30) / sf_s, 1), "count_only_gibps": round(n_bytes / float(1 << 30) / (time.perf_counter() - t1), 1),
                               "same_record_count": n_sf == n_records}
            finally:
                am.api.check(lib.am_automaton_set_kernel(handle, 0))
        gate_args = copy.copy(args)
        # (natural text runs the table walk, the route with the most intricate bookkeeping -- tokens, superblocks, k_dfa_place's sort --: its oracle sample is 512 spread haystacks)
        gate_args.kernel, gate_args.parity_oracle_mib = 0, (max(args.workloads_oracle_mib, 512) if w.get("natural") and args.workloads_oracle_mib else args.workloads_oracle_mib)
        parity = parity_gate(gate_args, w, needles, machine, handle, case, batch, text, n_hay, 0, 1, dev, lib) if not args.no_parity else {}
        contains_all = contains_all_row(args, w, needles, machine, handle, case, batch, text, n_hay, n_bytes, lib) if name == "cfg2_runText_10k_1GiB" and not args.no_parity else None
        # natural text from HOST slices: the result is 2.5 x its text, so the call is bound by the records' way back -- am_run goes up in segments and brings a segment's
        # records down while the next one is uploaded and scanned (csrc/am_abi.cpp run_segmented)
        host_slices = None
        if w.get("natural") and not args.no_h2d:
            host_slices = h2d_inclusive(args, w, handle, case, text, n_hay, lib, settle=VRAM_WIPE_SETTLE_S)
    finally:
        lib.am_batch_destroy(batch)
    avg_ms = ms.value / max(int(launches.value), 1)
    alg_bytes = n_bytes + 16.0 * n_records + 16.0 * n_hay                                       # 1 B per haystack byte + 16 B per record + 16 B per haystack
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, _src = pmc_traffic_entry(name, kernel, n_bytes)
    gib = n_bytes / float(1 << 30)
    return {"value": round(gib * steps / elapsed, 1), "unit": "GiB/s", "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "count_only_gibps": round(gib / count_s, 1),
            "config": {"n_needles": len(needles), "case": "IgnoreCase" if case else "CaseSensitive", "haystacks": n_hay, "haystack_bytes": w["hay_bytes"], "bytes": n_bytes},
            "records_per_step": n_records, "values_per_step": int(total.value),
            "roofline": {"kernel": kernel, "avg_launch_ms": round(avg_ms, 4), **({"per_kernel_ms": pair_ms} if pair_ms else {}), "launches": int(launches.value), "alg_bytes_per_launch": int(alg_bytes),
                         "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic},
            **({"other_route": other_route} if other_route else {}),
            "parity": {k: parity.get(k) for k in ("hashed", "kernels_agree", "oracle_checked", "oracle_bytes", "oracle_max_byte_offset", "oracle_what", "full_lists_checked", "matches_in_checked")},
            "build_s": round(build_s, 2), **({"build": build, "gpu_build_plus_run": build_plus_run(n_bytes, build, elapsed / steps * 1e3)} if build else {}),
            **({"contains_all": contains_all} if contains_all else {}), **({"h2d_inclusive": host_slices} if host_slices else {})}


def contains_all_row(args, w, needles, machine, handle, case, batch, text, n_hay, n_bytes, lib):
    """Searcher.containsAll (Searcher.hs:173-187) over the same resident batch, for the `Searcher Int` of buildNeedleIdSearcher: the direct route (k_sf's
    ids mode: the scan sets the needle-id bits itself, no record is written, a complete haystack is left) against the record route (full scan + k_idset),
    flags equal on every haystack and equal to the oracle's on the first ones."""
    import numpy as np
    import alfred_margaret_amd as am
    voff = np.ascontiguousarray(machine.values_off(), dtype=np.uint64)
    vals = np.ascontiguousarray(machine.values(), dtype=np.uint32)
    ids = C.c_void_p()
    am.api.check(lib.am_needle_ids_create(handle, voff.ctypes.data, vals.ctypes.data, len(needles), C.byref(ids)))
    try:
        def run():
            f = np.zeros(n_hay, np.uint8)
            t0 = time.perf_counter()
            am.api.check(lib.am_contains_all_batch(ids, case, batch, f.ctypes.data))
            return f, time.perf_counter() - t0
        run()
        direct, t_direct = min((run() for _ in range(3)), key=lambda x: x[1])
        am.debug_set("AM_NO_IDS_SCAN", 1)
        try:
            run()
            folded, t_fold = min((run() for _ in range(2)), key=lambda x: x[1])
        finally:
            am.debug_set("AM_NO_IDS_SCAN", -1)
    finally:
        lib.am_needle_ids_destroy(ids)
    if not np.array_equal(direct, folded):
        raise SystemExit("PARITY FAILURE: containsAll's direct route and its record route disagree")
    o, _ = get_oracle(needles)
    hb = w["hay_bytes"]
    k = min(n_hay, 64)
    host = text[:k * hb].cpu().numpy()
    exp = [o.contains_all(case, host[i * hb:(i + 1) * hb]) for i in range(k)]
    if [bool(x) for x in direct[:k]] != exp:
        raise SystemExit("PARITY FAILURE: containsAll differs from the oracle")
    gib = n_bytes / float(1 << 30)
    return {"gibps": round(gib / t_direct, 1), "ms": round(t_direct * 1e3, 3), "record_route_ms": round(t_fold * 1e3, 3), "routes_agree": n_hay, "oracle_checked": k,
            "true_flags": int(direct.sum())}


def extra_workloads(args, dev, lib, cfg3_machine, cfg3_needles):
    """Default run, one GPU: every other BASELINE configuration next to the headline, so that the driver's record of this run witnesses all of them
    (VERDICT r4 item 1).  cfg4 at ONE RANK'S SHARE of the 8-GPU job (131 072 of the 1 048 576 haystacks = 12.5 GiB; the whole 100 GiB on one GPU is
    --workload cfg4_100k_1M_haystacks), with the headline's automaton (the same needles)."""
    from alfred_margaret_amd import synth
    import torch
    out = {}
    t_all = time.time()
    for name in [n.strip() for n in args.workloads.split(",") if n.strip()]:
        t0 = time.time()
        if name not in EXTRA_WORKLOADS:
            raise SystemExit("--workloads: unknown workload %r (known: %s)" % (name, ", ".join(EXTRA_WORKLOADS)))
        if "replacer" in name:
            r = measure_replacer(args, name, 0, 1, dev, steps=args.workload_steps, warmup=2, extra=True)
            e = {"value": round(r["value"], 1), "unit": "GiB/s", "ms_per_step": r["ms_per_step"], "steps": r["steps"], "results": "device-resident",
                 "host_results": {k: r["host_results"][k] for k in ("value", "ms_per_step", "d2h_wire_gibps")},
                 "config": {k: r["config"][k] for k in ("n_pairs", "case", "haystacks_per_gpu", "haystack_bytes", "bytes_per_gpu")}, "passes": r["passes"],
                 "kernel_ms_per_step": {k: v for k, v in r["kernel_ms_per_step"].items() if v >= 0.05},
                 "roofline": {k: r["roofline"][k] for k in ("kernel", "avg_launch_ms", "launches", "alg_bytes_per_launch", "achieved", "frac", "traffic")},
                 "parity": {k: r["parity"][k] for k in ("loops_agree", "oracle_checked")} if "parity" in r else None, "build_s": r["config"]["build_s"]}
        elif name == "cfg4_100k_1M_haystacks":
            e = measure_scan_workload(args, name, dev, lib, machine=cfg3_machine, needles=cfg3_needles, n_hay=synth.WORKLOADS[name]["n_hay"] // 8, steps=args.workload_steps)
            e["config"]["share"] = "one rank's block of 8 (dist.shard_bounds): 131072 of 1048576 haystacks"
        else:
            e = measure_scan_workload(args, name, dev, lib, steps=args.workload_steps)
        e["wall_s"] = round(time.time() - t0, 1)
        out[name] = e
        torch.cuda.empty_cache()
        print("[bench] workload %s: %s" % (name, json.dumps(e)), file=sys.stderr, flush=True)
    out["wall_s"] = round(time.time() - t_all, 1)
    return out


class stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block (RCCL prints a version banner on stdout when a communicator is
    made; rank 0's stdout must carry the one JSON line only)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        C.CDLL(None).fflush(None)                     # the banner sits in C stdio's buffer: flush it while fd 1 still points at stderr
        os.dup2(self.saved, 1)
        os.close(self.saved)


def bench_single_process(args, w):
    """python bench.py --gpus N without torch.distributed.run: ONE process drives the N GPUs through the product's C ABI
    (ncclCommInitAll inside am_multi_create, ncclBroadcast of the automaton image, a host thread per device
    launching its block's scans concurrently, ncclAllReduce of the counts).  Same workload per GPU, same timing rule
    (K steps, all devices inside each step) and the same JSON line as the one-process-per-GPU path."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor

    import alfred_margaret_amd as am
    from alfred_margaret_amd import dist as amdist
    from alfred_margaret_amd import synth
    lib = am.api.libam()
    N = args.gpus
    if torch.cuda.device_count() < N:
        raise SystemExit("bench.py --gpus %d: only %d devices visible" % (N, torch.cuda.device_count()))
    case = w["case"]
    strong = bool(w.get("sharded_total")) and not args.hay_count
    hay_cells = w["hay_bytes"] // synth.CELL
    needles = synth.needles_for(args.workload)
    torch.cuda.set_device(0)
    t0 = time.time()
    machine = am.Automaton(needles)
    build_s = time.time() - t0
    multi = C.c_void_p()
    with stdout_to_stderr():
        am.api.check(lib.am_multi_create(N, C.byref(multi)))
    autos = (C.c_void_p * N)()
    am.api.check(lib.am_multi_broadcast_automaton(multi, machine.device, case, 0, autos))
    nbytes = C.c_size_t(0)
    am.api.check(lib.am_automaton_image_size(machine.device, case, C.byref(nbytes)))
    texts, batches, sizes, hays = [], [], [], []
    for i in range(N):
        if strong:
            lo, hi = amdist.shard_bounds(args.total_haystacks or w["n_hay"], i, N)
        else:
            lo, hi = i * (args.hay_count or w["n_hay"]), (i + 1) * (args.hay_count or w["n_hay"])
        dev = torch.device("cuda", i)
        torch.cuda.set_device(i)
        text, n_bytes = synth.haystacks_device(needles, w["mixed"], lo * hay_cells, (hi - lo) * hay_cells, dev, plants=args.plants, natural=bool(w.get("natural")))
        offs = torch.arange(hi - lo + 1, dtype=torch.int64, device=dev) * w["hay_bytes"]
        b = C.c_void_p()
        am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), hi - lo, n_bytes, C.byref(b)))   # the batch lives where its memory does
        am.api.check(lib.am_automaton_set_kernel(autos[i], args.kernel))
        texts.append((text, offs)); batches.append(b); sizes.append(n_bytes); hays.append(hi - lo)
    torch.cuda.set_device(0)
    # The step is ONE call of the product's device-resident multi-GPU entry point: am_multi_run_batch scans batches[i] on device i
    # (a host thread and a stream per device inside libam), leaves the records in each device's HBM and all-reduces their number.
    autos_arr = (C.c_void_p * N)(*[autos[i] for i in range(N)])
    batches_arr = (C.c_void_p * N)(*[b.value for b in batches])

    def step():
        res = (C.c_void_p * N)()
        total = C.c_uint64(0)
        am.api.check(lib.am_multi_run_batch(multi, autos_arr, case, batches_arr, res, C.byref(total)))
        recs = [int(lib.am_matches_size(res[i])) if res[i] else 0 for i in range(N)]
        for i in range(N):
            lib.am_matches_free(res[i])
        assert sum(recs) == int(total.value)
        return recs

    for _ in range(args.warmup):
        recs = step()
    am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
    for i in range(N):
        torch.cuda.synchronize(i)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        recs = step()
    for i in range(N):
        torch.cuda.synchronize(i)
    elapsed = time.perf_counter() - t0
    am.api.check(lib.am_profile_enable(0))

    local_totals = (C.c_uint64 * N)()
    job_total = C.c_uint64(0)
    am.api.check(lib.am_multi_count_batch(multi, autos_arr, case, batches_arr, None, local_totals, C.byref(job_total)))   # ncclAllReduce over the N devices
    total_matches, total_records, total_bytes = int(job_total.value), sum(recs), sum(sizes)
    assert total_matches == sum(int(x) for x in local_totals)
    kname = b"sf" if args.kernel != 1 else b"ac"
    ms, launches = C.c_double(0), C.c_uint64(0)
    am.api.check(lib.am_profile_read(kname, C.byref(ms), C.byref(launches)))
    launches_n = max(int(launches.value), 1)
    avg_ms = ms.value / launches_n                                                # average over every device's launches
    per_dev_bytes = total_bytes / N
    alg_bytes = per_dev_bytes + 16.0 * (total_records / N) + 16.0 * (sum(hays) / N)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    out = {
        "metric": "GiB/s haystack bytes scanned (match-emitting runLower, 100k-needle automaton)" if "cfg3" in args.workload else "GiB/s haystack bytes scanned (match-emitting run)",
        "value": round(total_bytes / float(1 << 30) * args.steps / elapsed, 3), "unit": "GiB/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": args.workload, "n_needles": len(needles), "case": "IgnoreCase" if case else "CaseSensitive", "haystacks_per_gpu": hays[0],
                   "haystack_bytes": w["hay_bytes"], "bytes_per_gpu": sizes[0], "parallelism": "haystack-sharded x%d, one process (am_multi_create: ncclCommInitAll, image broadcast; am_multi_run_batch / am_multi_count_batch on device-resident batches)" % N,
                   "kernel": kname.decode(), "automaton_image_bytes": int(nbytes.value), "build_s": round(build_s, 2)},
        "matches_per_s": round(total_matches * args.steps / elapsed, 1), "matches_per_step": total_matches, "records_per_step": total_records,
        "roofline": {"bound": "hbm", "kernel": "k_" + kname.decode(), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None, "avg_launch_ms": round(avg_ms, 4), "launches": int(launches.value),
                     "alg_bytes_per_launch": int(alg_bytes)},
        "collectives": "libam-rccl",
        "rccl": {"ranks": N, "self_check": "am_multi_create: one-word all-reduce of (rank + 1) verified on every device", "data_path_collectives": 0},
    }
    if not args.no_parity:
        # the same gate as the one-process-per-GPU path, device by device: on every device the route's kernel == k_ac on every haystack of its batch; the oracle's spread
        # sample in full on device 0, a few haystacks of each of the others
        import copy
        blocks = []
        for i in range(N):
            torch.cuda.set_device(i)
            a_i = copy.copy(args)
            if i:
                a_i.parity_oracle_mib = max(4, (4 * w["hay_bytes"]) >> 20)
            blocks.append(parity_gate(a_i, w, needles, machine, C.c_void_p(autos[i]), case, batches[i], texts[i][0], hays[i], 0, 1, torch.device("cuda", i), lib))
        torch.cuda.set_device(0)
        out["parity"] = {"per_device": [{k: b.get(k) for k in ("hashed", "kernels_agree", "oracle_checked", "oracle_bytes", "oracle_max_byte_offset", "full_lists_checked")} for b in blocks],
                         "hashed": sum(b["hashed"] for b in blocks), "kernels_agree": all(b["kernels_agree"] is not False for b in blocks), "oracle_what": blocks[0].get("oracle_what")}
    print(json.dumps(out), flush=True)
    for i in range(N):
        lib.am_batch_destroy(batches[i])
        lib.am_automaton_destroy(autos[i])
    lib.am_multi_destroy(multi)


VRAM_WIPE_SETTLE_S = 3.0


def measure_replacer(args, workload, rank, world, dev, steps=None, warmup=None, extra=False):
    """--workload cfg5_replacer_50k_1GiB (BASELINE.json configs[4]): one step = Replacer.run over the whole batch,
    every pass on the device.  Like the scan metric (records stay in HBM), `value` is measured with the rewritten texts
    left in device memory (am_replacer_run_batch_device); `host_results` in the same JSON line is the rate of
    am_replacer_run_batch, which also moves every result over PCIe into pinned host memory.  Every rank builds the same
    Replacer (0.4 s) and rewrites its own shard of haystacks; no collective on the data path."""
    import numpy as np
    import torch
    import torch.distributed as dist

    import alfred_margaret_amd as am
    from alfred_margaret_amd import dist as amdist
    from alfred_margaret_amd import synth
    lib = am.api.libam()
    w = synth.WORKLOADS[workload]
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    case = w["case"]
    n_hay = (0 if extra else args.hay_count) or w["n_hay"]
    hay_cells = w["hay_bytes"] // synth.CELL
    pairs = synth.replacer_pairs(workload)
    t0 = time.time()
    rep = am.Replacer(case, pairs)
    rdev = C.c_void_p(rep.device)
    build_s = time.time() - t0
    n_cells = n_hay * hay_cells
    needles = [p[0] for p in pairs]
    text, n_bytes = synth.haystacks_device(needles, w["mixed"], rank * n_cells, n_cells, dev)
    offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * w["hay_bytes"]
    batch = C.c_void_p()
    am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(batch)))
    last = {}

    def step(on_device=True):
        if "res" in last:                                  # hand the previous result's slabs back first, as a caller would
            lib.am_replaced_free(last.pop("res"))
        res = C.c_void_p()
        run = lib.am_replacer_run_batch_device if on_device else lib.am_replacer_run_batch
        am.api.check(run(rdev, batch, C.c_uint64(2**64 - 1), C.byref(res)))
        last["res"] = res
        return int(lib.am_replaced_passes(res)), int(lib.am_replaced_scanned_bytes(res))

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        passes, scanned = step()
    fence()
    elapsed = amdist.allreduce_max(time.perf_counter() - t0, dev)
    # the same steps with the results brought to the host (PCIe-inclusive; reported next to `value`, never as `value`).
    # First let the driver finish wiping VRAM that was freed a moment ago (the workload before this one, the generator's temporaries): amdgpu clears freed
    # VRAM in the background THROUGH THE SDMA ENGINES, and for about a second after a large hipFree every device->host copy runs at 27 instead of 53 GiB/s
    # (measured in round 5, tools/experiments/host_results/: 39 ms per step in that second, 22.4 ms before and after; LABNOTES R5.6)
    fence()
    time.sleep(VRAM_WIPE_SETTLE_S)
    step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(False)
    fence()
    elapsed_host = amdist.allreduce_max(time.perf_counter() - t0, dev)
    # the per-kernel breakdown comes from ONE extra step with the HIP-event brackets on: a pass is a few dozen launches of
    # microseconds each, and two event records per launch would be part of what is measured
    am.api.check(lib.am_profile_reset()); am.api.check(lib.am_profile_enable(1))
    step()
    fence()
    am.api.check(lib.am_profile_enable(0))
    prof_steps = 1
    prof = {}
    for k in (b"sf", b"ac", b"rp_lds", b"rp_loop", b"rp_pass", b"rp_splice", b"rp_scans", b"rp_windows", b"rp_merge", b"permute", b"rp_ranges", b"rp_route", b"pt_build", b"pt_materialise", b"scan", b"hidx"):
        ms, n = C.c_double(0), C.c_uint64(0)
        am.api.check(lib.am_profile_read(k, C.byref(ms), C.byref(n)))
        prof[k.decode()] = (ms.value, int(n.value))
    total_scanned = amdist.allreduce_sum([scanned], dev)[0]
    spliced = int(lib.am_replaced_spliced_bytes(last["res"]))
    if rank == 0:
        # dominant kernel: since the scans after the first pass only cover windows around the replacements, it is
        # k_rp_splice (replace, Replacer.hs:163-180).  Algorithmic bytes per launch: every byte of the rewritten text is
        # read once and written once.
        # ... or, when all passes of a haystack run inside one kernel, k_rp_loop: the algorithmic bytes of Replacer.run are the text read once and
        # the rewritten text written once; record lists, piece lists and re-scanned windows are bookkeeping.  It is a latency-bound kernel,
        # and the fraction says so
        kname = max(("rp_lds", "rp_loop", "rp_splice", "sf", "ac"), key=lambda k: prof[k][0])
        avg_ms = prof[kname][0] / max(prof[kname][1], 1)
        alg_bytes = {"rp_splice": 2.0 * spliced, "rp_loop": float(n_bytes + spliced), "rp_lds": float(n_bytes + spliced)}.get(kname, float(scanned)) * prof_steps / max(prof[kname][1], 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # ... and over the STEP: the text read once + the rewritten text written once are moved by the first scan and k_pt_materialise as much as by the loop
        # kernel, so the fraction that means something is (input + result bytes) / ms_per_step; the loop kernel's own figure stays beside it
        step_ms = elapsed / steps * 1e3
        step_achieved = float(n_bytes + spliced) / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0
        rp_traffic, rp_traffic_source = pmc_traffic_entry(workload, "k_" + kname, n_bytes)
        out = {
            "metric": "GiB/s of input rewritten by Replacer.run (50k pairs, all passes)", "value": round(n_bytes * world / float(1 << 30) * steps / elapsed, 3),
            "unit": "GiB/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "n_pairs": len(pairs), "case": "IgnoreCase" if case else "CaseSensitive", "haystacks_per_gpu": n_hay,
                       "haystack_bytes": w["hay_bytes"], "bytes_per_gpu": n_bytes, "parallelism": "haystack-sharded x%d" % world, "build_s": round(build_s, 2)},
            "results": "device-resident (am_replacer_run_batch_device)",
            "host_results": {"value": round(n_bytes * world / float(1 << 30) * steps / elapsed_host, 3), "unit": "GiB/s", "ms_per_step": round(elapsed_host / steps * 1e3, 3),
                             "what": "am_replacer_run_batch: the same passes plus every rewritten text copied over PCIe into pinned host memory",
                             "settled_s": VRAM_WIPE_SETTLE_S},
            "passes": passes, "scanned_gib_per_step": round(total_scanned / float(1 << 30), 2), "spliced_gib_per_step": round(spliced / float(1 << 30), 2),
            "kernel_ms_per_step": {k: round(v[0] / prof_steps, 3) for k, v in prof.items() if v[1]},
            "roofline": {"bound": "hbm", "kernel": "the step: first scan + k_%s + k_pt_materialise" % kname if kname in ("rp_loop", "rp_lds") else "k_" + kname,
                         "achieved": round(step_achieved if kname in ("rp_loop", "rp_lds") else achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round((step_achieved if kname in ("rp_loop", "rp_lds") else achieved) / HBM_PEAK_GBPS, 5),
                         "loop_kernel": {"kernel": "k_" + kname, "avg_launch_ms": round(avg_ms, 4), "achieved_over_its_launch": round(achieved, 2), "frac_over_its_launch": round(achieved / HBM_PEAK_GBPS, 5)},
                         "traffic": rp_traffic, "traffic_source": rp_traffic_source, "avg_launch_ms": round(step_ms if kname in ("rp_loop", "rp_lds") else avg_ms, 4), "launches": prof[kname][1],
                         "alg_bytes_per_launch": int(alg_bytes),
                         "note": "k_%s runs every pass of every haystack (one wavefront per haystack): bound by dependent-load latency, not by HBM" % kname if kname in ("rp_loop", "rp_lds") else None},
        }
        if world == 1 and not args.no_parity:
            out["parity"] = replacer_parity(args, w, pairs, case, rdev, batch, text, n_hay, n_bytes, last["res"], lib, dev, cpu_seconds=2.0 if extra else args.cpu_seconds)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = out["parity"].pop("cpu_baseline")
            else:
                out["parity"].pop("cpu_baseline")
        if world == 1:
            out["host_results"]["d2h_wire_gibps"] = d2h_wire_rate(dev)
    lib.am_replaced_free(last["res"])
    lib.am_batch_destroy(batch)
    return out if rank == 0 else None


def d2h_wire_rate(dev, mib=1024):
    """What the box's PCIe link gives a plain device -> pinned-host copy of 1 GiB (the better of two): the ceiling of every host-result number."""
    import torch
    src = torch.empty(mib << 20, dtype=torch.uint8, device=dev)
    dst = torch.empty(mib << 20, dtype=torch.uint8).pin_memory()
    best = float("inf")
    for _ in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(dev)
        best = min(best, time.perf_counter() - t0)
    return round(mib / 1024.0 / best, 2)


def replacer_parity(args, w, pairs, case, rdev, batch, text, n_hay, n_bytes, res_dev, lib, dev, cpu_seconds):
    """Parity of the Replacer step that was just timed (BASELINE configs[4]; Replacer.hs:203-274):
      1. EVERY haystack: the texts of the default route (k_rp_loop: all passes of a haystack in one kernel) == the texts of the pass-by-pass loops
         (AM_RP_LOOP=0; csrc/am_replace.hip), byte for byte -- two independent implementations of the pass loop;
      2. the first haystacks == the CPU oracle's Replacer.run, as many as fit cpu_seconds on the host's cores (at least 16); their single-core rate is the CPU baseline."""
    import numpy as np
    import alfred_margaret_amd as am
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle

    def host_texts(switch):
        am.debug_set("AM_RP_LOOP", switch)
        try:
            res = C.c_void_p()
            am.api.check(lib.am_replacer_run_batch(rdev, batch, C.c_uint64(2**64 - 1), C.byref(res)))
        finally:
            am.debug_set("AM_RP_LOOP", -1)
        return res

    def view(res, i):
        ptr, ln = C.c_void_p(), C.c_size_t(0)
        just = lib.am_replaced_get(res, i, C.byref(ptr), C.byref(ln))
        if just != 1:
            return None
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(ln.value,)) if ln.value else np.zeros(0, np.uint8)

    t0 = time.perf_counter()
    a, b = host_texts(-1), host_texts(0)
    try:
        for i in range(n_hay):
            x, y = view(a, i), view(b, i)
            if (x is None) != (y is None) or (x is not None and not np.array_equal(x, y)):
                raise SystemExit("PARITY FAILURE: the one-kernel Replacer loop and the pass-by-pass loop rewrite haystack %d differently" % i)
        loops_s = time.perf_counter() - t0
        hb = w["hay_bytes"]
        orc = oracle.Replacer(case, pairs)
        host = text[:min(n_bytes, 256 * hb)].cpu().numpy()
        threads = min(host_cores(), 64)
        t1 = time.perf_counter(); exp0 = orc.run(bytes(host[:hb])); per_hay = time.perf_counter() - t1
        k = int(max(16, min(n_hay, 256, threads * cpu_seconds / max(per_hay, 1e-6))))
        k = min(k, n_hay, host.size // hb)
        t1 = time.perf_counter()
        with ThreadPoolExecutor(threads) as pool:
            exp = [exp0] + list(pool.map(lambda i: orc.run(bytes(host[i * hb:(i + 1) * hb])), range(1, k)))
        oracle_s = time.perf_counter() - t1
        for i in range(k):
            x = view(a, i)
            if x is None or x.tobytes() != exp[i]:
                raise SystemExit("PARITY FAILURE: device Replacer output differs from the oracle on haystack %d" % i)
            buf, n = C.create_string_buffer(len(exp[i]) + 16), C.c_size_t(0)                  # the device-resident result of the timed steps, too
            if lib.am_replaced_read(res_dev, i, buf, len(buf), C.byref(n)) != 1 or buf.raw[:n.value] != exp[i]:
                raise SystemExit("PARITY FAILURE: device-resident Replacer output differs from the oracle on haystack %d" % i)
    finally:
        lib.am_replaced_free(a); lib.am_replaced_free(b)
    return {"loops_agree": n_hay, "loops": "k_rp_loop == pass-by-pass (AM_RP_LOOP=0), every text byte for byte", "loops_s": round(loops_s, 2),
            "oracle_checked": k, "oracle_cores": threads, "oracle_s": round(oracle_s, 2),
            "cpu_baseline": {"value": round(hb / float(1 << 30) / per_hay, 6), "unit": "GiB/s", "cores": 1, "kind": "port",
                             "sample": "the first haystack (%d KiB) through the oracle's Replacer.run on one core; %d haystacks on %d cores identical to the device's" % (hb >> 10, k, threads)}}


def h2d_inclusive(args, w, handle, case, text, n_hay, lib, settle=0.0):
    """What a Haskell `runText` caller gets through the shim of INTEGRATION.md (SURVEY 7.4 H4 / 8d "reported separately"): the one-shot
    entry points am_count / am_run on HOST slices -- gather into pinned staging, DMA over PCIe, scan, results back -- timed on the first
    --h2d-mib MiB of the benchmark batch, copied to pinned host memory first.  Never `value`."""
    import numpy as np
    import torch
    import alfred_margaret_amd as am
    hb = w["hay_bytes"]
    k = max(1, min(n_hay, (args.h2d_mib << 20) // hb))
    host = torch.empty(k * hb, dtype=torch.uint8).pin_memory()
    host.copy_(text[:k * hb])
    torch.cuda.synchronize()
    slices = (am.api.Slice * k)()
    base = host.data_ptr()
    for i in range(k):
        slices[i].ptr, slices[i].off, slices[i].len = base, i * hb, hb
    counts = np.zeros(k, dtype=np.uint64)
    out = {"sample": "first %d haystacks (%d MiB) of the batch as pinned host slices; the better of two calls each, after one warm-up call" % (k, k * hb >> 20)}
    am.api.check(lib.am_count(handle, case, slices, k, counts.ctypes.data))          # warm-up: staging buffers, workspaces
    t_count = float("inf")
    for _ in range(2):                                                                  # the better of two: the gather threads share the host with whatever else runs there
        t0 = time.perf_counter()
        am.api.check(lib.am_count(handle, case, slices, k, counts.ctypes.data))
        t_count = min(t_count, time.perf_counter() - t0)
    m = C.c_void_p()
    am.api.check(lib.am_run(handle, case, slices, k, C.byref(m)))
    lib.am_matches_data(m); lib.am_matches_free(m)
    if settle:
        # the warm-up call's first freed record array pushes the previous steps' whole-batch array (tens of GB) out of the library's cache of freed arrays: a hipFree, and
        # amdgpu wipes the VRAM through the SDMA engines -- D2H copies run at 27 instead of 42-50 GB/s until it is done (LABNOTES R5.6, R6.9)
        time.sleep(settle)
        out["settled_s"] = settle
    t_run = float("inf")
    for _ in range(2):
        t0 = time.perf_counter()
        am.api.check(lib.am_run(handle, case, slices, k, C.byref(m)))
        n_rec = int(lib.am_matches_size(m))
        lib.am_matches_data(m)                                                          # the records on the host: part of what the caller waits for
        t_run = min(t_run, time.perf_counter() - t0)
        lib.am_matches_free(m)
    gib = k * hb / float(1 << 30)
    out.update({"count_gibps": round(gib / t_count, 2), "run_gibps": round(gib / t_run, 2), "count_ms": round(t_count * 1e3, 2), "run_ms": round(t_run * 1e3, 2),
                "records": n_rec, "values": int(counts.sum())})
    return out


def host_cores():
    """Cores this process may actually use: the affinity mask, cut down to the cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, -(-int(parts[0]) // int(parts[1]))))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read().split()[0])
                if quota > 0:
                    n = min(n, max(1, -(-quota // period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


_ORACLE = {}


def get_oracle(needles):
    """The CPU oracle's machine for these needles (built once per process): (machine, build seconds)."""
    from oracle import oracle
    key = id(needles)
    if key not in _ORACLE:
        t0 = time.perf_counter()
        m = oracle.Machine(needles)
        _ORACLE[key] = (m, time.perf_counter() - t0)
    return _ORACLE[key]


def parity_gate(args, w, needles, machine, handle, case, batch, text, n_hay, rank, world, dev, lib):
    """SURVEY 8d "parity check at scale" (the reference's harness asserts result identity on every run,
    benchmark/benchmark.py:65-69).  Three layers, all on the batch that was just timed:
      1. every haystack of every rank: the order-sensitive 64-bit checksum of the fold sequence (am_matches_fold_hash,
         computed in HBM) of the suffix-filter kernel's result == that of the general AC-walk kernel's result -- two
         independent algorithms (k_ac is test infrastructure: libam_check.so, loaded here, after the timed region);
      2. rank 0, the first --parity-oracle-mib MiB of its shard: == the CPU oracle's runWithCase folded with the same
         hash function, on every host core (a haystack larger than that budget -- cfg2_single's one 1-GiB document -- is checked on
         a prefix of it scanned as a haystack of its own: one oracle thread);
      3. rank 0, 1 % of those haystacks: the expanded (matchPos, value) lists record by record.
    Any difference aborts the benchmark."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import alfred_margaret_amd as am
    from alfred_margaret_amd import synth
    am.api.load_check()            # k_ac, the gate's independent second algorithm, is test infrastructure (libam_check.so); loaded after the timed region
    # machineValues in flat form: rank 0 built the machine, the others attached to the broadcast image
    if rank == 0:
        voff = np.ascontiguousarray(machine.values_off(), dtype=np.uint64)
        vals = np.ascontiguousarray(machine.values(), dtype=np.uint32)
    if world > 1:
        sizes = torch.tensor([voff.size, vals.size] if rank == 0 else [0, 0], dtype=torch.int64, device=dev)
        dist.broadcast(sizes, 0)
        t_off = torch.from_numpy(voff.view(np.int64)).to(dev) if rank == 0 else torch.empty(int(sizes[0]), dtype=torch.int64, device=dev)
        t_val = torch.from_numpy(vals.astype(np.int64)).to(dev) if rank == 0 else torch.empty(int(sizes[1]), dtype=torch.int64, device=dev)
        dist.broadcast(t_off, 0); dist.broadcast(t_val, 0)
        if rank != 0:
            voff = t_off.cpu().numpy().view(np.uint64).copy()
            vals = t_val.cpu().numpy().astype(np.uint32)
    if vals.size == 0:
        vals = np.zeros(1, np.uint32)
    table = C.c_void_p()
    am.api.check(lib.am_needle_ids_create(handle, voff.ctypes.data, vals.ctypes.data, 0, C.byref(table)))

    def fold(kernel, b=batch, n=n_hay, keep_records=False, keep_handle=False):
        am.api.check(lib.am_automaton_set_kernel(handle, kernel))
        m = C.c_void_p()
        am.api.check(lib.am_run_batch(handle, case, b, C.byref(m)))
        h, c = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64)
        am.api.check(lib.am_matches_fold_hash(m, table, n, h.ctypes.data, c.ctypes.data))
        recs = am.api.matches_to_numpy(m) if keep_records else None
        if keep_handle:
            return h[:n], c[:n], m
        lib.am_matches_free(m)
        return h[:n], c[:n], recs

    def expand(rs):
        """records -> the (matchPos, value) sequence the reference folds: every record's value list in its order"""
        st = rs["state"].astype(np.int64)
        lens = (voff[st + 1] - voff[st]).astype(np.int64)
        gpos = np.repeat(rs["end_pos"], lens)
        gval = np.concatenate([vals[int(voff[x]):int(voff[x + 1])] for x in st]) if len(rs) else np.zeros(0, np.uint32)
        return gpos, gval

    t0 = time.perf_counter()
    primary = args.kernel if args.kernel in (1, 2) else 0
    m1 = None                                                   # rank 0 keeps the primary route's whole result: the full-list checks read single haystacks out of it IN PLACE
    try:
        h1, c1, m1 = fold(primary, keep_handle=(rank == 0))
        other = 2 if primary == 1 else 1
        try:
            h2, c2, _ = fold(other)
            agree = bool(np.array_equal(h1, h2) and np.array_equal(c1, c2))
        except am.AmError as e:                               # an automaton only one kernel can run
            if e.code != am.AM_ERR_UNSUPPORTED:
                raise
            agree = None
    finally:
        am.api.check(lib.am_automaton_set_kernel(handle, args.kernel))
    kernels_s = time.perf_counter() - t0
    if agree is False:
        bad = int(np.flatnonzero((h1 != h2) | (c1 != c2))[0])
        raise SystemExit("PARITY FAILURE: suffix-filter and general kernels fold different match sequences (first at haystack %d of rank %d)" % (bad, rank))
    n_all = amdist_sum([n_hay], dev)[0]
    out = {"hashed": n_all, "kernels_agree": agree, "fold": "am_matches_fold_hash (include/am.h): order-sensitive 64-bit checksum of every haystack's (matchPos, value) sequence"}
    if rank == 0:
        from concurrent.futures import ThreadPoolExecutor
        o, _ = get_oracle(needles)
        hb = w["hay_bytes"]
        budget = args.parity_oracle_mib << 20
        threads = min(host_cores(), 256)
        try:
            if hb <= budget:
                # A SPREAD sample of the batch that was timed (not its first haystacks): haystack 0, the last one, the haystacks around every 2^32-byte offset of the
                # batch (where a 32-bit truncation in the shared indexing code -- offsets, the per-KiB haystack index, the unit counter -- would first show) and evenly
                # spaced others, budget // hay_bytes in all; every one is compared through the whole-batch result, nothing is scanned again in a smaller batch.
                k = max(1, min(n_hay, budget // hb))
                idx, special = spread_sample(n_hay, hb, k)
                got = [(int(h1[i]), int(c1[i])) for i in idx]
                host = {i: text[i * hb:(i + 1) * hb].cpu().numpy() for i in idx}
                t1 = time.perf_counter()
                with ThreadPoolExecutor(threads) as pool:
                    exp = list(pool.map(lambda i: o.fold_hash(case, host[i]), idx))
                oracle_s = time.perf_counter() - t1
                if got != exp:
                    bad = next(idx[j] for j in range(len(idx)) if got[j] != exp[j])
                    raise SystemExit("PARITY FAILURE: device fold checksum differs from the oracle's at haystack %d" % bad)
                what = "%d haystacks spread over the batch: 0, %d, %d around the 2^32-byte offsets, the rest evenly spaced" % (len(idx), n_hay - 1, max(0, len(special) - 2))
                # full lists: the special ones (at most 12) + 1 % of the others, each read out of the whole-batch result where it lies
                others = [i for i in idx if i not in special]
                lists = sorted(set(special[:12]) | set(others[::100][:max(1, len(idx) // 100)]))
                for i in lists:
                    pos, val = o.run_list(case, host[i])
                    gpos, gval = expand(am.api.matches_of_haystack(m1, i))
                    if not (np.array_equal(gpos, pos) and np.array_equal(gval, val)):
                        raise SystemExit("PARITY FAILURE: match list of haystack %d differs from the oracle's" % i)
                out.update({"oracle_checked": len(idx), "oracle_bytes": int(len(idx) * hb), "oracle_what": what, "oracle_cores": threads, "oracle_s": round(oracle_s, 2),
                            "oracle_max_byte_offset": int((max(idx) + 1) * hb), "full_lists_checked": len(lists), "full_lists_what": "read in place out of the whole-batch result (am_matches_haystack_range)",
                            "matches_in_checked": int(sum(c for _, c in exp)), "kernels_s": round(kernels_s, 2)})
            else:
                # ONE document larger than the budget.  (a) its first min(budget, 32 MiB), cut on a code point boundary, scanned as a haystack of its own (what
                # runWithCase reports on a prefix is what it reports on the whole text up to there): fold checksum; (b) its LAST 32 MiB, out of the whole-document
                # result in place: the oracle walks text[tail - 64 KiB:] (a needle is far shorter than that, so every match that ends inside the tail is seen whole and
                # no other state of the automaton matters) and its matches with end positions in the tail must be the records with end_pos > tail, one by one;
                # (c) the full list of the first MiB.  One oracle thread each.
                size = boundary_at_or_before(lambda i: int(text[i]), min(budget, 32 << 20), hb)
                offs = torch.tensor([0, size], dtype=torch.int64, device=dev)
                sb = C.c_void_p()
                am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), 1, size, C.byref(sb)))
                try:
                    hp, cp, _ = fold(primary, sb, 1)
                finally:
                    am.api.check(lib.am_automaton_set_kernel(handle, args.kernel))
                    lib.am_batch_destroy(sb)
                host = text[:size].cpu().numpy()
                t1 = time.perf_counter()
                exp = [o.fold_hash(case, host)]
                if [(int(hp[0]), int(cp[0]))] != exp:
                    raise SystemExit("PARITY FAILURE: device fold checksum differs from the oracle's on the first %d bytes" % size)
                n_rec = int(lib.am_matches_size(m1))

                def record_at(j):
                    one = np.zeros(1, am.api.MATCH_DTYPE)
                    am.api.check(lib.am_matches_copy(m1, j, 1, one.ctypes.data))
                    return one[0]

                def first_after(pos):
                    """index of the first record of the (one-haystack) result whose end_pos > pos"""
                    lo, hi = 0, n_rec
                    while lo < hi:
                        mid = (lo + hi) // 2
                        if int(record_at(mid)["end_pos"]) <= pos: lo = mid + 1
                        else: hi = mid
                    return lo

                tail = boundary_at_or_before(lambda i: int(text[i]), hb - min(32 << 20, hb // 2), hb)
                start = boundary_at_or_before(lambda i: int(text[i]), max(0, tail - (64 << 10)), hb)
                pos, val = o.run_list(case, text[start:hb].cpu().numpy())
                keep = pos + start > tail
                j0 = first_after(tail)
                rs = np.zeros(n_rec - j0, am.api.MATCH_DTYPE)
                if len(rs):
                    am.api.check(lib.am_matches_copy(m1, j0, len(rs), rs.ctypes.data))
                gpos, gval = expand(rs)
                if not (np.array_equal(gpos, pos[keep] + start) and np.array_equal(gval, val[keep])):
                    raise SystemExit("PARITY FAILURE: the match list of the document's last %d bytes differs from the oracle's" % (hb - tail))
                size1 = boundary_at_or_before(lambda i: int(host[i]), min(size, 1 << 20), size)
                pos1, val1 = o.run_list(case, host[:size1])
                j1 = first_after(size1)
                rs1 = np.zeros(j1, am.api.MATCH_DTYPE)
                if j1:
                    am.api.check(lib.am_matches_copy(m1, 0, j1, rs1.ctypes.data))
                gpos1, gval1 = expand(rs1)
                if not (np.array_equal(gpos1, pos1) and np.array_equal(gval1, val1)):
                    raise SystemExit("PARITY FAILURE: the match list of the document's first %d bytes differs from the oracle's" % size1)
                oracle_s = time.perf_counter() - t1
                out.update({"oracle_checked": 1, "oracle_bytes": int(size + (hb - start)),
                            "oracle_what": "the one haystack: its first %d bytes as a haystack of their own (checksum) + the full lists of its first %d and its LAST %d bytes read in place out of the whole-document result" % (size, size1, hb - tail),
                            "oracle_max_byte_offset": int(hb), "oracle_cores": 1, "oracle_s": round(oracle_s, 2), "full_lists_checked": 2,
                            "matches_in_checked": int(exp[0][1]) + int(keep.sum()), "kernels_s": round(kernels_s, 2)})
        finally:
            if m1 is not None:
                lib.am_matches_free(m1)
    lib.am_needle_ids_destroy(table)
    return out


def spread_sample(n_hay, hay_bytes, k):
    """Haystack indices for the oracle: (sorted sample, the special ones first-to-last) -- 0, n_hay - 1, the haystacks around every multiple of 2^32 bytes of the batch, and
    evenly spaced others up to k in all (the special ones are never dropped)."""
    import numpy as np
    special = {0, n_hay - 1}
    total = n_hay * hay_bytes
    for m in range(1, (total >> 32) + 1):
        i = (m << 32) // hay_bytes
        for j in (i - 1, i, i + 1):
            if 0 <= j < n_hay and j * hay_bytes < total:
                special.add(j)
    idx = set(special)
    rest = k - len(idx)
    if rest > 0:
        idx.update(int(x) for x in np.linspace(0, n_hay - 1, rest + 2)[1:-1])
    return sorted(idx), sorted(special)


def boundary_at_or_before(byte_at, size, total):
    """The largest s <= size at which a code point starts (or s == total): a text cut there ends on a whole code point."""
    while 0 < size < total and (byte_at(size) & 0xC0) == 0x80:
        size -= 1
    return size


def amdist_sum(values, dev):
    from alfred_margaret_amd import dist as amdist
    return amdist.allreduce_sum(values, dev)


def pin_to_one_core():
    """The reference's harness runs under `taskset -c 1` (benchmark/benchmark.py:49): pin this thread to core 1 if
    the process may use it, else to the lowest core it may use.  Returns (previous mask, core)."""
    try:
        prev = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        return None, None
    core = 1 if 1 in prev else min(prev)
    try:
        os.sched_setaffinity(0, {core})
    except OSError:
        return None, None
    return prev, core


def cpu_governor(core):
    try:
        with open("/sys/devices/system/cpu/cpu%d/cpufreq/scaling_governor" % (core or 0)) as f:
            return f.read().strip()
    except OSError:
        return "unknown"


def cpu_baseline(args, w, needles, case, hay_cells, handle, batch, lib):
    """The oracle (C restatement of the reference algorithm) timed by the reference's own protocol
    (benchmark/benchmark.py:49,54-55,65-69; benchmark/report.py:13-31; benchmark/haskell/app/Main.hs:62-64,73):
    pinned to ONE core, 5 repetitions of the same bounded sample, each repetition = build + run (the reference times
    both together), identical counts asserted across repetitions, mean +- stdev and min reported, for `run` alone and
    for `build + run`.  The sample = the first k haystacks of rank 0's shard, k chosen so that the five runs fit
    --cpu-seconds.  Then SURVEY 8d (ii): the same oracle on every host core.  GPU counts on the samples must be
    identical (the bench aborts otherwise)."""
    import numpy as np
    from alfred_margaret_amd import synth
    from oracle import oracle
    import alfred_margaret_amd as am
    REPS = 5
    n_hay = int(lib.am_batch_total_bytes(batch)) // w["hay_bytes"]
    box = {}

    def pinned_leg():
        # in a THREAD OF ITS OWN: sched_setaffinity(0) pins the calling thread, and threads created while a thread is pinned inherit its mask -- pinning
        # the main thread would leave every runtime thread HIP starts later (copy-completion handlers ...) on that one core for the rest of the process
        # (seen in round 5: the Replacer's host-result rate fell from 44 to 26 GiB/s in runs that had taken this leg first)
        prev, core = pin_to_one_core()
        try:
            # calibration: one haystack (also warms the page cache of the tables)
            o, build0 = get_oracle(needles)
            hay0 = synth.haystacks_host(needles, w["mixed"], 0, hay_cells, plants=args.plants, natural=bool(w.get("natural")))
            t1 = time.perf_counter(); o.count_matches(case, hay0); per_hay = time.perf_counter() - t1
            k = int(max(1, min(n_hay, 64, (args.cpu_seconds / REPS) / max(per_hay, 1e-6))))
            hays = [hay0] + [synth.haystacks_host(needles, w["mixed"], i * hay_cells, hay_cells, plants=args.plants, natural=bool(w.get("natural"))) for i in range(1, k)]
            run_s, build_s, counts = [], [], None
            for _ in range(REPS):
                t1 = time.perf_counter()
                m = oracle.Machine(needles)                              # build inside the timed region, as the reference does
                t2 = time.perf_counter()
                c = [m.count_matches(case, h) for h in hays]
                t3 = time.perf_counter()
                assert counts is None or counts == c, "oracle should have consistent output"      # benchmark.py:65-69
                counts = c
                build_s.append(t2 - t1); run_s.append(t3 - t2)
                del m
            box.update(o=o, k=k, hays=hays, run_s=run_s, build_s=build_s, counts=counts, core=core)
        except BaseException as e:                                        # noqa: BLE001 (carried to the caller's thread)
            box["error"] = e
        finally:
            if prev is not None:
                os.sched_setaffinity(0, prev)

    import threading
    t = threading.Thread(target=pinned_leg)
    t.start(); t.join()
    if "error" in box:
        raise box["error"]
    o, k, hays, run_s, build_s, counts, core = (box[x] for x in ("o", "k", "hays", "run_s", "build_s", "counts", "core"))
    scanned = sum(h.size for h in hays)
    gpu_counts = np.zeros(n_hay, np.uint64)
    am.api.check(lib.am_count_batch(handle, case, batch, gpu_counts.ctypes.data, None))
    if [int(c) for c in gpu_counts[:k]] != counts:
        raise SystemExit("PARITY FAILURE: GPU counts differ from the oracle on the CPU-baseline sample")
    run_s, both_s = np.array(run_s), np.array(run_s) + np.array(build_s)
    gib = scanned / float(1 << 30)
    out = {"value": round(gib / float(run_s.mean()), 5), "unit": "GiB/s", "cores": 1, "kind": "port",
           "sample": "first %d haystacks (%d MiB) of the same workload; GPU counts on the sample identical" % (k, scanned >> 20),
           "protocol": "pinned to core %s, %d repetitions, counts identical across repetitions (benchmark/benchmark.py:49,54-55,65-69)" % (core, REPS),
           "governor": cpu_governor(core),
           "run_s": {"mean": round(float(run_s.mean()), 4), "stdev": round(float(run_s.std()), 4), "min": round(float(run_s.min()), 4)},
           "build_plus_run_s": {"mean": round(float(both_s.mean()), 4), "stdev": round(float(both_s.std()), 4), "min": round(float(both_s.min()), 4)},
           "build_s_mean": round(float(np.mean(build_s)), 4),
           "value_min_time": round(gib / float(run_s.min()), 5), "value_build_plus_run": round(gib / float(both_s.mean()), 5)}
    # the same oracle on every host core of the box, one haystack per task (ctypes releases the GIL)
    threads = min(host_cores(), 256)
    if threads > 1 and not args.no_cpu_all_cores:
        from concurrent.futures import ThreadPoolExecutor
        per_hay = float(run_s.mean()) / k
        n_tasks = min(n_hay, 1024, max(threads, int(threads * min(args.cpu_seconds, 10.0) / max(per_hay, 1e-6))))
        hays = hays + [synth.haystacks_host(needles, w["mixed"], i * hay_cells, hay_cells, plants=args.plants, natural=bool(w.get("natural"))) for i in range(k, n_tasks)]
        hays = hays[:n_tasks]
        with ThreadPoolExecutor(threads) as pool:
            t1 = time.perf_counter()
            mt_counts = list(pool.map(lambda h: o.count_matches(case, h), hays))
            mt_s = time.perf_counter() - t1
        if [int(c) for c in gpu_counts[:n_tasks]] != mt_counts:
            raise SystemExit("PARITY FAILURE: GPU counts differ from the oracle on the all-cores CPU sample")
        out["all_cores"] = {"value": round(sum(h.size for h in hays) / float(1 << 30) / mt_s, 4), "unit": "GiB/s", "cores": threads,
                            "sample": "%d haystacks (%d MiB), one per task" % (n_tasks, sum(h.size for h in hays) >> 20)}
    return out


if __name__ == "__main__":
    main()
