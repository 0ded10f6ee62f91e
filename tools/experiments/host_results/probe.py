"""Experiment (round 5): why does am_replacer_run_batch (results to pinned host memory) run at 44 GiB/s in some processes and 26 GiB/s in others?
Times the same call after a series of disturbances.  Run on the GPU box: python tools/experiments/host_results/probe.py"""
import ctypes as C, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
import alfred_margaret_amd as am
from alfred_margaret_amd import synth

lib = am.api.libam()
dev = torch.device("cuda:0")
name = "cfg5_replacer_50k_1GiB"
w = synth.WORKLOADS[name]
pairs = synth.replacer_pairs(name)
rep = am.Replacer(w["case"], pairs)
rdev = C.c_void_p(rep.device)
n_hay = w["n_hay"]; cells = w["hay_bytes"] // synth.CELL
text, n_bytes = synth.haystacks_device([p[0] for p in pairs], w["mixed"], 0, n_hay * cells, dev)
offs = torch.arange(n_hay + 1, dtype=torch.int64, device=dev) * w["hay_bytes"]
batch = C.c_void_p()
am.api.check(lib.am_batch_from_device(text.data_ptr(), offs.data_ptr(), n_hay, n_bytes, C.byref(batch)))
last = {}

def step(host=True):
    if "res" in last: lib.am_replaced_free(last.pop("res"))
    res = C.c_void_p()
    run = lib.am_replacer_run_batch if host else lib.am_replacer_run_batch_device
    am.api.check(run(rdev, batch, C.c_uint64(2**64 - 1), C.byref(res)))
    last["res"] = res

def timed(label, n=4):
    step(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-44s %s ms" % (label, " ".join("%.1f" % t for t in ts)), flush=True)

def wire(label):
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev); dst = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(src, non_blocking=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-44s wire %.1f GiB/s" % (label, 1.0 / best), flush=True)


timed("fresh process")
for which in sys.argv[1:] or ["devmem", "hostmem", "pinthread", "release", "pinned_big", "release"]:
    if which == "devmem":
        x = torch.empty(40 << 30, dtype=torch.uint8, device=dev); x.fill_(1); torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
    elif which == "hostmem":
        y = np.ones(12 << 30, np.uint8); del y
    elif which == "pinthread":
        def leg():
            prev = os.sched_getaffinity(0); os.sched_setaffinity(0, {max(prev)})
            z = np.ones(1 << 30, np.uint8); z.sum(); os.sched_setaffinity(0, prev)
        t = threading.Thread(target=leg); t.start(); t.join()
    elif which == "release":
        if "res" in last: lib.am_replaced_free(last.pop("res"))
        lib.am_release_host_memory()
    elif which == "pinned_big":
        p = torch.empty(6 << 30, dtype=torch.uint8).pin_memory(); del p
    elif which == "sleep":
        time.sleep(3.0)
    elif which == "watch":
        t_end = time.time() + 4.0
        while time.time() < t_end:
            timed("  watch %.1f" % (t_end - time.time()), 1); time.sleep(0.4)
    elif which == "fillonly":
        x = torch.empty(40 << 30, dtype=torch.uint8, device=dev); x.fill_(1); torch.cuda.synchronize()
    elif which == "allocfree":
        x = torch.empty(40 << 30, dtype=torch.uint8, device=dev); torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
    elif which == "count10g":
        big = torch.empty(10 << 30, dtype=torch.uint8, device=dev); big.fill_(97); torch.cuda.synchronize(); del big; torch.cuda.empty_cache()
    timed("after " + which); wire("after " + which)
