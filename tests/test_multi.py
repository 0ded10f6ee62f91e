"""Several GPUs behind the C ABI (include/am.h am_multi_*): automaton broadcast + block-sharded scan + count
all-reduce inside libam, one process.  The driver's GPU box has ONE device, so the -m gpu tests run the degenerate
1-device path (RCCL communicator of size 1, same code); tests/c/multi_driver.c takes every visible device and must print
the same answer for any device count (its 2..8-device runs skip where the box has fewer).  Thread concurrency of the
single-device ABI is tested here too: calls from different host threads go to different HIP streams."""
import ctypes as C
import os
import subprocess
import threading
import time

import numpy as np
import pytest

import alfred_margaret_amd as am
from alfred_margaret_amd import synth
from oracle import oracle
from tests.conftest import ROOT
from tests.helpers import expand_records, oracle_triples


def _build_driver(tmp_path):
    exe = str(tmp_path / "multi_driver")
    lib = am.build.LIB
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "multi_driver.c"),
                           "-L", lib, "-lam", "-Wl,-rpath," + lib, "-o", exe])
    return exe


def _dump(tmp_path, needles, text):
    m = oracle.Machine(needles)
    m.transitions().tofile(tmp_path / "tr"); m.offsets().tofile(tmp_path / "of"); m.root_ascii().tofile(tmp_path / "ra")
    np.diff(m.values_off()).astype(np.uint32).tofile(tmp_path / "vl")
    (tmp_path / "hay").write_bytes(bytes(text))
    return m, [str(tmp_path / n) for n in ("tr", "of", "ra", "vl", "hay")]


def test_multi_driver_links_without_a_gpu(tmp_path):
    import torch
    am.api.libam()
    exe = _build_driver(tmp_path)
    _, files = _dump(tmp_path, ["tshirt", "shirts", "shorts"], b"short tshirts!!!")
    p = subprocess.run([exe] + files + ["2", "0"], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert p.returncode == 0, p.stderr
    else:
        assert p.returncode == 2 and "no HIP device" in p.stderr, (p.returncode, p.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("n_devices", [1, 2, 4, 8])
def test_multi_driver_same_answer_for_any_device_count(tmp_path, n_devices):
    """The plain-C consumer on 1 / 2 / 4 / 8 devices: total, per-haystack counts and the concatenated record list equal
    the oracle's, whatever the device count.  Runs with more devices than the box has are skipped (exit code 3)."""
    exe = _build_driver(tmp_path)
    workload = "cfg3_runLower_100k_10GiB"
    needles = synth.needles_for(workload)[:4000]
    n_hay, hay_cells = 37, 8                                   # 37 does not divide by 2, 4 or 8: uneven blocks
    text = synth.haystacks_host(needles, True, 3, n_hay * hay_cells)
    m, files = _dump(tmp_path, needles, text)
    p = subprocess.run([exe] + files + [str(n_hay), "1", str(n_devices)], capture_output=True, text=True)
    if p.returncode == 3:
        pytest.skip("box has fewer than %d devices" % n_devices)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.split("\n")
    while lines and not lines[0].startswith("devices "):       # RCCL prints a version banner on stdout when a communicator is made
        lines.pop(0)
    assert lines[0] == "devices %d" % n_devices
    hb = hay_cells * 1024
    hays = [text[i * hb:(i + 1) * hb] for i in range(n_hay)]
    exp_counts = [m.count_matches(1, h) for h in hays]
    assert lines[1] == "total %d" % sum(exp_counts)
    assert [int(x) for x in lines[2].split()[1:]] == exp_counts
    k = int(lines[3].split()[1])
    recs = np.array([[int(x) for x in l.split()] for l in lines[4:4 + k]], dtype=np.int64).reshape(k, 3)
    got = expand_records(m.values_off(), m.values(), recs[:, 0], recs[:, 2], recs[:, 1])
    assert got == oracle_triples(m, 1, hays) and k > 100
    # the device-resident entry points (am_multi_batch_upload / am_multi_count_batch / am_multi_run_batch) gave the same job the same answer
    assert lines[4 + k] == "resident ok", lines[4 + k:]
    # ONE haystack (the whole text as a single document) cut into n_devices ranges: am_multi_run_single / am_multi_count_single == a one-device
    # scan of the document, and that count is the oracle's for the document
    assert lines[5 + k].startswith("single ok"), lines[5 + k:]
    assert int(lines[5 + k].split("(")[1].split()[2]) == m.count_matches(1, bytes(text))


@pytest.mark.gpu
def test_multi_entry_points_one_device():
    """am_multi_* from Python on the box's single device: am_multi_create(0) (all visible devices) and the rank flavour
    (am_multi_unique_id + am_multi_create_rank with one rank): broadcast, count, run, all-reduce."""
    lib = am.api.libam()
    needles = synth.needles_for("cfg2_runText_10k_1GiB")[:3000]
    a = am.Automaton(needles)
    o = oracle.Machine(needles)
    hays = [bytes(synth.haystacks_host(needles, False, 16 * i, 16)) for i in range(9)] + [b"", b"abc"]
    exp = oracle_triples(o, 0, hays)
    exp_counts = [o.count_matches(0, h) for h in hays]
    ident = (C.c_uint8 * 128)()
    am.api.check(lib.am_multi_unique_id(ident))
    for flavour in ("all", "rank"):
        m = C.c_void_p()
        if flavour == "all":
            am.api.check(lib.am_multi_create(0, C.byref(m)))
        else:
            am.api.check(lib.am_multi_create_rank(1, 0, ident, C.byref(m)))
        try:
            D = lib.am_multi_local_devices(m)
            assert D >= 1 and lib.am_multi_world_size(m) == D and lib.am_multi_device(m, 0) == 0
            autos = (C.c_void_p * D)()
            am.api.check(lib.am_multi_broadcast_automaton(m, a.device, 0, 0, autos))
            s = am.api._Slices(hays)
            counts, total = np.zeros(len(hays), np.uint64), C.c_uint64(0)
            am.api.check(lib.am_multi_count(m, autos, 0, s.arr, s.n, counts.ctypes.data, C.byref(total)))
            assert [int(c) for c in counts] == exp_counts and int(total.value) == sum(exp_counts)
            p, n = C.c_void_p(), C.c_size_t(0)
            am.api.check(lib.am_multi_run(m, autos, 0, s.arr, s.n, C.byref(p), C.byref(n)))
            recs = np.frombuffer((C.c_char * (n.value * am.api.MATCH_DTYPE.itemsize)).from_address(p.value), dtype=am.api.MATCH_DTYPE).copy() if n.value else np.zeros(0, am.api.MATCH_DTYPE)
            lib.am_multi_matches_free(p)
            assert expand_records(o.values_off(), o.values(), recs["haystack"], recs["state"], recs["end_pos"]) == exp
            # the device-resident forms: block i of the haystacks uploaded onto local device i, records left there, counts all-reduced
            batches, results, dcounts = (C.c_void_p * D)(), (C.c_void_p * D)(), []
            bounds = [(len(hays) * i // D, len(hays) * (i + 1) // D) for i in range(D)]
            for i, (lo, hi) in enumerate(bounds):
                part = am.api._Slices(hays[lo:hi])
                bh = C.c_void_p()
                am.api.check(lib.am_multi_batch_upload(m, i, part.arr, part.n, C.byref(bh)))
                batches[i] = bh.value
                dcounts.append(np.zeros(hi - lo + 1, np.uint64))
            cptrs = (C.c_void_p * D)(*[c.ctypes.data for c in dcounts])
            local_totals, job_total, n_recs = (C.c_uint64 * D)(), C.c_uint64(0), C.c_uint64(0)
            am.api.check(lib.am_multi_count_batch(m, autos, 0, batches, cptrs, local_totals, C.byref(job_total)))
            assert int(job_total.value) == sum(exp_counts) == sum(int(x) for x in local_totals)
            assert [int(c) for i, (lo, hi) in enumerate(bounds) for c in dcounts[i][:hi - lo]] == exp_counts
            am.api.check(lib.am_multi_run_batch(m, autos, 0, batches, results, C.byref(n_recs)))
            got = []
            for i, (lo, hi) in enumerate(bounds):
                k = int(lib.am_matches_size(results[i]))
                ptr = lib.am_matches_data(results[i])
                r = np.frombuffer((C.c_char * (k * am.api.MATCH_DTYPE.itemsize)).from_address(ptr), dtype=am.api.MATCH_DTYPE).copy() if k else np.zeros(0, am.api.MATCH_DTYPE)
                got += expand_records(o.values_off(), o.values(), r["haystack"] + lo, r["state"], r["end_pos"])
                lib.am_matches_free(results[i]); lib.am_batch_destroy(batches[i])
            assert got == exp and int(n_recs.value) == len(recs)
            # a NULL batch is "no work for that device", and a bad argument comes back as an error from every entry point (no hang)
            none = (C.c_void_p * D)()
            am.api.check(lib.am_multi_count_batch(m, autos, 0, none, None, None, C.byref(job_total)))
            assert int(job_total.value) == 0
            assert lib.am_multi_batch_upload(m, D, s.arr, s.n, C.byref(C.c_void_p())) == am.AM_ERR_INVALID
            vals = np.arange(D * 3, dtype=np.uint64) + 5
            before = vals.copy()
            am.api.check(lib.am_multi_allreduce_sum(m, vals.ctypes.data, 3))
            assert np.array_equal(vals.reshape(D, 3), np.tile(before.reshape(D, 3).sum(axis=0), (D, 1)))
            for i in range(D):
                lib.am_automaton_destroy(autos[i])
        finally:
            lib.am_multi_destroy(m)


@pytest.mark.gpu
def test_calls_from_two_threads_overlap():
    """include/am.h: every calling thread launches on its own HIP stream and one-shot calls share no lock.  Thread A
    scans 1 GiB with the general kernel (tens of milliseconds, many short workgroups); thread B starts a moment later
    and makes 20 small am_count calls.  With a process-wide stream or lock B's first call could only complete after A's
    kernel; on per-thread streams B's calls complete (with the right answers) while A is still running.  Asserted: the two
    threads (and the main thread) launch on three different streams, am_set_stream only affects its own thread, all answers
    are right, AND B's calls finish inside A's call: small batches take k_sf's light configuration (4-wavefront workgroups),
    which gets onto a CU next to A's short workgroups like any of them (round 2's 16-wavefront workgroups sometimes waited for
    A's whole grid to drain, and this test could only report that as xfail).  One thing libam cannot control: HIP multiplexes streams
    onto a few hardware queues; when both threads' streams share one, B runs behind A.  The test then gives B a fresh stream
    (am_set_stream) and tries again, up to five times."""
    import torch
    lib = am.api.libam()
    needles = synth.needles_for("cfg2_runText_10k_1GiB")[:2000]
    big, small = am.Automaton(needles), am.Automaton(needles)
    big.set_kernel(1)
    o = oracle.Machine(needles)
    text = bytes(synth.haystacks_host(needles, False, 0, 16))
    exp = o.count_matches(0, text)
    n_cells, hay_cells = 1 << 20, 1024                          # 1 GiB in 1-MiB haystacks
    dev_text, n_bytes = synth.haystacks_device(needles, False, 0, n_cells, torch.device("cuda:0"))
    offs = torch.arange(n_cells // hay_cells + 1, dtype=torch.int64, device="cuda:0") * (hay_cells * 1024)
    batch = C.c_void_p()
    am.api.check(lib.am_batch_from_device(dev_text.data_ptr(), offs.data_ptr(), n_cells // hay_cells, n_bytes, C.byref(batch)))
    stamps = {}

    attempts, errors = 5, []

    def judge():                                                    # once per attempt, when both threads have finished it
        stamps["inside"] = sum(1 for t in stamps["b_done"] if stamps["a0"] < t < stamps["a1"])     # B's calls that completed while A's call was running
        stamps["a_ms"] = (stamps["a1"] - stamps["a0"]) * 1e3

    start, end = threading.Barrier(2), threading.Barrier(2, action=judge)

    def guarded(body):
        def run():
            try:
                body()
            except BaseException as e:                              # noqa: BLE001 -- reported by the main thread
                errors.append(e)
                start.abort(); end.abort()
        return run

    def my_stream():
        st = C.c_void_p()
        am.api.check(lib.am_get_stream(C.byref(st)))
        return st.value

    def thread_a():
        total = C.c_uint64(0)
        am.api.check(lib.am_count_batch(big.device, 0, batch, None, C.byref(total)))     # warm-up ON THIS THREAD: its stream and workspaces exist afterwards
        stamps["a_stream"] = my_stream()
        for _ in range(attempts):
            start.wait(120)
            stamps["a0"] = time.perf_counter()
            am.api.check(lib.am_count_batch(big.device, 0, batch, None, C.byref(total)))
            stamps["a1"] = time.perf_counter()
            stamps["a_total"] = int(total.value)
            end.wait(120)
            if stamps["inside"] >= 5:
                break

    def thread_b():
        s = am.api._Slices([text])
        c = np.zeros(1, np.uint64)
        for _ in range(3):                                          # warm-up on this thread: a first call allocates, and hipMalloc may wait for running kernels
            am.api.check(lib.am_count(small.device, 0, s.arr, 1, c.ctypes.data))
        stamps["b_stream"] = my_stream()
        side = torch.cuda.Stream()
        am.api.check(lib.am_set_stream(C.c_void_p(side.cuda_stream)))          # per calling thread: A keeps its own
        stamps["b_set"] = my_stream() == side.cuda_stream
        am.api.check(lib.am_set_stream(None))
        extra = []
        for attempt in range(attempts):
            if attempt > 0:
                # no overlap in the previous attempt: HIP maps streams onto a few hardware queues round-robin (4 by default), and two
                # streams that land on the SAME hardware queue run in order whatever libam does.  A fresh stream takes the next queue.
                extra.append(torch.cuda.Stream())
                am.api.check(lib.am_set_stream(C.c_void_p(extra[-1].cuda_stream)))
            start.wait(120)
            time.sleep(0.004)                                       # let A's kernel get going
            done = []
            for _ in range(20):
                am.api.check(lib.am_count(small.device, 0, s.arr, 1, c.ctypes.data))
                assert int(c[0]) == exp
                done.append(time.perf_counter())
            stamps["b_done"] = done
            end.wait(120)
            if stamps["inside"] >= 5:
                break

    try:
        ts = [threading.Thread(target=guarded(thread_a)), threading.Thread(target=guarded(thread_b))]
        for t in ts: t.start()
        for t in ts: t.join()
        assert not errors, errors
        inside, a_ms = stamps["inside"], stamps["a_ms"]
        assert stamps["a_total"] > n_cells * 0.9
        # the mechanism, deterministic: one library stream per calling thread, am_set_stream local to its thread
        assert stamps["a_stream"] and stamps["b_stream"] and stamps["a_stream"] != stamps["b_stream"] != my_stream()
        assert stamps["b_set"]
        # its effect: B's small calls (light configuration: 4-wavefront workgroups) complete while A's kernel is running
        assert inside >= 5, "no overlap in %d attempts (A's call %.1f ms, %d of B's 20 calls inside)" % (attempts, a_ms, inside)
    finally:
        lib.am_batch_destroy(batch)


@pytest.mark.gpu
@pytest.mark.parametrize("single_process", [False, True])
def test_bench_line_reduced(single_process):
    """bench.py end to end at a reduced size: the JSON line carries the metric, roofline, the parity gate (fold checksum:
    suffix-filter kernel == general kernel on every haystack, == oracle on the sample) and cpu_baseline; with
    AM_BENCH_SINGLE_PROCESS=1 the same workload goes through am_multi_create (one process driving the devices)."""
    import json
    import sys
    env = dict(os.environ)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--hay-count", "96", "--steps", "2", "--warmup", "1"]
    if single_process:
        env["AM_BENCH_SINGLE_PROCESS"] = "1"
    else:
        cmd += ["--cpu-seconds", "2", "--parity-oracle-mib", "16"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, lines                                # ONE JSON line on stdout
    d = json.loads(lines[0])
    assert d["unit"] == "GiB/s" and d["n_gpus"] == 1 and d["value"] > 60 and d["dtype"] == "u8"
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    if single_process:
        assert "am_multi_create" in d["config"]["parallelism"]
    else:
        assert d["parity"]["hashed"] == 96 and d["parity"]["kernels_agree"] is True and d["parity"]["oracle_checked"] == 16
        assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["run_s"]["min"] > 0
