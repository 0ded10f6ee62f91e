"""csrc/am_multi.cpp with 2 / 4 / 8 RANKS on a box with one GPU: one process per rank -- am_multi_create_rank, the image broadcast
and attach, block-sharded counts and records, device-resident batches, the count all-reduce, one haystack in ranges, and the flag word that carries one
rank's failure to all -- with RCCL replaced by the file-based stand-in of tests/native/rccl_stub.cpp (loaded under RCCL's soname, so libam's dlopen
table binds it).  Every line of am_multi.cpp's rank path runs; what stays unexecuted on this box are the real ncclBroadcast / ncclAllReduce.
Also the pairing "an image made by ANOTHER process -> k_sf": ranks > 0 never see the needles, only the image they receive (item 6c; the file-borne
variant is test_image_written_by_one_process_scanned_by_another)."""
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

import alfred_margaret_amd as am
from alfred_margaret_amd import synth
from oracle import oracle
from tests.conftest import ROOT
from tests.helpers import expand_records, oracle_triples

CHILD = os.path.join(ROOT, "tests", "native", "multi_rank_child.py")


def test_rccl_stub_builds_and_exports_what_libam_binds():
    """(CPU) the stand-in has RCCL's soname and exactly the nine entry points of am_multi.cpp's dlopen table."""
    path = am.build.build_rccl_stub()
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    got = sorted(line.split()[-1] for line in out.splitlines() if " T " in line)
    src = open(os.path.join(ROOT, "alfred-margaret_amd", "csrc", "am_multi.cpp")).read()
    import re
    bound = sorted(set(re.findall(r'sym\("(nccl[A-Za-z]+)"\)', src)))
    assert got == bound and len(bound) == 9
    assert "librccl.so.1" in subprocess.check_output(["readelf", "-d", path], text=True)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rank_processes_share_one_gpu(tmp_path, world):
    am.build.build_rccl_stub()
    needles = synth.needles_for("cfg3_runLower_100k_10GiB")[:3000] + ["k", "straße"]
    n_hay, hay_cells = 21, 4                                   # 21 divides by none of 2, 4, 8: uneven blocks, and with 8 ranks some get 2, some 3
    text = synth.haystacks_host(needles[:3000], True, 11, n_hay * hay_cells)
    hb = hay_cells * 1024
    hays = [bytes(text[i * hb:(i + 1) * hb]) for i in range(n_hay)]
    hays[5] = b""                                              # an empty haystack inside a block
    single = ("KKk Straße " * 50).encode() + bytes(text[:40 * 1024])
    work = str(tmp_path)
    json.dump({"case": 1, "needles": needles, "n_hay": n_hay}, open(os.path.join(work, "job.json"), "w"))
    for i, h in enumerate(hays):
        open(os.path.join(work, "hay_%d" % i), "wb").write(h)
    open(os.path.join(work, "single"), "wb").write(single)
    fail_rank = world - 1
    procs = [subprocess.Popen([sys.executable, CHILD, ROOT, str(r), str(world), work, str(fail_rank)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    try:
        logs = []
        for p in procs:
            try:
                logs.append(p.communicate(timeout=240)[0])
            except subprocess.TimeoutExpired:
                p.kill()
                logs.append("TIMEOUT\n" + p.communicate()[0])
        assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    finally:
        for d in glob.glob("/dev/shm/am_rccl_stub_*"):
            shutil.rmtree(d, ignore_errors=True)
    outs = [json.load(open(os.path.join(work, "out_%d.json" % r))) for r in range(world)]

    m = oracle.Machine(needles)
    exp_counts = [m.count_matches(1, h) for h in hays]
    # block bounds: contiguous, in rank order, covering everything, differing by at most one haystack
    bounds = [o["block"] for o in outs]
    assert bounds[0][0] == 0 and bounds[-1][1] == n_hay and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    assert max(b[1] - b[0] for b in bounds) - min(b[1] - b[0] for b in bounds) <= 1
    # every rank: its own counts, and the SAME job total (the all-reduce)
    assert sum((o["counts"] for o in outs), []) == exp_counts
    assert {o["job_total"] for o in outs} == {sum(exp_counts)}
    assert {o["lower_hash"] for o in outs} == {am.lower_table_hash(None)}            # the attached images carry the table they were baked with
    recs = np.array(sum((o["records"] for o in outs), []), dtype=np.int64).reshape(-1, 3)
    got = expand_records(m.values_off(), m.values(), recs[:, 0], recs[:, 2], recs[:, 1])
    assert got == oracle_triples(m, 1, hays) and len(got) > 50
    for o in outs:
        lo, hi = o["block"]
        assert o["resident"]["local_total"] == sum(exp_counts[lo:hi]) and o["resident"]["job_total"] == sum(exp_counts)
        assert o["resident"]["job_records"] == len(recs) and o["resident"]["local_records"] == int(((recs[:, 0] >= lo) & (recs[:, 0] < hi)).sum())
        assert o["allreduce"] == [world * (world + 1) // 2, world * 10 ** 12 + world * (world - 1) // 2]
    # one haystack in ranges: the ranks' records, in rank order, are the whole document's
    pos, val = m.run_list(1, single)
    srecs = np.array(sum((o["single"]["records"] for o in outs), []), dtype=np.int64).reshape(-1, 2)
    got1 = expand_records(m.values_off(), m.values(), np.zeros(len(srecs), np.int64), srecs[:, 1], srecs[:, 0])
    assert [(p_, v_) for _, p_, v_ in got1] == [(int(p_), int(v_)) for p_, v_ in zip(pos, val)]
    assert {o["single"]["total"] for o in outs} == {len(pos)} and sum(o["single"]["local_count"] for o in outs) == len(pos)
    assert {o["single"]["job_records"] for o in outs} == {len(srecs)}
    # the injected failure: EVERY rank returns an error (none blocked: all processes came back), the failing one with its own message, the
    # others with "a peer failed"; the next collective works again
    for o in outs:
        assert o["failure_rc"] != 0, o
        assert ("null automaton" in o["failure_msg"]) == (o["rank"] == fail_rank), o["failure_msg"]
        assert o["after_failure"] == [0, sum(exp_counts)]
        assert o["broadcast_without_root_rc"] != 0


@pytest.mark.gpu
def test_image_written_by_one_process_scanned_by_another(tmp_path):
    """process A flattens and writes the serialised image (am_automaton_image_read); process B -- which never sees the needles --
    loads the file (am_automaton_from_host_image: checksum, bounds, table hash) and scans with k_sf; the records are the oracle's."""
    needles = synth.needles_for("cfg3_runLower_100k_10GiB")[:5000]
    text = bytes(synth.haystacks_host(needles, True, 7, 256))
    img_path, hay_path, out_path = (str(tmp_path / n) for n in ("image.bin", "hay.bin", "recs.bin"))
    open(hay_path, "wb").write(text)
    writer = ("import sys; sys.path.insert(0, %r)\n"
              "import ctypes as C, json, alfred_margaret_amd as am\n"
              "lib = am.api.libam(); a = am.Automaton(json.load(open(%r)))\n"
              "n = C.c_size_t(0); am.api.check(lib.am_automaton_image_size(a.device, 1, C.byref(n)))\n"
              "buf = C.create_string_buffer(n.value); am.api.check(lib.am_automaton_image_read(a.device, 1, buf, n.value))\n"
              "open(%r, 'wb').write(buf.raw)\n" % (ROOT, str(tmp_path / "needles.json"), img_path))
    reader = ("import sys; sys.path.insert(0, %r)\n"
              "import ctypes as C, numpy as np, alfred_margaret_amd as am\n"
              "lib = am.api.libam(); img = open(%r, 'rb').read(); h = C.c_void_p()\n"
              "am.api.check(lib.am_automaton_from_host_image(img, len(img), C.byref(h)))\n"
              "am.api.check(lib.am_automaton_set_kernel(h, 2))\n"
              "s = am.api._Slices([open(%r, 'rb').read()]); m = C.c_void_p()\n"
              "am.api.check(lib.am_run(h, 1, s.arr, 1, C.byref(m)))\n"
              "am.api.matches_to_numpy(m).tofile(%r)\n" % (ROOT, img_path, hay_path, out_path))
    json.dump(needles, open(str(tmp_path / "needles.json"), "w"))
    subprocess.check_call([sys.executable, "-c", writer], timeout=240)
    subprocess.check_call([sys.executable, "-c", reader], timeout=240)
    recs = np.fromfile(out_path, dtype=am.api.MATCH_DTYPE)
    m = oracle.Machine(needles)
    got = expand_records(m.values_off(), m.values(), recs["haystack"], recs["state"], recs["end_pos"])
    assert got == oracle_triples(m, 1, [text]) and len(got) > 100


@pytest.mark.gpu
def test_bench_n_rank_path_with_the_stand_in(tmp_path):
    """bench.py's own N-rank code (what the driver launches for SCALE: torch.distributed.run, one process per rank, am_multi_create_rank, image
    broadcast, per-rank shard, all-reduce, the `rccl` object of the JSON line) with two ranks on this box's one GPU: torch.distributed over gloo for the
    128-byte id, libam's collectives bound to the stand-in through AM_RCCL_LIBRARY.  cfg4 (BASELINE configs[3], strong scaling: ONE batch block-sharded
    over the ranks) reduced to 4096 haystacks (--total-haystacks)."""
    env = dict(os.environ, AM_BENCH_SAME_DEVICE="1", AM_BENCH_BACKEND="gloo", AM_RCCL_LIBRARY=am.build.build_rccl_stub(),
               MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29617",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg4_100k_1M_haystacks", "--total-haystacks", "4096", "--parity-oracle-mib", "16"]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    finally:
        for d in glob.glob("/dev/shm/am_rccl_stub_*"):
            shutil.rmtree(d, ignore_errors=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["collectives"] == "libam-rccl"
    assert out["rccl"]["ranks"] == 2 and out["rccl"]["image_broadcast_ms"] > 0 and out["rccl"]["allreduce_ms"] > 0 and out["rccl"]["data_path_collectives"] == 0
    assert out["config"]["haystacks_per_gpu"] == 2048 and out["parity"]["hashed"] == 4096 and out["parity"]["kernels_agree"] is True
    assert out["parity"]["oracle_checked"] >= 100
