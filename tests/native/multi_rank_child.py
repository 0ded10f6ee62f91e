"""One RANK of tests/test_multi_ranks.py (test infrastructure): a process of its own that drives libam's one-process-per-GPU entry points
(am_multi_create_rank and everything after it, include/am.h "several GPUs") with the file-based RCCL stand-in of tests/native/rccl_stub.cpp loaded
under RCCL's soname -- N such processes share the box's one GPU.  Writes what it saw to <out>.json; the parent compares with the oracle.

usage: multi_rank_child.py <repo root> <rank> <world> <work dir> <fail_rank>"""
import ctypes as C
import json
import os
import sys
import time

root, rank, world, work, fail_rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
sys.path.insert(0, root)
C.CDLL(os.path.join(root, "alfred-margaret_amd", "lib", "libam_rccl_stub.so"), mode=C.RTLD_GLOBAL)      # before libam looks for "librccl.so.1"

import numpy as np                      # noqa: E402

import alfred_margaret_amd as am        # noqa: E402

lib = am.api.libam()
job = json.load(open(os.path.join(work, "job.json")))
case = job["case"]
needles = job["needles"]
hays = [open(os.path.join(work, "hay_%d" % i), "rb").read() for i in range(job["n_hay"])]
single = open(os.path.join(work, "single"), "rb").read()
out = {"rank": rank}

# ---- the communicator: rank 0 makes the id, the launcher (here: a file) hands it round
id_path = os.path.join(work, "id")
if rank == 0:
    buf = (C.c_uint8 * 128)()
    am.api.check(lib.am_multi_unique_id(buf))
    with open(id_path + ".tmp", "wb") as f:
        f.write(bytes(buf))
    os.rename(id_path + ".tmp", id_path)
t0 = time.time()
while not os.path.exists(id_path):
    if time.time() - t0 > 60:
        raise SystemExit("rank %d: no id from rank 0" % rank)
    time.sleep(0.01)
ident = (C.c_uint8 * 128).from_buffer_copy(open(id_path, "rb").read())
multi = C.c_void_p()
am.api.check(lib.am_multi_create_rank(world, rank, ident, C.byref(multi)))
assert lib.am_multi_world_size(multi) == world and lib.am_multi_local_devices(multi) == 1 and lib.am_multi_device(multi, 0) == 0

# ---- the automaton: built and flattened on rank 0 only; every other rank attaches to the image it RECEIVES and scans with k_sf on it
machine = am.Automaton(needles) if rank == 0 else None
autos = (C.c_void_p * 1)()
am.api.check(lib.am_multi_broadcast_automaton(multi, machine.device if rank == 0 else None, case, 0, autos))
handle = C.c_void_p(autos[0])
out["lower_hash"] = int(lib.am_automaton_lower_hash(handle))

# ---- this rank's block of the haystacks (the launcher shards; libam shards a process's list over its LOCAL devices: one here)
lo, hi = len(hays) * rank // world, len(hays) * (rank + 1) // world
mine = am.api._Slices(hays[lo:hi])
counts = np.zeros(max(hi - lo, 1), np.uint64)
total = C.c_uint64(0)
am.api.check(lib.am_multi_count(multi, autos, case, mine.arr, hi - lo, counts.ctypes.data, C.byref(total)))
out["block"] = [lo, hi]
out["counts"] = [int(c) for c in counts[:hi - lo]]
out["job_total"] = int(total.value)
n = C.c_size_t(0)
p = C.c_void_p()
am.api.check(lib.am_multi_run(multi, autos, case, mine.arr, hi - lo, C.byref(p), C.byref(n)))
arr = np.frombuffer((C.c_char * (n.value * 16)).from_address(p.value), dtype=am.api.MATCH_DTYPE).copy() if n.value else np.zeros(0, am.api.MATCH_DTYPE)
lib.am_multi_matches_free(p)
out["records"] = [[int(r["haystack"]) + lo, int(r["end_pos"]), int(r["state"])] for r in arr]

# ---- device-resident batches + the all-reduce of record counts
b = C.c_void_p()
am.api.check(lib.am_multi_batch_upload(multi, 0, mine.arr, hi - lo, C.byref(b)))
batches = (C.c_void_p * 1)(b)
local_total, job_total = C.c_uint64(0), C.c_uint64(0)
am.api.check(lib.am_multi_count_batch(multi, autos, case, batches, None, C.byref(local_total), C.byref(job_total)))
res = (C.c_void_p * 1)()
n_all = C.c_uint64(0)
am.api.check(lib.am_multi_run_batch(multi, autos, case, batches, res, C.byref(n_all)))
out["resident"] = {"local_total": int(local_total.value), "job_total": int(job_total.value), "local_records": int(lib.am_matches_size(res[0])) if res[0] else 0, "job_records": int(n_all.value)}
lib.am_matches_free(res[0])
sums = np.array([rank + 1, 10 ** 12 + rank], dtype=np.uint64)
am.api.check(lib.am_multi_allreduce_sum(multi, sums.ctypes.data, 2))
out["allreduce"] = [int(x) for x in sums]

# ---- ONE haystack over all ranks: every process passes the whole text, rank r owns the end positions in (len r / W, len (r + 1) / W]
one = am.api._Slices([single])
lc, tot = C.c_uint64(0), C.c_uint64(0)
am.api.check(lib.am_multi_count_single(multi, autos, case, one.arr, C.byref(lc), C.byref(tot)))
p, n, nrec = C.c_void_p(), C.c_size_t(0), C.c_uint64(0)
am.api.check(lib.am_multi_run_single(multi, autos, case, one.arr, C.byref(p), C.byref(n), C.byref(nrec)))
arr = np.frombuffer((C.c_char * (n.value * 16)).from_address(p.value), dtype=am.api.MATCH_DTYPE).copy() if n.value else np.zeros(0, am.api.MATCH_DTYPE)
lib.am_multi_matches_free(p)
out["single"] = {"local_count": int(lc.value), "total": int(tot.value), "records": [[int(r["end_pos"]), int(r["state"])] for r in arr], "job_records": int(nrec.value)}

# ---- a failure on ONE rank must reach every rank through the collective (nobody blocks, everybody returns an error), and the job goes on
bad_autos = (C.c_void_p * 1)(None) if rank == fail_rank else autos
rc = lib.am_multi_count_batch(multi, bad_autos, case, batches, None, None, C.byref(job_total))
out["failure_rc"] = int(rc)
out["failure_msg"] = (lib.am_last_error() or b"").decode("utf-8", "replace")
rc2 = lib.am_multi_count_batch(multi, autos, case, batches, None, None, C.byref(job_total))
out["after_failure"] = [int(rc2), int(job_total.value)]
# the root without its automaton: every rank learns that there is no image (size 0 travels), nobody waits for a blob
autos2 = (C.c_void_p * 1)()
out["broadcast_without_root_rc"] = int(lib.am_multi_broadcast_automaton(multi, None, case, 0, autos2))

lib.am_batch_destroy(b)
lib.am_automaton_destroy(handle)
lib.am_multi_destroy(multi)
with open(os.path.join(work, "out_%d.json.tmp" % rank), "w") as f:
    json.dump(out, f)
os.rename(os.path.join(work, "out_%d.json.tmp" % rank), os.path.join(work, "out_%d.json" % rank))
