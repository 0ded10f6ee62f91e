"""world_size-2 run of the multi-GPU plumbing on CPU (backend gloo): rank 0 flattens the automaton,
the image bytes are broadcast, each rank scans its block of haystacks (with the TEST-ONLY host
interpreter of the image standing in for the HIP kernels -- no GPU here), counts are all-reduced and
must equal the oracle's.  Proves that the image is position independent (usable in another
process as received) and that sharding + collectives are wired correctly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import alfred_margaret_amd as am
        from alfred_margaret_amd import dist as amdist, synth
        from oracle import oracle
        from tests.helpers import ImgCheck, expand_records
        dev = torch.device("cpu")
        needles = [am.lower_utf8(n).decode() for n in synth.make_needles(3000, True)]
        n_hay, hay_cells = 23, 16                      # 23 haystacks: uneven split over 2 ranks
        chk = ImgCheck()
        image = None
        if rank == 0:                                  # only rank 0 builds + flattens
            image = torch.from_numpy(chk.flatten(am.Automaton(needles), am.IGNORE_CASE))
        image = amdist.broadcast_image(image, dev, src=0)
        img = image.numpy()
        lo, hi = amdist.shard_bounds(n_hay, rank, world)
        hays = [synth.haystacks_host(needles, True, h * hay_cells, hay_cells) for h in range(lo, hi)]
        n, recs = chk.scan(img, 1, hays)               # suffix-filter logic on the received image
        o = oracle.Machine(needles)                    # the checker
        vlen = np.diff(o.values_off())
        local_values = int(vlen[recs[1]].sum()) if n else 0
        assert local_values == sum(o.count_matches(am.IGNORE_CASE, h) for h in hays)
        total_values, total_records, total_hays = amdist.allreduce_sum([local_values, n, hi - lo], dev)
        assert total_hays == n_hay
        if rank == 0:
            all_hays = [synth.haystacks_host(needles, True, h * hay_cells, hay_cells) for h in range(n_hay)]
            assert total_values == sum(o.count_matches(am.IGNORE_CASE, h) for h in all_hays)
            assert total_values > n_hay * hay_cells // 2
        assert amdist.allreduce_max(float(rank), dev) == float(world - 1)
        # match lists: gathered on the host in haystack order (rank 0 gets the global list)
        local = np.zeros(max(n, 0), dtype=am.api.MATCH_DTYPE)
        if n:
            local["haystack"], local["state"], local["end_pos"] = recs[0], recs[1], recs[2]
        everything = amdist.gather_records(local, lo, dst=0)
        if rank == 0:
            assert len(everything) == total_records
            keys = list(zip(everything["haystack"].tolist(), everything["end_pos"].tolist()))
            assert keys == sorted(keys) and everything["haystack"].max() == n_hay - 1
            exp_all = [(i, int(p), int(v)) for i, h in enumerate(all_hays) for p, v in zip(*o.run_list(am.IGNORE_CASE, h))]
            assert expand_records(o.values_off(), o.values(), everything["haystack"], everything["state"], everything["end_pos"]) == exp_all
        else:
            assert everything is None
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("%d %d" % (total_values, total_records))
    finally:
        dist.destroy_process_group()


def test_two_ranks_broadcast_shard_allreduce(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [open(os.path.join(tmp_path, "ok%d" % r)).read() for r in range(world)]
    assert got[0] == got[1]


def test_shard_bounds_cover_everything():
    from alfred_margaret_amd import dist as amdist
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 4, 8):
            spans = [amdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
