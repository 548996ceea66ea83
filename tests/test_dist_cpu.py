"""Multi-GPU path on CPU: world_size-2 gloo processes exercise frame sharding and the fixed-shape
all-gather of padded detections (the RCCL collective of the GPU path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from focalformer3d_amd import dist as fdist


def _frames(n, M=6, D=9, seed=0):
    g = torch.Generator().manual_seed(seed)
    boxes = torch.randn(n, M, D, generator=g)
    scores = torch.rand(n, M, generator=g)
    labels = torch.randint(0, 10, (n, M), generator=g, dtype=torch.int32)
    count = torch.randint(0, M + 1, (n,), generator=g, dtype=torch.int32)
    return boxes, scores, labels, count


def test_shard_range_partitions_all_frames():
    for n in (0, 1, 7, 32, 33):
        for w in (1, 2, 3, 8):
            spans = [fdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip():
    b, s, l, c = _frames(5)
    res = fdist.unpack_detections(fdist.pack_detections(b, s, l, c))
    for i, (bb, ss, ll) in enumerate(res):
        n = int(c[i])
        assert torch.equal(bb, b[i, :n]) and torch.equal(ss, s[i, :n]) and torch.equal(ll, l[i, :n])
    b7 = b[..., :7].contiguous()                       # no-velocity boxes (Waymo): zero padded to 9 columns
    res7 = fdist.unpack_detections(fdist.pack_detections(b7, s, l, c))
    assert all(r[0].shape[1] == 7 for r in res7)


def _worker(rank, world, port, total, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        b, s, l, c = _frames(total, seed=42)            # the full job, identical on every rank
        lo, hi = fdist.shard_range(total, rank, world)
        gathered = fdist.gather_detections(b[lo:hi], s[lo:hi], l[lo:hi], c[lo:hi])
        expect = fdist.pack_detections(b, s, l, c)
        ok = torch.equal(gathered, expect)
        # the bench's pipelined form: two batches through the two-slot gather, results in rank order
        ag = fdist.AsyncDetectionGather(hi - lo, b.shape[1], 'cpu')
        ag.submit(b[lo:hi] * 2, s[lo:hi], l[lo:hi], c[lo:hi])
        ag.submit(b[lo:hi], s[lo:hi], l[lo:hi], c[lo:hi])
        ok = ok and torch.equal(ag.result(), expect)
        t = torch.tensor([1.0 + rank])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)        # the bench's max-over-ranks timing reduction
        # bench.py's check of the exchange itself (round 6): own rows at [rank * B, (rank + 1) * B), identical record on every rank
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        own = fdist.pack_detections(b[lo:hi], s[lo:hi], l[lo:hi], c[lo:hi])
        rec = bench.gathered_record_check(gathered, own, rank)
        ok = ok and rec['gathered_rows_of_this_rank_are_its_own'] and rec['gathered_record_identical_on_every_rank'] \
            and rec['ranks_in_the_check'] == world and rec['gathered_frames'] == total
        # a rank that holds a DIFFERENT record (two frames swapped: same multiset of rows) must be noticed by every rank
        bad = gathered.clone()
        if rank == 1:
            bad[[0, 1]] = bad[[1, 0]]
        rec2 = bench.gathered_record_check(bad, own, rank)
        ok = ok and rec2['gathered_record_identical_on_every_rank'] is False
        # ... and a rank whose own rows are not where they belong fails the first check on all ranks (MIN-reduced)
        rec3 = bench.gathered_record_check(gathered, own * (2.0 if rank == 0 else 1.0), rank)
        ok = ok and rec3['gathered_rows_of_this_rank_are_its_own'] is False
        ret[rank] = bool(ok and t.item() == float(world))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_gather_of_detections():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    world, total = 2, 8
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


@pytest.mark.timeout(300)
def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` without a launcher around it must start two ranks itself (tools/dist_test.sh:9-11's
    role).  There is no GPU here, so every rank stops at the device check - which proves the re-exec happened."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=280)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout
    else:
        assert r.returncode != 0
        assert 'bench.py needs an MI355X' in r.stderr or 'device(s) visible' in r.stderr
