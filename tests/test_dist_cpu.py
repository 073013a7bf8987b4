"""Multi-process (world_size 2, gloo, CPU) tests of the clip-level data-parallel path: the single
arena broadcast, the conditioning broadcast and the batch sharding.  The HIP library is not
involved - this checks the collective plumbing the 8-GPU bench relies on."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT  # noqa: F401  (also puts the repo root on sys.path)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from foley_amd.host import config as C, distributed as D, packers, sampler, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = C.TINY
        arena = cond = None
        if rank == 0:
            sd = synth.synth_dit_state_dict(cfg)
            arena = packers.Arena.from_packed(packers.pack_dit(sd, cfg, torch.bfloat16), "cpu")
            cond = synth.synth_conditioning(cfg, 1.0, t2a=False)
        arena = D.broadcast_arena(arena, "cpu")
        cond = D.broadcast_tensors(cond, "cpu")
        # every rank must now hold bit-identical weights / conditioning
        ref = packers.pack_dit(synth.synth_dit_state_dict(cfg), cfg, torch.bfloat16)
        same = all(torch.equal(arena.view(k), v) for k, v in ref.items())
        cref = synth.synth_conditioning(cfg, 1.0, t2a=False)
        same = same and all(torch.equal(cond[k], cref[k]) for k in cref)
        # batch sharding: same CPU-generator draw on every rank, disjoint contiguous slices
        gen = torch.Generator("cpu").manual_seed(1234)
        noise = sampler.draw_noise(5, 128, 50, torch.float32, gen)
        lo, hi = D.shard_range(5, rank, world)
        q.put((rank, same, lo, hi, float(noise[lo:hi].double().sum()), float(noise.double().sum())))
    finally:
        dist.destroy_process_group()


def test_world2_broadcast_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "broadcast arena / conditioning differ from rank 0's"
    (_, _, lo0, hi0, s0, tot0), (_, _, lo1, hi1, s1, tot1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 3, 3, 5)
    assert tot0 == tot1 and abs((s0 + s1) - tot0) < 1e-9


def test_shard_range_covers_batch():
    from foley_amd.host import distributed as D
    for total in (1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            spans = [D.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
