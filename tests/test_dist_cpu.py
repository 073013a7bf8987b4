"""Multi-process (world_size 2, gloo, CPU) tests of the clip-level data-parallel path: the single
bundle broadcast, the layout every rank derives locally, the batch sharding - both through
host/distributed.py directly and through bench.py's OWN rank-spawning code (`python bench.py
--gpus 2` with no launcher).  The HIP library is not involved - this checks the collective
plumbing the 8-GPU bench relies on."""
import json

import pytest
import os
import socket
import subprocess
import sys

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
    n_coll = [0]
    real_bcast = dist.broadcast

    def counting_broadcast(*a, **k):
        n_coll[0] += 1
        return real_bcast(*a, **k)

    dist.broadcast = counting_broadcast
    try:
        cfg, dcfg = C.TINY, C.DAC_TINY
        spec = D.bundle_spec(cfg, dcfg, torch.bfloat16, 1.0)          # computed locally on every rank
        bundle = D.Bundle(spec, "cpu")
        if rank == 0:
            sd = synth.synth_dit_state_dict(cfg)
            cond = synth.synth_conditioning(cfg, 1.0, t2a=False)
            bundle.fill(packers.pack_dit(sd, cfg, torch.bfloat16),
                        packers.pack_dac(synth.synth_dac_state_dict(dcfg), dcfg), cond)
        else:
            bundle.buffer.fill_(0xAB)
        D.broadcast_bundle(bundle)
        # every rank must now hold bit-identical weights / conditioning
        ref = packers.pack_dit(synth.synth_dit_state_dict(cfg), cfg, torch.bfloat16)
        arena = bundle.dit_arena()
        same = all(torch.equal(arena.view(k), v) for k, v in ref.items())
        dref = packers.pack_dac(synth.synth_dac_state_dict(dcfg), dcfg)
        same = same and all(torch.equal(bundle.dac_arena().view(k), v) for k, v in dref.items())
        cref = synth.synth_conditioning(cfg, 1.0, t2a=False)
        got = bundle.cond_views()
        same = same and torch.equal(got["clip"], cref["clip"]) and torch.equal(got["sync"], cref["sync"])
        for k in ("text", "uncond_text"):       # carried zero-padded to the fixed text length
            T = cref[k].shape[1]
            same = same and torch.equal(got[k][:, :T], cref[k]) and float(got[k][:, T:].abs().sum()) == 0.0
        # batch sharding: same CPU-generator draw on every rank, disjoint contiguous slices
        gen = torch.Generator("cpu").manual_seed(1234)
        noise = sampler.draw_noise(5, 128, 50, torch.float32, gen)
        lo, hi = D.shard_range(5, rank, world)
        q.put((rank, same, lo, hi, float(noise[lo:hi].double().sum()), float(noise.double().sum()), n_coll[0]))
    finally:
        dist.destroy_process_group()


def test_world2_single_broadcast_and_sharding():
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
    assert all(r[1] for r in res), "broadcast bundle differs from rank 0's"
    assert all(r[6] == 1 for r in res), "the data-parallel setup must be exactly ONE collective"
    (_, _, lo0, hi0, s0, tot0, _), (_, _, lo1, hi1, s1, tot1, _) = res
    assert (lo0, hi0, lo1, hi1) == (0, 3, 3, 5)
    assert tot0 == tot1 and abs((s0 + s1) - tot0) < 1e-9


@pytest.mark.parametrize("config,bs", [("c2", 1), ("c4", 8)])
def test_bench_spawns_its_own_ranks(config, bs):
    """`python bench.py --gpus 2` without a launcher: bench.py spawns 2 ranks itself, they rendezvous on
    127.0.0.1, rank 0 packs and ONE broadcast ships the bundle; exit code 0 and one JSON line.  `--config c4`
    (BASELINE configs[3]: video features in the bundle, 8 clips per GPU) shards 16 clips over the two ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run",
                        "--config", config, "--model", "tiny", "--duration", "1"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len([ln for ln in r.stdout.splitlines() if ln.strip()]) == 1, r.stdout[:500]     # stdout = the ONE JSON line
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["collectives"] == 1 and out["bundle_bytes"] > 0
    assert [x["shard"] for x in out["ranks"]] == [[0, bs], [bs, 2 * bs]] and all(x["ok"] for x in out["ranks"])
    assert abs(sum(x["noise_sum"] for x in out["ranks"]) - out["noise_total"]) < 1e-9


def test_bench_eight_ranks_c4_dry_run():
    """The 8-GPU job of BASELINE configs[3] (bs = 64: eight ranks x eight clips) through bench.py's own spawner, on CPU
    tensors over gloo: eight ranks rendezvous on 127.0.0.1, rank 0 packs, ONE broadcast ships the bundle (video features
    included), rank r owns clips [8r, 8r + 8), every rank verifies what it received against a local re-synthesis, and
    stdout is one JSON line.  (The largest world the GPU side can be tried at before the driver's 8-GPU run is 2.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--dry-run",
                        "--config", "c4", "--model", "tiny", "--duration", "1"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len([ln for ln in r.stdout.splitlines() if ln.strip()]) == 1, r.stdout[:500]
    out = json.loads(r.stdout.strip())
    assert out["n_gpus"] == 8 and out["collectives"] == 1 and out["backend"] == "gloo"
    assert [x["rank"] for x in out["ranks"]] == list(range(8))
    assert [x["shard"] for x in out["ranks"]] == [[8 * i, 8 * i + 8] for i in range(8)] and all(x["ok"] for x in out["ranks"])
    assert abs(sum(x["noise_sum"] for x in out["ranks"]) - out["noise_total"]) < 1e-9


def test_bench_rejects_inconsistent_launch():
    """A mismatch between --gpus and the launcher's WORLD_SIZE is a clean error (exit code 2), not an assert."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_bundle_layout_is_config_only():
    """Every rank derives the same byte layout from (config, dtype, duration) alone - no metadata exchange."""
    from foley_amd.host import config as C, distributed as D, packers, synth
    spec = D.bundle_spec(C.TINY, C.DAC_TINY, torch.bfloat16, 1.0)
    total, table = packers.arena_layout(packers.pack_dit(synth.synth_dit_state_dict(C.TINY), C.TINY, torch.bfloat16))
    assert (spec.dit_bytes, spec.dit_table) == (total, table)
    total, table = packers.arena_layout(packers.pack_dac(synth.synth_dac_state_dict(C.DAC_TINY), C.DAC_TINY))
    assert (spec.dac_bytes, spec.dac_table) == (total, table)
    assert spec.cond_shapes["clip"] == (1, 8, 768) and spec.cond_shapes["sync"] == (1, 16, 768)
    assert spec.dac_off % 256 == 0 and spec.cond_off % 256 == 0 and spec.total > spec.cond_off
    xxl = D.bundle_spec(C.XXL, C.DAC48K, torch.bfloat16, 5.0)      # meta tensors: no 10 GB allocation here
    assert 10.2e9 < xxl.dit_bytes < 10.4e9 and xxl.cond_shapes["sync"] == (1, 112, 768)


def test_shard_range_covers_batch():
    from foley_amd.host import distributed as D
    for total in (1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            spans = [D.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
