"""Headline benchmark: audio-seconds generated per wall-second for the Foley sampling path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bs B] [--precision bf16|fp32]

A "step" is ONE pass of the whole hot path over one batch of synthetic input: step-invariant
precompute + 50-iteration Euler/CFG loop over the xxl DiT + DAC-VAE 48 kHz decode of `bs` clips
of 5 s (BASELINE.json configs[1]: T2A 5 s, 50 steps, CFG 4.5, bf16, hunyuanvideo-foley-xxl).
Inputs (noise, conditioning, weights) are resident in HBM when the timed region starts.  With
N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL) every rank processes its
own `bs` clips (weak scaling) after ONE broadcast of the packed weight arena; value is the
whole-job aggregate.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from foley_amd.host import config as C  # noqa: E402
from foley_amd.host import distributed as D  # noqa: E402
from foley_amd.host import packers, sampler, synth  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
DURATION_S, STEPS_PER_CLIP, GUIDANCE = 5.0, 50, 4.5


def flops_forward(cfg: C.DiTConfig, la: int, lv: int, ls: int, lt: int = 77) -> float:
    """Algorithmic FLOPs of one DiT forward for one sample (exact)."""
    Dm, Nt, Ns, Hc = cfg.hidden, cfg.depth_triple, cfg.depth_single, cfg.conv_hidden
    lin = 2 * Dm * Dm * (Nt * (18 + 14 * (la + lv) + 2 * lt) + Ns * 36 * la)
    g = ((256 * Dm + Dm * Dm) + lt * (768 * Dm + Dm * Dm) + 128 * Dm * la + lv * (1536 * Dm + Dm * Dm)
         + ls * (768 * Dm + 3 * Dm * Hc) + la * 2 * Dm * Dm + 128 * Dm * la)
    att = 4 * Dm * (Nt * ((la + lv) ** 2 + (la + lv) * lt) + Ns * la * la)
    return float(lin + 2 * g + att)


def flops_clip(cfg: C.DiTConfig, duration: float, steps: int, guidance: float) -> float:
    la, lv, ls = C.lengths(duration, cfg)
    return steps * (2 if guidance > 1.0 else 1) * flops_forward(cfg, la, lv, ls) + 2.30933e9 * la


def cpu_baseline(sd_gpu, dsd_gpu, cfg, cond, noise, threads=16):
    """The CPU oracle (a port of the reference's fp32 CPU path) on this box's host cores, on a
    bounded sample of the same workload: three DiT forwards of the conditional half (the loop does
    2 x 50 of them per clip) + the DAC decode, extrapolated.  Thread count is capped: the torch
    CPU kernels stop scaling (and thrash) far below this box's 256 hardware threads."""
    from oracle import foley_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = max(1, min(threads, avail))
    torch.set_num_threads(cores)
    sd = {k: v.float().cpu() for k, v in sd_gpu.items()}
    dsd = {k: v.float().cpu() for k, v in dsd_gpu.items()}
    cc = {k: v.float().cpu() for k, v in cond.items()}
    x = noise[:1].float().cpu()
    n_fwd = STEPS_PER_CLIP * 2
    n_sample = 3
    with torch.inference_mode():
        t0 = time.perf_counter()
        for i in range(n_sample):
            v = O.dit_forward(sd, cfg.heads, x, torch.tensor([1000.0 - 20.0 * i]), O.pad_or_trim_text(cc["text"]),
                              cc["clip"], cc["sync"])
        t_fwd = (time.perf_counter() - t0) / n_sample
        t0 = time.perf_counter()
        O.dac_decode(dsd, x - v)
        t_dec = time.perf_counter() - t0
    t_clip = n_fwd * t_fwd + t_dec
    return {"value": DURATION_S / t_clip, "unit": "audio-sec/sec", "cores": cores, "kind": "port",
            "sample": f"{n_sample} of {n_fwd} DiT forwards ({t_fwd:.2f}s each, fp32 torch-CPU oracle, {cores} threads of "
                      f"{avail} available) + the DAC decode ({t_dec:.2f}s) of the same 5 s clip, extrapolated"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bs", type=int, default=1, help="clips per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--model", default="xxl", choices=["xxl", "xl", "tiny"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("FOLEY_BENCH_FORCE_DIST"))
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    cfg = C.dit_config(a.model)
    dtype = packers.torch_dtype(a.precision)

    # ---- setup (untimed): rank 0 synthesises + packs, ONE broadcast ships the arena
    sd = dsd = None
    if rank == 0:
        sd = synth.synth_dit_state_dict(cfg, device=dev)
        dsd = synth.synth_dac_state_dict(C.DAC48K, device=dev)
        cond = synth.synth_conditioning(cfg, DURATION_S, t2a=True, sd=sd, device=dev)
        dit_arena = packers.Arena.from_packed(packers.pack_dit(sd, cfg, dtype), dev)
        dac_arena = packers.Arena.from_packed(packers.pack_dac(dsd, C.DAC48K), dev)
    else:
        cond = dit_arena = dac_arena = None
    if use_dist:
        dit_arena = D.broadcast_arena(dit_arena, dev)
        dac_arena = D.broadcast_arena(dac_arena, dev)
        cond = D.broadcast_tensors(cond, dev)
    model = sampler.FoleyModel.from_arena(cfg, dit_arena, dtype, dev)
    dac = sampler.FoleyDAC.from_arena(dac_arena, dev)
    visual = {"siglip2_feat": cond["clip"], "syncformer_feat": cond["sync"]}
    text = {"text_feat": cond["text"], "uncond_text_feat": cond["uncond_text"]}
    la = int(DURATION_S * cfg.frame_rate)
    gen = torch.Generator("cpu").manual_seed(1234)
    noise_all = sampler.draw_noise(world * a.bs, cfg.latent_dim, la, dtype, gen)     # same on every rank
    lo, hi = D.shard_range(world * a.bs, rank, world)
    noise = noise_all[lo:hi].to(dev)

    def one_pass():
        audio, _sr = sampler.denoise_process_with_generator(
            visual, text, DURATION_S, model, dac, GUIDANCE, STEPS_PER_CLIP, a.bs, "euler", noise=noise,
            use_graph=not a.no_graph)
        return audio

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    ev_ms = 0.0
    for _ in range(a.steps):
        audio = one_pass()
        # HIP-event time of the device-resident loop / decoder, recorded on the launch stream by the library
    barrier()
    dt = time.perf_counter() - t0
    # event-timed sampler loop of the last pass (prepare + decode excluded) for the roofline object
    model.ctx.sample(noise.float().clone(), use_graph=not a.no_graph)
    torch.cuda.synchronize()
    loop_ms = model.ctx.last_elapsed_ms()
    model.ctx.dac_decode(torch.zeros(a.bs, cfg.latent_dim, la, device=dev))
    torch.cuda.synchronize()
    dac_ms = model.ctx.last_elapsed_ms()
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert audio.shape == (a.bs, 1, la * 960) and bool(torch.isfinite(audio).all())

    if rank == 0:
        clips = world * a.bs * a.steps
        f_clip = flops_clip(cfg, DURATION_S, STEPS_PER_CLIP, GUIDANCE)
        f_loop = f_clip - 2.30933e9 * la
        peak = PEAK_TFLOPS[a.precision]
        ach = a.bs * f_loop / (loop_ms * 1e-3) / 1e12
        traffic = None   # PMC counters cannot be read live; the committed rocprofv3 --pmc passes give it
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if a.bs == 1 and a.precision == "bf16" and a.model == "xxl":
                traffic = tj["hbm_bytes_per_loop_iteration"] * STEPS_PER_CLIP
        except Exception:
            pass
        out = {
            "metric": "audio-sec/sec (5s clip, 50-step Euler, CFG 4.5)",
            "value": clips * DURATION_S / dt, "unit": "audio-sec/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"T2A 5 s, 50 Euler steps, CFG 4.5, hunyuanvideo-foley-{a.model}, "
                                   f"bs={a.bs}/GPU, {a.precision} GEMM operands / fp32 accumulate, "
                                   f"DAC-VAE fp32 decode to 48 kHz",
                       "clips_per_gpu": a.bs, "parallelism": f"dp{world}", "hip_graph": not a.no_graph},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                         "traffic": traffic, "traffic_unit": "bytes per foley_sample launch (rocprofv3 PMC, offline pass)",
                         "kernel": "gemm_ws_kernel + gemm_conv3_kernel (MFMA GEMM/conv engine; ~80% of the device-resident sampler loop)",
                         "launch": "one foley_sample call = 50 captured iterations, HIP-event timed on its stream",
                         "loop_ms": loop_ms, "dac_decode_ms": dac_ms,
                         "algorithmic_tflop_per_clip": f_clip / 1e12},
        }
        if world == 1 and not a.no_cpu_baseline and a.model == "xxl":
            out["cpu_baseline"] = cpu_baseline(sd, dsd, cfg, cond, noise_all)
        print(json.dumps(out), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
