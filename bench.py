"""Headline benchmark: audio-seconds generated per wall-second for the Foley sampling path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5] [--bs B] [--precision bf16|fp32] [--progress]

A "step" is ONE pass of the whole hot path over one batch of synthetic input, host to host: H2D of
the CPU-generator noise, step-invariant precompute, the 50-iteration Euler/CFG loop over the xxl
DiT, the DAC-VAE 48 kHz decode of `bs` clips and the D2H of the waveform.  Weights and
conditioning are resident in HBM when the timed region starts.

Configurations (BASELINE.json `configs`):
  c2 (default)  T2A 5 s, 50 Euler steps, CFG 4.5, bf16, hunyuanvideo-foley-xxl        - the headline metric
  c3            V2A: same shapes, non-empty SigLIP2 / Synchformer stand-in features (seed 2)
  c4            V2A features, bs = 8 clips per GPU: with --gpus 8 this is BASELINE configs[3] (bs=64 over 8 GPUs)
  c5            fp8_e4m3fn weight storage, 30 s clip, negative-prompt CFG

The default line (c2, bs=1/GPU at every N, so that the driver's per-N values are one scaling curve) also carries, as
`extra`: bs8 (the bs=8/GPU half of the metric), c4 (V2A features, bs=8/GPU - at N=8 the C4 workload), progress (the
per-iteration host callback path ComfyUI's progress bar takes) and, at N=1, c3 and c5 - three timed passes each (four for the bs=8 lines).

Multi-GPU: `python bench.py --gpus N` SPAWNS its own N ranks (one process per GPU, RCCL) when it
is not already running under a launcher; under `torch.distributed.run` (RANK / WORLD_SIZE set) it
joins that job instead.  Clips are independent, so every rank processes its own `bs` clips (weak
scaling) after ONE broadcast of the bundle [DiT arena | DAC arena | conditioning]
(host/distributed.py); value is the whole-job aggregate.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import PKG_DIR, load_package  # noqa: E402

load_package()
from foley_amd.host import config as C  # noqa: E402
from foley_amd.host import distributed as D  # noqa: E402
from foley_amd.host import packers, sampler, synth  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
STEPS_PER_CLIP, GUIDANCE = 50, 4.5
CONFIGS = {
    "c2": dict(duration=5.0, t2a=True, quantization="none",
               desc="T2A 5 s, 50 Euler steps, CFG 4.5"),
    "c3": dict(duration=5.0, t2a=False, quantization="none",
               desc="V2A 5 s @ 8 fps (SigLIP2 + Synchformer stand-in features), 50 Euler steps, CFG 4.5"),
    "c4": dict(duration=5.0, t2a=False, quantization="none", bs=8,
               desc="V2A 5 s (SigLIP2 + Synchformer stand-in features), 50 Euler steps, CFG 4.5, data-parallel clips, one RCCL "
                    "broadcast of weights + conditioning (BASELINE configs[3] at --gpus 8: bs=64)"),
    "c5": dict(duration=30.0, t2a=True, quantization="fp8_e4m3fn",
               desc="fp8_e4m3fn weight storage, 30 s long-form, negative-prompt CFG 4.5, 50 Euler steps"),
}


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


def kernel_src_sha() -> str:
    """Identity of the kernel sources a counter pass was taken on (profiles/*_pmc_traffic.json)."""
    h = hashlib.sha256()
    d = os.path.join(PKG_DIR, "csrc")
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".h")):
            h.update(n.encode())
            h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


def kernel_key(name: str) -> str:
    """A kernel symbol reduced to what identifies the instance: `void f<a, b>(Args)` -> `f<a,b>` (return type, argument list and
    blanks dropped) - the demanglers of rocprofv3 and of libfoley_hip.so (abi::__cxa_demangle) differ only in those."""
    n = name.strip().replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")   # tools/prof_summary.py's spelling
    if n.startswith("void "):
        n = n[5:]
    depth, end = 0, len(n)
    for i, ch in enumerate(n):          # cut at the '(' that opens the argument list (outside any template bracket)
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            end = i
            break
    return n[:end].replace(" ", "")


def physical_cores() -> int:
    """Physical cores this process may run on (distinct (package, core) pairs of /proc/cpuinfo within the affinity mask)."""
    try:
        allowed = os.sched_getaffinity(0)
    except Exception:
        allowed = set(range(os.cpu_count() or 1))
    try:
        cores, cpu, phys = set(), None, None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cpu, phys = int(v), None
            elif k == "physical id":
                phys = int(v)
            elif k == "core id" and cpu in allowed:
                cores.add((phys, int(v)))
        if cores:
            return len(cores)
    except Exception:
        pass
    return max(1, len(allowed))


def cpu_baseline(sd_gpu, dsd_gpu, cfg, cond, noise, duration, threads=16, full=False):
    """The CPU oracle (a port of the reference's fp32 CPU path) on this box's host cores, per BASELINE.md section 4, bounded to
    ~35 s of CPU work in the default run:
      * BASELINE configs[0] (C1: text-to-audio 1 s, 10 Euler iterations, CFG off + DAC decode) run IN FULL once at `threads`
        (16: where the torch CPU kernels stop scaling on these hosts);
      * the metric's own configuration (C2): three DiT forwards of the CFG pair's conditional half + the DAC decode of the same
        clip, extrapolated to the 2 x 50 forwards of a clip (the extrapolation is stated in `sample`).
    `full` (--cpu-baseline-full, minutes): C1 three times (median) at `threads` AND at threads = physical cores - on the 128-core
    hosts of this pool one C1 pass takes 154 s at 128 threads against 24 s at 16 (profiles/r05_bench_c2.json), which is why the
    default run does not pay for it.  `value` is the C2 figure (the metric's configuration); a reported baseline only."""
    from oracle import foley_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    phys = physical_cores()
    sd = {k: v.float().cpu() for k, v in sd_gpu.items()}
    dsd = {k: v.float().cpu() for k, v in dsd_gpu.items()}
    cc = {k: v.float().cpu() for k, v in cond.items()}
    x = noise[:1].float().cpu()
    # C1: its own conditioning (1 s: Lv 8, Ls 16) and noise, like golden g6
    c1 = synth.synth_conditioning(cfg, 1.0, t2a=True, sd=sd)
    n1 = torch.randn((1, cfg.latent_dim, int(cfg.frame_rate)), generator=torch.Generator("cpu").manual_seed(1234))

    def c1_pass():
        t0 = time.perf_counter()
        with torch.inference_mode():
            lat = O.sample_latents(sd, cfg.heads, n1, c1["text"], c1["uncond_text"], c1["clip"], c1["sync"], 10, 1.0)
            O.dac_decode(dsd, lat)
        return time.perf_counter() - t0

    best = max(1, min(threads, avail))
    c1_runs = []
    for nthreads in dict.fromkeys([best] + ([max(1, min(phys, avail))] if full else [])):
        torch.set_num_threads(nthreads)
        times = sorted(c1_pass() for _ in range(3 if full else 1))
        c1_runs.append({"threads": nthreads, "repeats": len(times), "median_s": round(times[len(times) // 2], 3),
                        "audio_sec_per_sec": round(1.0 / times[len(times) // 2], 4)})
    if full:
        best = min(c1_runs, key=lambda r: r["median_s"])["threads"]
    torch.set_num_threads(best)
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
    return {"value": duration / t_clip, "unit": "audio-sec/sec", "cores": best, "kind": "port",
            "physical_cores": phys, "hardware_threads": avail,
            "c1_full": {"workload": "BASELINE configs[0]: T2A 1 s, 10 Euler iterations, CFG off, fp32, bs 1, incl. DAC decode, run in full",
                        "runs": c1_runs},
            "sample": f"{n_sample} of {n_fwd} DiT forwards ({t_fwd:.2f}s each, fp32 torch-CPU oracle, {best} threads; {phys} physical cores / "
                      f"{avail} hardware threads available) + the DAC decode ({t_dec:.2f}s) of the same {duration:g} s clip, extrapolated "
                      f"x{n_fwd}/{n_sample}; C1 in full: " + "; ".join(f"{r['median_s']}s (median of {r['repeats']}) at {r['threads']} threads" for r in c1_runs)}


def encoder_pass(cfg, duration, dev, dtype, frame_rate=16.0, hw=(480, 640), repeats=2):
    """BASELINE configs[2]: 'SigLIP2 + Synchformer conditioning' from synthetic frames (uint8
    [int(dur*fr), H, W, 3] uniform noise, seed 3) - frame selection, the two v2 pipelines, the SigLIP2 vision tower
    (google/siglip2-base-patch16-512's architecture: ViT-B/16 at 512 px, random init - no checkpoints in the image) and
    the Synchformer visual extractor (synthesised weights), all on the GPU, both encoders on the HIP engine
    (host/encoders_hip.py).  Returns (visual features, per-stage milliseconds of the last of `repeats` passes)."""
    from transformers import SiglipVisionConfig, SiglipVisionModel
    from foley_amd.host import encoders as E
    torch.manual_seed(0)
    sig = SiglipVisionModel(SiglipVisionConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                                               intermediate_size=3072, image_size=512, patch_size=16)).eval().to(dev, dtype)
    sync_sd = {k: v.to(dev, dtype) for k, v in synth.materialize(E.synchformer_schema()).items()}
    g = torch.Generator().manual_seed(3)
    n = int(duration * frame_rate)
    image = torch.randint(0, 256, (n, hw[0], hw[1], 3), generator=g, dtype=torch.uint8).float() / 255.0   # a ComfyUI IMAGE batch
    # CLAP text encoder (laion/larger_clap_general's architecture: RoBERTa-base, random init) on two synthetic prompts of 12 / 5
    # tokens (+ <s> </s>), right-padded like the tokenizer does
    from transformers import ClapTextConfig, ClapTextModelWithProjection
    from foley_amd.host import encoders_hip as EH
    clap = ClapTextModelWithProjection(ClapTextConfig()).eval()
    clap_sd = {k: v.detach().to(dev, dtype) for k, v in clap.state_dict().items() if k.startswith("text_model.") and v.is_floating_point()}
    ids = torch.ones(2, 14, dtype=torch.long)
    ids[0] = torch.cat([torch.tensor([0]), torch.randint(3, 50000, (12,), generator=g), torch.tensor([2])])
    ids[1, :7] = torch.cat([torch.tensor([0]), torch.randint(3, 50000, (5,), generator=g), torch.tensor([2])])
    ids = ids.to(dev)
    tm = {}
    for _ in range(repeats):
        tm = {}
        torch.cuda.synchronize(dev)
        tc = time.perf_counter()
        text = EH.clap_text_hidden_hip(clap_sd, ids, (ids != 1).long(), dtype, eps=clap.config.layer_norm_eps)
        torch.cuda.synchronize(dev)
        tm["clap_text_ms"] = 1e3 * (time.perf_counter() - tc)
        assert text.shape == (2, 14, 768) and bool(torch.isfinite(text).all())
        t0 = time.perf_counter()
        f8, f25 = E.select_frames(image, duration, frame_rate, device=dev)      # incl. the H2D copy of the selected frames
        tm["select_frames_ms"] = 1e3 * (time.perf_counter() - t0)
        feats, alen = E.video_features(f8, f25, sig, sync_sd, dev, model_dtype=dtype, timings=tm)
        torch.cuda.synchronize(dev)
        tm["total_ms"] = 1e3 * (time.perf_counter() - t0) + tm["clap_text_ms"]
    assert abs(alen - duration) < 1e-6 and all(bool(torch.isfinite(v).all()) for v in feats.values())
    tm = {k: round(v, 2) for k, v in tm.items()}
    tm["frames"] = f"{n} frames {hw[0]}x{hw[1]} uint8 noise (seed 3) at {frame_rate:g} fps -> {f8.shape[0]} @ 8 fps (SigLIP2 512 px) + {f25.shape[0]} @ 25 fps (Synchformer 224 px)"
    tm["weights"] = "SigLIP2 ViT-B/16-512 and CLAP RoBERTa-base random init; Synchformer synthesised; all three on libfoley_hip.so"
    return feats, tm


# ----------------------------------------------------------------------------- self-launch
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n: int, argv) -> int:
    """Run this script as `n` ranks of one job (LOCAL_RANK = RANK, rendezvous on 127.0.0.1).  Rank 0
    inherits stdout (its JSON line is this process's output); the first failing rank ends the job."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FOLEY_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


# ----------------------------------------------------------------------------- one rank
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--bs", type=int, default=None, help="clips per GPU per step (default 1, plus an extra bs=8 measurement)")
    ap.add_argument("--duration", type=float, default=None)
    ap.add_argument("--quantization", default=None, choices=["none", "fp8_e4m3fn", "fp8_e5m2"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--model", default="xxl", choices=["xxl", "xl", "tiny"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="BASELINE.md section 4 to the letter: C1 three times (median) at 16 threads AND at threads = physical cores (minutes)")
    ap.add_argument("--no-extra", action="store_true", help="skip the per-kernel profile pass and the bs=8 measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--with-encoders", action="store_true",
                    help="c3 / c4: compute the conditioning from synthetic uint8 frames (seed 3) through the full-size SigLIP2 "
                         "vision tower and Synchformer on the HIP engine (random-init / synthesised weights) and report encoder_ms")
    ap.add_argument("--progress", action="store_true",
                    help="time the path a ComfyUI run takes: a host progress callback after every loop iteration "
                         "(graph replay + D2D copy + stream synchronise per iteration, foley_rt.hip foley_sample)")
    ap.add_argument("--dry-run", action="store_true",
                    help="setup only (pack, the single broadcast, sharding) on CPU tensors - no HIP work; used by the gloo tests")
    return ap.parse_args(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    if a.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not launched and a.gpus > 1:
        if not a.dry_run and torch.cuda.device_count() < a.gpus and not os.environ.get("FOLEY_BENCH_SHARE_DEVICE"):
            print(f"bench.py: --gpus {a.gpus} but only {torch.cuda.device_count()} HIP device(s) are visible",
                  file=sys.stderr)
            return 2
        return spawn_ranks(a.gpus, argv)
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    if a.gpus != world:
        print(f"bench.py: --gpus {a.gpus} does not match the launcher's WORLD_SIZE={world}", file=sys.stderr)
        return 2
    # stdout carries the ONE JSON line and nothing else: while the rank runs, file descriptor 1 points at stderr (RCCL's
    # version banner, gloo's connection notice and anything a library prints land there); emit() writes to the real one
    global _STDOUT_FD
    sys.stdout.flush()
    _STDOUT_FD = os.dup(1)
    os.dup2(2, 1)
    try:
        return run_rank(a, world, rank, local, launched)
    finally:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)      # RCCL printf()s its banner into the C stdio buffer: drain it while 1 -> stderr
        except OSError:
            pass
        os.dup2(_STDOUT_FD, 1)
        os.close(_STDOUT_FD)
        _STDOUT_FD = None


_STDOUT_FD = None


def emit(obj) -> None:
    line = (json.dumps(obj) + "\n").encode()
    if _STDOUT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_STDOUT_FD, line)


def run_rank(a, world: int, rank: int, local: int, launched: bool) -> int:
    import torch.distributed as dist
    conf = dict(CONFIGS[a.config])
    if a.duration is not None:
        conf["duration"] = a.duration
    if a.quantization is not None:
        conf["quantization"] = a.quantization
    duration, quant = conf["duration"], conf["quantization"]
    use_dist = world > 1 or (launched and bool(os.environ.get("FOLEY_BENCH_FORCE_DIST")))
    on_gpu = not a.dry_run
    if on_gpu and os.environ.get("FOLEY_BENCH_SHARE_DEVICE"):      # tests on a 1-GPU box: the ranks share the visible devices (gloo only)
        local %= torch.cuda.device_count()
    dev = torch.device("cuda", local) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(dev)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    cfg = C.dit_config(a.model)
    dac_cfg = C.DAC48K if a.model != "tiny" else C.DAC_TINY
    dtype = packers.torch_dtype(a.precision)
    if quant != "none" and dtype == torch.float32:
        print("bench.py: fp8 weight storage needs bf16 compute (the reference cannot run it in fp32 either)", file=sys.stderr)
        return 2

    # ---- setup (untimed): rank 0 synthesises + packs INTO the bundle, ONE broadcast ships it
    store = packers.FP8_DTYPES.get(quant)          # fp8 weight storage stays fp8 in the arena (and in the broadcast)
    spec = D.bundle_spec(cfg, dac_cfg, dtype, duration, weight_store=store)
    bundle = D.Bundle(spec, dev)
    sd = dsd = None
    if rank == 0:
        from foley_amd import nodes
        sd = synth.synth_dit_state_dict(cfg, device=dev)
        dsd = synth.synth_dac_state_dict(dac_cfg, device=dev)
        cond = synth.synth_conditioning(cfg, duration, t2a=conf["t2a"], sd=sd, device=dev, seed=1)
        sdq = nodes.fp8_round_state_dict(sd, quant, autocast=True, param_dtype=dtype) if quant != "none" else sd
        bundle.fill(packers.pack_dit(sdq, cfg, dtype, weight_store=store), packers.pack_dac(dsd, dac_cfg), cond)
        del sdq
    bcast_s = D.broadcast_bundle(bundle) if use_dist else 0.0
    cond = {k: v.clone() for k, v in bundle.cond_views().items()}
    la = int(duration * cfg.frame_rate)
    bs_main = a.bs or conf.get("bs", 1)
    gen = torch.Generator("cpu").manual_seed(1234)
    noise_all = sampler.draw_noise(world * bs_main, cfg.latent_dim, la, dtype, gen)     # same on every rank
    lo, hi = D.shard_range(world * bs_main, rank, world)

    if a.dry_run:
        # the collective plumbing only (CPU tests): every rank checks what it received against a local re-synthesis
        ref_sd = synth.synth_dit_state_dict(cfg)
        ok = all(torch.equal(bundle.dit_arena().view(k), v) for k, v in packers.pack_dit(ref_sd, cfg, dtype).items())
        ok = ok and all(torch.equal(bundle.dac_arena().view(k), v)
                        for k, v in packers.pack_dac(synth.synth_dac_state_dict(dac_cfg), dac_cfg).items())
        cref = synth.synth_conditioning(cfg, duration, t2a=conf["t2a"], sd=ref_sd, seed=1)
        ok = ok and torch.equal(cond["clip"], cref["clip"]) and torch.equal(cond["sync"], cref["sync"])
        ok = ok and torch.equal(cond["text"][:, :cref["text"].shape[1]], cref["text"])
        rec = {"rank": rank, "ok": bool(ok), "shard": [lo, hi], "noise_sum": float(noise_all[lo:hi].double().sum())}
        recs = [None] * world
        if use_dist:
            dist.all_gather_object(recs, rec)
        else:
            recs = [rec]
        if rank == 0:
            emit({"dry_run": True, "n_gpus": world, "backend": a.backend if use_dist else None,
                              "collectives": 1 if use_dist else 0, "broadcast_s": bcast_s, "bundle_bytes": spec.total,
                              "ranks": recs, "noise_total": float(noise_all.double().sum())})
        if use_dist:
            dist.destroy_process_group()
        return 0 if ok else 1

    model = sampler.FoleyModel.from_arena(cfg, bundle.dit_arena(), dtype, dev, dac_cfg=dac_cfg, quantization=quant)
    dac = sampler.FoleyDAC.from_arena(bundle.dac_arena(), dev, dac_cfg)
    visual = {"siglip2_feat": cond["clip"], "syncformer_feat": cond["sync"]}
    text = {"text_feat": cond["text"], "uncond_text_feat": cond["uncond_text"]}
    graph = not a.no_graph
    encoders = None
    if a.with_encoders:
        if conf["t2a"]:
            print("bench.py: --with-encoders needs a video-to-audio configuration (c3 / c4)", file=sys.stderr)
            return 2
        visual, encoders = encoder_pass(cfg, duration, dev, dtype)     # every rank encodes the same synthetic clip

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    pieces = {}

    def measure(bs: int, noise_cpu: torch.Tensor, steps: int, warmup: int, visual=visual, text=text, model=model,
                duration=duration, progress=a.progress):
        """Host-to-host passes: pinned CPU noise in, CPU waveform out."""
        noise_cpu = noise_cpu.pin_memory()
        la = noise_cpu.shape[2]
        ticks = []
        cb = (lambda i, n: ticks.append(i)) if progress else None

        def one_pass():
            audio, _sr = sampler.denoise_process_with_generator(
                visual, text, duration, model, dac, GUIDANCE, STEPS_PER_CLIP, bs, "euler",
                noise=noise_cpu.to(dev, non_blocking=True), use_graph=graph, progress=cb)
            return audio.cpu()

        for _ in range(warmup):
            one_pass()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            tq = time.perf_counter()
            audio = one_pass()
            if os.environ.get("FOLEY_BENCH_TRACE"):      # per-pass host times (audio.cpu() has synchronised)
                print(f"bench.py: pass {1e3 * (time.perf_counter() - tq):.1f} ms", file=sys.stderr)
        barrier()
        dt = time.perf_counter() - t0
        # pieces of one more pass: precompute (host clock around a synchronised call), loop / decode (HIP events
        # recorded by the library on the launch stream)
        lat = noise_cpu.to(dev).float().contiguous()
        plan = sampler.build_plan(model, visual, text, la, GUIDANCE, STEPS_PER_CLIP, bs, "euler")
        torch.cuda.synchronize()
        tp = time.perf_counter()
        model.ctx.prepare(plan)
        torch.cuda.synchronize()
        pieces["prepare_ms"] = 1e3 * (time.perf_counter() - tp)
        model.ctx.sample(lat, use_graph=graph)
        torch.cuda.synchronize()
        loop_ms = model.ctx.last_elapsed_ms()
        model.ctx.dac_decode(lat)
        torch.cuda.synchronize()
        dac_ms = model.ctx.last_elapsed_ms()
        pieces["local"] = {"clips": bs * steps, "pass_ms": round(1e3 * dt / steps, 2), "loop_ms": round(loop_ms, 2)}
        if use_dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        assert audio.shape == (bs, 1, la * dac_cfg.hop) and bool(torch.isfinite(audio).all())
        assert not progress or len(ticks) == (steps + warmup) * STEPS_PER_CLIP
        return dt, loop_ms, dac_ms

    def device_identity():
        pr = torch.cuda.get_device_properties(dev)
        pci = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
        return {"device": str(dev), "name": pr.name, "uuid": str(getattr(pr, "uuid", "") or "pci-" + pci), "pci": pci}

    def gather_ranks(local_rec):
        """Every rank's own record, all-gathered on the job's communicator: what each rank ran on and how long ITS passes took
        (the line's `value` uses the slowest rank) - so that "did RCCL see N ranks, did each do its clips" reads off the line."""
        rec = dict(rank=rank, **device_identity(), **local_rec)
        if not use_dist:
            return [rec]
        recs = [None] * world
        dist.all_gather_object(recs, rec)
        return recs

    dt, loop_ms, dac_ms = measure(bs_main, noise_all[lo:hi], a.steps, a.warmup)
    prepare_ms = pieces.get("prepare_ms")
    rank_records = gather_ranks(dict(pieces["local"], shard=[lo, hi]))
    f_clip = flops_clip(cfg, duration, STEPS_PER_CLIP, GUIDANCE)
    f_loop = f_clip - 2.30933e9 * la
    peak = PEAK_TFLOPS[a.precision]

    # ---- per-kernel profile (HIP-event brackets around every launch of an eager forward, mid-loop iteration)
    kernels, bracket_us = [], None
    if not a.no_extra:
        lat = noise_all[lo:hi].to(dev).float().contiguous()
        prof, bracket_us = model.ctx.profile_forward(lat, it=STEPS_PER_CLIP // 2, repeats=2)
        for e in prof:
            tf = e["flop_per_launch"] / (e["avg_us"] * 1e-6) / 1e12 if e["avg_us"] > 0 else 0.0
            gbs = e["bytes_per_launch"] / (e["avg_us"] * 1e-6) / 1e9 if e["avg_us"] > 0 else 0.0
            kernels.append({"name": e["label"], "kernel": e["kernel"], "calls_per_iteration": e["calls_per_forward"],
                            "avg_us": round(e["avg_us"], 2), "us_per_iteration": round(e["avg_us"] * e["calls_per_forward"], 1),
                            "gflop_per_launch": round(e["flop_per_launch"] / 1e9, 3),
                            "mbytes_per_launch": round(e["bytes_per_launch"] / 1e6, 3),
                            "tflops": round(tf, 1), "frac": round(tf / peak, 4), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4)})
        kernels.sort(key=lambda k: -k["us_per_iteration"])

    # ---- the bs=8-per-GPU half of the BASELINE metric, same JSON line
    extra = {}
    if not a.no_extra and a.bs is None and a.config == "c2" and not a.progress:
        def extra_line(bs, dur, workload, passes=3, **kw):
            """`passes` timed passes (1 warm-up) of another configuration of the same path; clips sharded like the main line."""
            la_x = int(dur * cfg.frame_rate)
            gx = torch.Generator("cpu").manual_seed(1234)
            nx = sampler.draw_noise(world * bs, cfg.latent_dim, la_x, dtype, gx)
            lox, hix = D.shard_range(world * bs, rank, world)
            pk = kw.pop("peak", peak)
            dtx, loopx, dacx = measure(bs, nx[lox:hix], passes, 1, duration=dur, **kw)
            fl = flops_clip(cfg, dur, STEPS_PER_CLIP, GUIDANCE) - 2.30933e9 * la_x
            line = {"value": world * bs * passes * dur / dtx, "unit": "audio-sec/sec", "workload": workload, "clips_per_gpu": bs,
                    "steps": passes, "warmup": 1, "ms_per_step": 1e3 * dtx / passes, "loop_ms": loopx, "dac_decode_ms": dacx,
                    "loop_frac": bs * fl / (loopx * 1e-3) / 1e12 / pk}
            if use_dist:                       # per-rank evidence: every rank's own pass time and clip count
                rr = gather_ranks(dict(pieces["local"], shard=[lox, hix]))
                line["ranks"] = {"n": len(rr), "clips_total": sum(r["clips"] for r in rr),
                                 "pass_ms_min": min(r["pass_ms"] for r in rr), "pass_ms_max": max(r["pass_ms"] for r in rr)}
            return line

        extra["bs8"] = extra_line(8, duration, "c2 at bs=8 per GPU (the bs=8 half of the BASELINE metric)", passes=4)
        # stand-in V2A features: a pure function of the seed, synthesised on every rank like the noise
        c3c = synth.synth_conditioning(cfg, duration, t2a=False, device=dev, seed=1)
        vis3 = {"siglip2_feat": c3c["clip"], "syncformer_feat": c3c["sync"]}
        extra["c4"] = extra_line(8, duration, f"c4: {CONFIGS['c4']['desc']}; {world} GPU(s) x 8 clips", passes=4, visual=vis3)
        extra["progress"] = extra_line(1, duration, "c2 bs=1 with a host progress callback after every loop iteration "
                                       "(the path a ComfyUI run takes)", progress=True)
        if world == 1:
            extra["c3"] = extra_line(1, duration, f"c3: {CONFIGS['c3']['desc']}", visual=vis3)
            # BASELINE configs[2] end to end: conditioning from synthetic frames through the encoders on the HIP engine
            _v, enc = encoder_pass(cfg, duration, dev, dtype)
            extra["c3"]["encoder_ms"] = enc
            extra["c3"]["end_to_end_ms"] = round(enc["total_ms"] + extra["c3"]["ms_per_step"], 1)
            from foley_amd import nodes
            m5 = nodes.HunyuanModelLoader.pack_state_dict(sd, "bf16", "fp8_e4m3fn", device=dev, cfg=cfg, dac_cfg=dac_cfg)
            c5c = synth.synth_conditioning(cfg, CONFIGS["c5"]["duration"], t2a=True, sd=sd, device=dev, seed=1)
            extra["c5"] = extra_line(1, CONFIGS["c5"]["duration"], f"c5: {CONFIGS['c5']['desc']}", model=m5,
                                     visual={"siglip2_feat": c5c["clip"], "syncformer_feat": c5c["sync"]})
            del m5
            # the PARITY mode (precision=fp32: fp32 operands on v_mfma_f32_32x32x2_f32, no split-K) - the mode that carries
            # north_star's 1e-3 waveform gate (golden g17) gets a throughput line too, against the fp32 matrix peak
            if a.precision != "fp32":
                m32 = nodes.HunyuanModelLoader.pack_state_dict(sd, "fp32", "none", device=dev, cfg=cfg, dac_cfg=dac_cfg)
                extra["fp32"] = extra_line(1, duration, "c2 bs=1 in the fp32 parity mode (fp32 operands / fp32 accumulate; the mode gated at 1e-3 "
                                           "on the waveform against the reference's fp32 sampler, golden g17); loop_frac against the 157.3 TFLOP/s "
                                           "fp32 matrix peak", passes=2, model=m32, peak=PEAK_TFLOPS["fp32"])
                del m32

    if rank == 0:
        clips = world * bs_main * a.steps
        loop_tf = bs_main * f_loop / (loop_ms * 1e-3) / 1e12
        traffic, traffic_src = None, None
        import glob
        for tf_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):   # newest round first
            try:
                tj = json.load(open(tf_path))
            except Exception:
                continue
            # only a counter pass taken on exactly these kernel sources and this workload is reported
            if tj.get("kernel_src_sha") == kernel_src_sha() and tj.get("workload") == f"{a.config}/bs{bs_main}/{a.precision}/{a.model}":
                traffic = tj["hbm_bytes_per_loop_iteration"]
                traffic_src = {"file": os.path.relpath(tf_path, ROOT), "kernel_src_sha": tj["kernel_src_sha"],
                               "unit": "bytes per loop iteration (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes on these kernel sources)"}
                break
        mfma_busy = None
        for mf_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")), reverse=True):
            try:
                mj = json.load(open(mf_path))
            except Exception:
                continue
            if mj.get("kernel_src_sha") == kernel_src_sha() and mj.get("workload") == f"{a.config}/bs{bs_main}/{a.precision}/{a.model}":
                mfma_busy = {"value": mj["mfma_busy"], "kernel": mj["kernel"], "file": os.path.relpath(mf_path, ROOT),
                             "definition": mj["definition"]}
                break
        dom = next((k for k in kernels if k["gflop_per_launch"] > 0), None)   # largest time per iteration among the MFMA kernels
        # the same kernel's average in the rocprofv3 --kernel-trace --stats run of this command (profiles/rNN_rocprof_dominant.json,
        # written by tools/collect_profiles.sh; reported only for exactly these kernel sources): the dispatch-event average above
        # and the traced one differ by ~3 %
        rocprof = None
        for rp_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprof_dominant.json")), reverse=True):
            try:
                rj = json.load(open(rp_path))
            except Exception:
                continue
            if rj.get("kernel_src_sha") == kernel_src_sha() and rj.get("workload") == f"{a.config}/bs{bs_main}/{a.precision}/{a.model}" and dom:
                # the traced row is matched BY NAME: the symbol foley_profile_forward reports for the dominant op (the kernel its
                # launch ran, demangled) against the rows of the trace, both normalised; no row or more than one -> null.  When other
                # ops run the same kernel instance the traced average covers them too: `ops_sharing_kernel` lists them.
                cand = [k for k in rj.get("kernels", []) if kernel_key(k["name"]) == kernel_key(dom.get("kernel", "")) and dom.get("kernel")]
                if len(cand) == 1:
                    rk = cand[0]
                    sharing = [k["name"] for k in kernels if k is not dom and kernel_key(k.get("kernel", "")) == kernel_key(dom["kernel"])]
                    tfr = dom["gflop_per_launch"] * 1e9 / (rk["avg_us"] * 1e-6) / 1e12
                    rocprof = {"kernel": rk["name"], "avg_us": rk["avg_us"], "calls": rk["calls"], "achieved": round(tfr, 1),
                               "frac": round(tfr / peak, 4), "file": os.path.relpath(rp_path, ROOT), "matched_by": "kernel symbol",
                               "ops_sharing_kernel": sharing}
                break
        roof = {"bound": "mfma", "peak": peak, "unit": "TFLOP/s",
                "achieved": dom["tflops"] if dom else loop_tf, "frac": dom["frac"] if dom else loop_tf / peak,
                "kernel": dom["name"] if dom else "whole sampler loop",
                "launch": ("dominant kernel of the loop (largest time per iteration): algorithmic FLOPs of one launch / the dispatch's own "
                           "start->stop HIP events on the launch stream (hipExtLaunchKernelGGL), eager forward at iteration 25, after the timed region"),
                "frac_rocprof": rocprof["frac"] if rocprof else None, "rocprof": rocprof,
                "traffic": traffic, "traffic_source": traffic_src, "mfma_busy": mfma_busy,
                "loop_frac": loop_tf / peak, "loop_achieved": loop_tf, "loop_ms": loop_ms, "dac_decode_ms": dac_ms, "prepare_ms": prepare_ms,
                "algorithmic_tflop_per_clip": f_clip / 1e12, "event_bracket_us": bracket_us, "kernels": kernels}
        out = {
            "metric": f"audio-sec/sec ({duration:g}s clip, 50-step Euler, CFG 4.5)",
            "value": clips * duration / dt, "unit": "audio-sec/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"{a.config}: {conf['desc']}, hunyuanvideo-foley-{a.model}, bs={bs_main}/GPU, "
                                   f"{a.precision} GEMM operands / fp32 accumulate, weight storage {quant}, "
                                   f"DAC-VAE fp32 decode to 48 kHz, host-to-host (H2D noise + D2H waveform timed)",
                       "clips_per_gpu": bs_main, "parallelism": f"dp{world}", "hip_graph": graph,
                       "progress_callback": bool(a.progress),
                       "collectives": 1 if use_dist else 0, "broadcast_s": bcast_s, "bundle_bytes": spec.total,
                       "rccl_nranks": dist.get_world_size() if use_dist else 1, "backend": a.backend if use_dist else None,
                       "ranks": rank_records},
            "roofline": roof,
        }
        if encoders is not None:
            out["encoder_ms"] = encoders
            out["end_to_end_ms_per_step"] = encoders["total_ms"] + 1e3 * dt / a.steps
        if extra:
            out["extra"] = extra
        if world == 1 and not a.no_cpu_baseline and a.model == "xxl" and a.config != "c5":
            out["cpu_baseline"] = cpu_baseline(sd, dsd, cfg, cond, noise_all, duration, full=a.cpu_baseline_full)
        emit(out)
    if use_dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
