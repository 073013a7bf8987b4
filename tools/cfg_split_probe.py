"""Probe (round 5): would running the two CFG halves of a bs=1 clip as two CONCURRENT half-size forwards (M = 250 each, two
streams) beat the one M = 500 forward?  Two contexts share ONE weight arena on one device; each runs the 50-iteration loop with
guidance off (ncfg = 1: one half) on its own thread / stream, against the regular ncfg = 2 run of one context.  Same FLOPs, same
weight bytes if the follower finds the leader's weights in the L2 / Infinity Cache."""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from foley_amd import nodes  # noqa: E402
from foley_amd.host import config as C, sampler, synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = C.XXL
sd = synth.synth_dit_state_dict(cfg, device=dev)
cond = synth.synth_conditioning(cfg, 5.0, t2a=True, sd=sd, device=dev)
model = nodes.HunyuanModelLoader.pack_state_dict(sd, "bf16", "none", device=dev, cfg=cfg)
del sd
vis = {"siglip2_feat": cond["clip"], "syncformer_feat": cond["sync"]}
txt = {"text_feat": cond["text"], "uncond_text_feat": cond["uncond_text"]}
noise = sampler.draw_noise(1, 128, 250, torch.bfloat16, torch.Generator("cpu").manual_seed(1234)).to(dev).float()


def loop(m, guidance, lat):
    m.ctx.prepare(sampler.build_plan(m, vis, txt, 250, guidance, 50, 1, "euler"))
    m.ctx.sample(lat, use_graph=True)


# one context, both halves in one forward (today)
for rep in range(3):
    lat = noise.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(model, 4.5, lat)
    torch.cuda.synchronize()
    t_full = time.perf_counter() - t0
print("ncfg = 2, one context: %.1f ms (event %.1f ms)" % (1e3 * t_full, model.ctx.last_elapsed_ms()))
for rep in range(3):
    lat = noise.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(model, 1.0, lat)
    torch.cuda.synchronize()
    t_half = time.perf_counter() - t0
print("ncfg = 1, one context (half the rows): %.1f ms" % (1e3 * t_half))
m2 = sampler.FoleyModel.from_arena(cfg, model.arena, model.dtype, dev, quantization=model.quantization)   # second context, SAME weights


def work(m, out):
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        lat = noise.clone()
        loop(m, 1.0, lat)
        st.synchronize()


for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(m, None)) for m in (model, m2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    t_two = time.perf_counter() - t0
print("2 x (ncfg = 1) concurrently, shared weights: %.1f ms  -> %.2fx of the ncfg = 2 run" % (1e3 * t_two, t_full / t_two))
