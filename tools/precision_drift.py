"""How far does the bf16 throughput mode drift from the fp32 parity mode over a full C2 run
(xxl, 5 s, 50 Euler steps, CFG 4.5, same noise)?  Prints rel-L2 of final latents and waveform.
The reference's own bf16 path differs from its fp32 path by a comparable amount (chaotic
amplification over 100 forwards), so this is a reported quantity, not a parity gate."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from foley_amd.host import config as C, sampler, synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = C.dit_config("xxl")
sd = synth.synth_dit_state_dict(cfg, device=dev)
cond = synth.synth_conditioning(cfg, 5.0, t2a=True, sd=sd, device=dev)
dac = sampler.FoleyDAC(synth.synth_dac_state_dict(C.DAC48K, device=dev), dev)
noise = sampler.draw_noise(1, 128, 250, torch.float32, torch.Generator("cpu").manual_seed(7))
res = {}
for steps in (10, 50):
    for prec in ("fp32", "bf16"):
        model = sampler.FoleyModel(cfg, sd, torch.float32 if prec == "fp32" else torch.bfloat16, dev)
        audio, _sr, lat = sampler.denoise_process_with_generator(
            {"siglip2_feat": cond["clip"], "syncformer_feat": cond["sync"]},
            {"text_feat": cond["text"], "uncond_text_feat": cond["uncond_text"]}, 5.0, model, dac, 4.5, steps, 1, "euler",
            noise=noise, return_latents=True)
        res[(steps, prec)] = (lat.float().cpu(), audio.float().cpu())
        del model
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print(f"{steps} steps: bf16 vs fp32  latents rel-L2 {rel(res[(steps, 'bf16')][0], res[(steps, 'fp32')][0]):.3e}   "
          f"waveform rel-L2 {rel(res[(steps, 'bf16')][1], res[(steps, 'fp32')][1]):.3e}")
