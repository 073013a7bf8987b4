"""Generate golden vectors by RUNNING THE REFERENCE (build container only).

    python tests/golden/make_golden.py [--only g1,g5,...] [--check-oracle]

Imports /root/reference through `ref_harness` (stub third-party modules, the
reference's own arithmetic untouched), feeds it the deterministic synthesised
checkpoints of `host/synth.py`, and freezes inputs + reference outputs as small
`.npz` fixtures next to this file.  With --check-oracle (default on) every
fixture is also compared with `oracle/foley_oracle.py` and the script fails if
the restatement drifts from the reference by more than 1e-5 (relative L2).

Fixtures:
  g1_scheduler   sigmas / timesteps / solver steps for all four solvers
  g2_rope        cos/sin tables, rotary application, interleaved positions
  g3_layers      ConvMLP k=3, snake, residual units, decoder blocks, weight-norm fold
  g4_blocks      one full-width (D=1536) triple + single block forward
  g5_dit_tiny    tiny-config full DiT forward
  g6_c1_xxl      config C1 at real xxl dims: 1 s, 10 Euler steps, CFG off, bs 1 (the 1e-3 gate)
  g7_sampler     tiny sampler: CFG 4.5 bs 2 Euler + the three multi-stage solvers
  g9_dac         full-width DAC decoder on [1,128,10]
  g10 - g16      DAC encoder, V2A conditioning, bf16 / fp16 reference runs, full-size single forwards (see each function)
  g17_c2_loop / g18_c2_bf16_loop   the HEADLINE run (xxl, 5 s, 50 Euler iterations, CFG 4.5) through the reference's own
                 sampler loop + DAC decoder: fp32 (the 1e-3 waveform gate) and as the reference runs a bf16 model (~15 min)
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn.functional as F
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_harness  # noqa: E402

PKG = os.path.join(ROOT, "comfyui-hunyuanvideo-foley_amd")
pkg = types.ModuleType("foley_amd")
pkg.__path__ = [PKG]
sys.modules.setdefault("foley_amd", pkg)
from foley_amd.host import config as C  # noqa: E402
from foley_amd.host import synth  # noqa: E402
from oracle import foley_oracle as O  # noqa: E402

NS = None
CHECK = True
TOL = 1e-5


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def check(name, oracle_out, ref_out, tol=None):
    if not CHECK:
        return
    e = rel(oracle_out, ref_out)
    flag = "ok" if e <= (tol or TOL) else "FAIL"
    print(f"    oracle-vs-reference {name}: rel {e:.3e} [{flag}]")
    if flag == "FAIL":
        raise SystemExit(f"oracle drifted from reference on {name}")


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def ref_cfg(c: C.DiTConfig):
    raw = yaml.safe_load(open(os.path.join(ref_harness.REFERENCE_ROOT, "configs", "hunyuanvideo-foley-xxl.yaml")))
    raw["model_config"]["model_kwargs"].update(
        depth_triple_blocks=c.depth_triple, depth_single_blocks=c.depth_single,
        hidden_size=c.hidden, num_heads=c.heads)
    return NS.cfgu.AttributeDict(raw)


def build_ref_dit(c: C.DiTConfig, sd):
    with torch.device("meta"):
        m = NS.hifi.HunyuanVideoFoley(ref_cfg(c), dtype=torch.float32)
    m.load_state_dict(sd, strict=True, assign=True)
    return m.eval()


def build_ref_dac(dc: C.DACConfig, sd):
    kw = dict(NS.utils._DAC_KWARGS)
    kw["decoder_dim"] = dc.decoder_dim
    m = NS.dac.DAC(**kw).eval()
    r = m.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys
    assert all(k.startswith(("encoder", "quant_conv")) for k in r.missing_keys), r.missing_keys
    return m


class ModelDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def run_ref_sampler(md, c, cond, dur, g, steps, bs, solver, gen):
    """The reference's denoise_process_with_generator (utils.py:125-258) + its final latents, captured by
    spying on FlowMatchDiscreteScheduler.step (the function only returns the decoded audio)."""
    trace = []
    orig_step = NS.sched.FlowMatchDiscreteScheduler.step

    def spy(self, *a, **k):
        r = orig_step(self, *a, **k)
        trace.append(r[0])
        return r
    NS.sched.FlowMatchDiscreteScheduler.step = spy
    try:
        audio, _sr = NS.utils.denoise_process_with_generator(
            {"siglip2_feat": cond["clip"], "syncformer_feat": cond["sync"]},
            {"text_feat": cond["text"], "uncond_text_feat": cond["uncond_text"]},
            dur, md, ref_cfg(c), g, steps, bs, solver, generator=gen)
    finally:
        NS.sched.FlowMatchDiscreteScheduler.step = orig_step
    return audio, trace[-1].float().clone()


# ----------------------------------------------------------------------------- G1
def g1():
    out = {}
    for n in (10, 50):
        s = NS.sched.FlowMatchDiscreteScheduler(shift=1.0, solver="euler")
        s.set_timesteps(n, device="cpu")
        out[f"sigmas_{n}"], out[f"timesteps_{n}"] = s.sigmas, s.timesteps
        check(f"sigmas{n}", O.flow_sigmas(n), s.sigmas, 0)
        check(f"timesteps{n}", O.flow_timesteps(O.flow_sigmas(n)), s.timesteps, 0)
    s = NS.sched.FlowMatchDiscreteScheduler(shift=3.0, solver="euler")
    s.set_timesteps(10, device="cpu")
    out["sigmas_10_shift3"] = s.sigmas
    check("shift3", O.flow_sigmas(10, 3.0), s.sigmas, 1e-7)
    g = torch.Generator().manual_seed(7)
    x0 = torch.randn(2, 128, 16, generator=g)
    vs = torch.randn(8, 2, 128, 16, generator=g)
    out["x0"], out["v"] = x0, vs
    for solver in ("euler", "heun-2", "midpoint-2", "kutta-4"):
        s = NS.sched.FlowMatchDiscreteScheduler(shift=1.0, solver=solver)
        s.set_timesteps(8, device="cpu")
        st = O.SolverState(O.flow_sigmas(8), solver)
        x, xo, tr = x0, x0, []
        for i, t in enumerate(s.timesteps):
            x = s.step(vs[i], t, x)[0]
            xo = st.step(vs[i], xo)
            tr.append(x)
        out["trace_" + solver.replace("-", "_")] = torch.stack(tr)
        check("solver " + solver, xo, x, 1e-7)
    save("g1_scheduler", **out)


# ----------------------------------------------------------------------------- G2
def g2():
    out = {}
    for n in (77, 250, 500):
        cos, sin = NS.posemb.get_nd_rotary_pos_embed([128], [n], theta=10000, use_real=True,
                                                     theta_rescale_factor=1.0)
        oc, os_ = O.rope_table(O.rope_positions(n), 128)
        check(f"rope{n}", torch.stack([oc, os_]), torch.stack([cos, sin]), 1e-7)
        if n != 500:
            out[f"cos_{n}"], out[f"sin_{n}"] = cos, sin
        else:   # store a strided view; the full table is 256 KB
            out["cos_500_s7"], out["sin_500_s7"] = cos[::7], sin[::7]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 6, 2, 128, generator=g)
    cos, sin = O.rope_table(O.rope_positions(6), 128)
    rq, _ = NS.attn.apply_rotary_emb(x, x, (cos, sin), head_first=False)
    rqh, _ = NS.attn.apply_rotary_emb(x.transpose(1, 2), x.transpose(1, 2), (cos, sin), head_first=True)
    check("apply_rope", O.apply_rope(x, cos, sin), rq, 1e-7)
    check("apply_rope head_first", O.apply_rope(x, cos, sin).transpose(1, 2), rqh, 1e-7)
    out["rope_x"], out["rope_y"] = x, rq
    # interleaved audio/visual positions: run the reference interleave machinery on
    # "position-coded" tensors and read the positions back.
    for la, lv in ((250, 40), (50, 8), (1500, 240), (55, 8), (251, 40)):
        a = torch.zeros(1, la, 1, 2)
        v = torch.zeros(1, lv, 1, 2)
        inter = NS.hifi.interleave_two_sequences(a, v)
        pos = torch.arange(2 * la, dtype=torch.float32).view(1, 2 * la, 1, 1).expand(1, 2 * la, 1, 2)
        pa, pv = NS.hifi.decouple_interleaved_two_sequences(inter + pos, la, lv)
        opa, opv = O.interleaved_positions(la, lv)
        assert torch.equal(pa[0, :, 0, 0].long(), opa) and torch.equal(pv[0, :, 0, 0].long(), opv), (la, lv)
        # and the token identity: token j must come back as itself
        tag = torch.arange(lv, dtype=torch.float32).view(1, lv, 1, 1).expand(1, lv, 1, 2)
        _, tv = NS.hifi.decouple_interleaved_two_sequences(NS.hifi.interleave_two_sequences(a, tag), la, lv)
        assert torch.equal(tv[0, :, 0, 0], torch.arange(lv, dtype=torch.float32))
        out[f"pos_v_{la}_{lv}"] = opv
    print("    interleaved positions verified for 5 (La, Lv) pairs")
    save("g2_rope", **out)


# ----------------------------------------------------------------------------- G3
def g3():
    import importlib
    mlp = importlib.import_module("hunyuanvideo_foley.models.nn.mlp_layers")
    layers = importlib.import_module("hunyuanvideo_foley.models.dac_vae.nn.layers")
    out = {}
    g = torch.Generator().manual_seed(11)
    # ConvMLP k=3 at small width incl. edge padding
    D, Hc = 256, 768
    cm = mlp.ConvMLP(D, D * 4, kernel_size=3, padding=1).eval()
    sd = {"w1.weight": synth.synth_tensor("g3.w1", (Hc, D, 3), 0.03),
          "w2.weight": synth.synth_tensor("g3.w2", (D, Hc, 3), 0.02),
          "w3.weight": synth.synth_tensor("g3.w3", (Hc, D, 3), 0.03)}
    cm.load_state_dict(sd)
    x = torch.randn(2, 9, D, generator=g)
    with torch.inference_mode():
        y = cm(x)
    oy = O.conv1d_cl(torch.nn.functional.silu(O.conv1d_cl(x, sd["w1.weight"], None, 1))
                     * O.conv1d_cl(x, sd["w3.weight"], None, 1), sd["w2.weight"], None, 1)
    check("convmlp", oy, y)
    out["convmlp_x"], out["convmlp_y"] = x, y
    # snake
    al = synth.synth_tensor("g3.alpha", (1, 16, 1), 0.15, 1.0)
    xs = torch.randn(2, 16, 33, generator=g) * 2
    ys = layers.snake(xs, al)
    check("snake", O.snake(xs, al), ys, 1e-7)
    out["snake_x"], out["snake_alpha"], out["snake_y"] = xs, al, ys
    # weight-norm fold, Conv1d vs ConvTranspose1d
    c1 = layers.WNConv1d(8, 12, kernel_size=7, padding=3)
    ct = layers.WNConvTranspose1d(8, 12, kernel_size=10, stride=5, padding=3, output_padding=1)
    for nm, m in (("conv", c1), ("convT", ct)):
        gk, vk = m.parametrizations.weight.original0, m.parametrizations.weight.original1
        check("wn fold " + nm, O.weight_norm_fold(gk.detach(), vk.detach()), m.weight.detach(), 1e-6)
    # residual units + decoder blocks (stride 2 and odd stride 5) at C=64 via a narrow DAC
    dc = C.DACConfig(decoder_dim=128, rates=(5, 2))
    dsd = synth.synth_dac_state_dict(dc)
    kw = dict(NS.utils._DAC_KWARGS)
    kw.update(decoder_dim=128, decoder_rates=[5, 2])
    dac = NS.dac.DAC(**kw).eval()
    r = dac.load_state_dict(dsd, strict=False)
    assert not r.unexpected_keys
    z = torch.randn(2, 128, 7, generator=g)
    with torch.inference_mode():
        yr = dac.decode(z)
    taps = {}
    yo = O.dac_decode(dsd, z, rates=(5, 2), taps=taps)
    check("dac(5,2)", yo, yr)
    out["dac52_z"], out["dac52_y"] = z, yr
    out["dac52_stage0"] = taps["stage0"]
    save("g3_layers", **out)


# ----------------------------------------------------------------------------- G4
def g4():
    c = C.DiTConfig(name="xxl-1-1", depth_triple=1, depth_single=1)
    sd = synth.synth_dit_state_dict(c)
    m = build_ref_dit(c, sd)
    La, Lv, Ls = C.lengths(1.0, c)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 128, La, generator=g)
    t = torch.tensor([620.0])
    cond = torch.randn(1, 77, 768, generator=g)
    clip = torch.randn(1, Lv, 768, generator=g)
    sync = torch.randn(1, Ls, 768, generator=g)
    caught = {}
    h1 = m.triple_blocks[0].register_forward_hook(lambda mod, i, o: caught.__setitem__("triple", o))
    h2 = m.single_blocks[0].register_forward_hook(lambda mod, i, o: caught.__setitem__("single", o))
    with torch.inference_mode():
        y = m(x=x, t=t, cond=cond, clip_feat=clip, sync_feat=sync)["x"]
    h1.remove(), h2.remove()
    taps = {}
    yo = O.dit_forward(sd, c.heads, x, t, cond, clip, sync, taps=taps)
    check("triple block audio", taps["triple0"], caught["triple"][0])
    check("single block", taps["single0"], caught["single"])
    check("forward", yo, y)
    save("g4_blocks", x=x, t=t, cond=cond, clip=clip, sync=sync, y=y,
         triple_audio=caught["triple"][0], triple_v=caught["triple"][2], single=caught["single"])


# ----------------------------------------------------------------------------- G5
def g5():
    c = C.TINY
    sd = synth.synth_dit_state_dict(c)
    m = build_ref_dit(c, sd)
    out = {}
    g = torch.Generator().manual_seed(5)
    for tag, dur in (("a", 1.0), ("b", 2.2)):
        La, Lv, Ls = C.lengths(dur, c)
        x = torch.randn(2, 128, La, generator=g)
        t = torch.tensor([980.0, 980.0])
        cond = torch.randn(2, 77, 768, generator=g)
        clip = torch.randn(2, Lv, 768, generator=g)
        sync = torch.randn(2, Ls, 768, generator=g)
        with torch.inference_mode():
            y = m(x=x, t=t, cond=cond, clip_feat=clip, sync_feat=sync)["x"]
        check(f"tiny forward {tag} (La={La},Lv={Lv},Ls={Ls})", O.dit_forward(sd, c.heads, x, t, cond, clip, sync), y)
        for k, v in (("x", x), ("t", t), ("cond", cond), ("clip", clip), ("sync", sync), ("y", y)):
            out[f"{tag}_{k}"] = v
    save("g5_dit_tiny", **out)


# ----------------------------------------------------------------------------- G6
def g6():
    c = C.XXL
    t0 = time.time()
    sd = synth.synth_dit_state_dict(c)
    dsd = synth.synth_dac_state_dict(C.DAC48K)
    print(f"    synthesised xxl + DAC weights in {time.time() - t0:.0f}s")
    m = build_ref_dit(c, sd)
    dac = build_ref_dac(C.DAC48K, dsd)
    cond = synth.synth_conditioning(c, 1.0, t2a=True, sd=sd)
    md = ModelDict(foley_model=m, dac_model=dac, device=torch.device("cpu"))
    # capture per-step latents through the reference scheduler
    trace = []
    orig_step = NS.sched.FlowMatchDiscreteScheduler.step

    def spy(self, *a, **k):
        r = orig_step(self, *a, **k)
        trace.append(r[0].clone())
        return r
    NS.sched.FlowMatchDiscreteScheduler.step = spy
    gen = torch.Generator("cpu").manual_seed(1234)
    t0 = time.time()
    try:
        with torch.inference_mode():
            audio, sr = NS.utils.denoise_process_with_generator(
                {"siglip2_feat": cond["clip"], "syncformer_feat": cond["sync"]},
                {"text_feat": cond["text"], "uncond_text_feat": cond["uncond_text"]},
                1.0, md, ref_cfg(c), 1.0, 10, 1, "euler", generator=gen)
    finally:
        NS.sched.FlowMatchDiscreteScheduler.step = orig_step
    print(f"    reference C1 sampler ran in {time.time() - t0:.0f}s -> {tuple(audio.shape)} @ {sr}")
    gen = torch.Generator("cpu").manual_seed(1234)
    noise = torch.randn((1, 128, 50), generator=gen, dtype=torch.float32)
    otrace, taps = [], {}
    t0 = time.time()
    with torch.inference_mode():
        lat = O.sample_latents(sd, c.heads, noise, cond["text"], cond["uncond_text"], cond["clip"],
                               cond["sync"], 10, 1.0, trace=otrace)
        wav = O.dac_decode(dsd, lat)
        # block-level probes of the first forward, for localising drift on the GPU
        La, Lv, Ls = C.lengths(1.0, c)
        O.dit_forward(sd, c.heads, noise, torch.tensor([1000.0]), O.pad_or_trim_text(cond["text"]),
                      cond["clip"], cond["sync"], taps=taps)
    print(f"    oracle C1 sampler ran in {time.time() - t0:.0f}s")
    check("C1 latents", torch.stack(otrace), torch.stack(trace))
    check("C1 waveform", wav, audio)
    probes = {k: v[0, ::7, ::48].clone() for k, v in taps.items() if v.dim() == 3}
    norms = {k: float(v.double().norm()) for k, v in taps.items() if v.dim() == 3}
    save("g6_c1_xxl", noise=noise, latents=torch.stack(trace), waveform=audio,
         probe_names=np.array(list(probes.keys())), probes=torch.stack(
             [torch.nn.functional.pad(p, (0, 0, 0, 8 - p.shape[0])) for p in probes.values()]),
         probe_norms=np.array([norms[k] for k in probes.keys()]))


# ----------------------------------------------------------------------------- G7
def g7():
    c = C.TINY
    sd = synth.synth_dit_state_dict(c)
    dsd = synth.synth_dac_state_dict(C.DAC_TINY)
    m = build_ref_dit(c, sd)
    dac = build_ref_dac(C.DAC_TINY, dsd)
    md = ModelDict(foley_model=m, dac_model=dac, device=torch.device("cpu"))
    out = {}
    for tag, t2a, dur, g, bs, solver, steps in (
            ("cfg_euler", False, 1.0, 4.5, 2, "euler", 10),
            ("t2a_nocfg", True, 1.0, 1.0, 1, "euler", 10),
            ("heun", False, 1.0, 4.5, 1, "heun-2", 10),
            ("midpoint", False, 1.0, 4.5, 1, "midpoint-2", 10),
            ("kutta", False, 1.0, 4.5, 1, "kutta-4", 12),
            ("v2a_2s", False, 2.0, 3.0, 2, "euler", 12)):
        cond = synth.synth_conditioning(c, dur, t2a=t2a, sd=sd)
        if hasattr(m, "_text_len_fixed"):
            del m._text_len_fixed
        gen = torch.Generator("cpu").manual_seed(1234)
        with torch.inference_mode():
            audio, ref_lat = run_ref_sampler(md, c, cond, dur, g, steps, bs, solver, gen)
        gen = torch.Generator("cpu").manual_seed(1234)
        noise = torch.randn((bs, 128, int(dur * 50)), generator=gen, dtype=torch.float32)
        with torch.inference_mode():
            lat = O.sample_latents(sd, c.heads, noise, cond["text"], cond["uncond_text"], cond["clip"],
                                   cond["sync"], steps, g, solver)
            wav = O.dac_decode(dsd, lat)
        check(f"sampler {tag} latents", lat, ref_lat, 3e-5)
        check(f"sampler {tag}", wav, audio, 3e-5)
        out[tag + "_noise"], out[tag + "_latents"] = noise, ref_lat     # the REFERENCE's final latents (scheduler spy)
        out[tag + "_wave_s5"] = audio[..., ::5]   # subsampled waveform keeps the fixture small
    save("g7_sampler", **out)


# ----------------------------------------------------------------------------- G8
def g8():
    """fp8 weight-only storage (utils.py:316-485):
    (a) which modules of the DiT the reference wraps (its deny list never matches: every Linear /
        Conv1d, nothing else - learned feature rows and position tables stay full precision);
    (b) a wrapped Linear and a wrapped channels-last Conv1d forward (weight fp8 -> activation dtype,
        bias untouched);
    (c) the TimestepEmbedder under bf16 autocast: embed_layers.py:134 casts the sinusoid features to
        `mlp[0].weight.dtype`, i.e. to fp8 once the layer is wrapped, and the wrapper then casts the
        first bias to fp8 as well (utils.py:362-363) - both get rounded through fp8 ("Q14")."""
    c = C.TINY
    sd = synth.synth_dit_state_dict(c)
    out = {}
    g = torch.Generator().manual_seed(8)
    xl = torch.randn(5, c.hidden, generator=g)
    xc = torch.randn(2, 9, c.hidden, generator=g)           # channels-last conv input
    tt = torch.tensor([700.0, 3.0])
    out["xl"], out["xc"], out["t"] = xl, xc, tt
    for q in ("fp8_e4m3fn", "fp8_e5m2"):
        qd = torch.float8_e4m3fn if q == "fp8_e4m3fn" else torch.float8_e5m2
        m = build_ref_dit(c, sd)
        counts, _saved = NS.utils._wrap_fp8_inplace(m, quantization=q, state_dict=None)
        wrapped = sorted(n for n, mod in m.named_modules() if type(mod).__name__ == "FP8WeightWrapper")
        assert len(wrapped) == sum(counts.values())
        n_lin = sum(1 for k, v in sd.items() if k.endswith(".weight") and v.dim() >= 2)
        print(f"  {q}: {len(wrapped)} wrapped modules; state dict has {n_lin} >=2-D '.weight' tensors")
        out[q + "_wrapped"] = np.array("\n".join(wrapped))
        with torch.inference_mode():
            lin = m.triple_blocks[0].audio_mlp.fc1          # wrapped nn.Linear
            conv = m.single_blocks[0].linear1                # ChannelLastConv1d -> wrapped nn.Conv1d, channels-last input
            yl, yc = lin(xl), conv(xc)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                yt = m.time_in(tt)
        rq = lambda w: w.to(qd).to(torch.float32)
        check(f"{q} wrapped linear", F.linear(xl, rq(sd["triple_blocks.0.audio_mlp.fc1.weight"]), sd["triple_blocks.0.audio_mlp.fc1.bias"]), yl)
        check(f"{q} wrapped conv1d (channels-last)", O.conv1d_cl(xc, rq(sd["single_blocks.0.linear1.weight"]), sd["single_blocks.0.linear1.bias"], 1), yc)
        check(f"{q} time_in under autocast", O.time_embed_fp8_autocast(sd, tt, qd), yt.float(), 2e-2)
        out[q + "_lin_y"], out[q + "_conv_y"], out[q + "_time_y"] = yl, yc, yt.float()
    save("g8_fp8", **out)


# ----------------------------------------------------------------------------- G9
def g9():
    dsd = synth.synth_dac_state_dict(C.DAC48K)
    dac = build_ref_dac(C.DAC48K, dsd)
    g = torch.Generator().manual_seed(9)
    z = torch.randn(1, 128, 10, generator=g)
    with torch.inference_mode():
        y = dac.decode(z)
    check("dac full width", O.dac_decode(dsd, z), y)
    save("g9_dac", z=z, y=y)


# ----------------------------------------------------------------------------- G10
def g10():
    """DAC encoder half (row N4): narrow codec (32 -> 64 -> 128 channels, rates (2,3), odd stride
    included) on ragged-length audio through preprocess + encode, the posterior's mean / std, and the
    real 128 -> 4096-channel encoder on two latent frames of audio."""
    out = {}
    for tag, dc, n_samp in (("tiny", C.DAC_ENC_TINY, 6 * 17 - 4), ("full", C.DAC48K, 960 * 2)):
        dsd = synth.synth_dac_state_dict(dc, encoder=True)
        m = NS.dac.DAC(encoder_dim=dc.encoder_dim, encoder_rates=list(dc.encoder_rates), latent_dim=dc.latent_dim,
                       decoder_dim=dc.decoder_dim, decoder_rates=list(dc.rates), n_codebooks=9, codebook_size=1024,
                       codebook_dim=8, quantizer_dropout=False, sample_rate=dc.sample_rate, continuous=True)
        missing, unexpected = m.load_state_dict(dsd, strict=False)
        assert not unexpected and all(k.startswith("quantizer") for k in missing), (missing[:5], unexpected[:5])
        g = torch.Generator().manual_seed(10)
        audio = 0.5 * torch.randn(2 if tag == "tiny" else 1, 1, n_samp, generator=g)
        with torch.inference_mode():
            a = m.preprocess(audio, dc.sample_rate)
            post = m.encode(a)[0]
        hop = 1
        for r in dc.encoder_rates:
            hop *= r
        assert m.hop_length == hop
        ap = O.dac_preprocess(audio, hop)
        check(f"dac preprocess {tag}", ap, a)
        check(f"dac encode {tag}", O.dac_encode(dsd, ap, dc.encoder_rates), post.parameters)
        mean, std = O.gaussian_posterior(post.parameters)
        check(f"posterior mean {tag}", mean, post.mode())
        check(f"posterior std {tag}", std, post.std)
        out[tag + "_audio"], out[tag + "_params"], out[tag + "_std"] = audio, post.parameters, post.std
    save("g10_dac_encode", **out)


def g11():
    """Video-to-Audio conditioning: the reference's Synchformer visual extractor, built exactly as
    Synchformer.__init__ builds it (synchformer.py:22-28: divided space-time MotionFormer + spatial
    aggregation layer, Identity over time), loaded with synthesised weights and run as Synchformer.forward
    runs it (:44-50) on two overlapping 16-frame segments - pins host/encoders.py::synchformer_segments;
    plus the frame-index selection of nodes.py:293-317 for a few (clip length, duration, fps) cases."""
    from foley_amd.host import encoders as E
    mf = ref_harness.load_motionformer()
    schema = E.synchformer_schema()
    sd = synth.materialize(schema)
    own = mf.state_dict()
    p = "vfeat_extractor."
    missing = [k for k in own if p + k not in sd and not k.startswith("patch_embed.proj")]
    assert not missing, missing                      # the schema covers every tensor the forward touches
    assert all(tuple(own[k[len(p):]].shape) == tuple(v.shape) for k, v in sd.items())
    mf.load_state_dict({k[len(p):]: v for k, v in sd.items()}, strict=False)
    frames = synth.synth_tensor("g11.frames", (24, 3, 224, 224), 0.55)          # pre-processed frames in ~[-1, 1]
    segs = torch.stack([frames[0:16], frames[8:24]]).unsqueeze(0)                 # [B=1, S=2, T=16, C, H, W]
    with torch.inference_mode():
        vis = segs.permute(0, 1, 3, 2, 4, 5)                                      # Synchformer.forward (synchformer.py:46)
        ref = mf(vis)                                                             # [1, 2, 8, 768]
        mine = E.synchformer_segments(sd, segs[0])
        feat = E.encode_video_with_sync(sd, frames)
    check("synchformer segments", mine, ref[0], tol=2e-5)
    check("encode_video_with_sync", feat, ref.reshape(1, 16, 768), tol=2e-5)
    out = {"sync_feat": ref.reshape(1, 16, 768)}
    for tag, (total, dur, fps) in {"a": (120, 5.0, 24.0), "b": (60, 5.0, 24.0), "c": (300, 10.0, 30.0), "d": (17, 1.0, 16.0)}.items():
        n = int(dur * fps)                                                        # nodes.py:296-317, restated with the reference's expressions
        out[f"idx8_{tag}"] = torch.linspace(0, n - 1, int(dur * 8)).long()
        out[f"idx25_{tag}"] = torch.linspace(0, n - 1, int(dur * 25)).long()
        out[f"case_{tag}"] = torch.tensor([total, dur, fps], dtype=torch.float64)
    save("g11_v2a", **out)


def g12():
    """The BENCHMARKED precision pinned to the reference itself (not to the oracle on rounded weights): the
    reference run exactly as its sampler runs a bf16 model - parameters `.to(bfloat16)`, inputs cast to the
    parameter dtype, `torch.autocast(bfloat16)` (utils.py:222-234) - on this container's CPU:
      fwd_*      one tiny-config forward on golden g5's inputs: bf16 output next to the fp32 output;
      cfg_*      the 10-step CFG 4.5 Euler run of g7 ('cfg_euler', bs 2): final latents + waveform in
                 bf16 next to the fp32 run (noise drawn in the MODEL dtype like the reference does);
      c5_*       config C5's structure at reduced width: bf16 model wrapped by the reference's own
                 `_wrap_fp8_inplace(fp8_e4m3fn)`, ONE forward at the 30 s shapes (La 1500, Lv 240,
                 Ls 736, negative-prompt CFG pair = batch 2)."""
    c = C.TINY
    sd = synth.synth_dit_state_dict(c)
    dsd = synth.synth_dac_state_dict(C.DAC_TINY)
    out = {}
    m32 = build_ref_dit(c, sd)
    m16 = build_ref_dit(c, sd).to(torch.bfloat16)
    bf = lambda t: t.to(torch.bfloat16)
    # ---- single forward
    g = torch.Generator().manual_seed(5)
    La, Lv, Ls = C.lengths(1.0, c)
    x = torch.randn(2, 128, La, generator=g)
    t = torch.tensor([980.0, 980.0])
    cond = torch.randn(2, 77, 768, generator=g)
    clip = torch.randn(2, Lv, 768, generator=g)
    sync = torch.randn(2, Ls, 768, generator=g)
    with torch.inference_mode():
        y32 = m32(x=x, t=t, cond=cond, clip_feat=clip, sync_feat=sync)["x"]
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y16 = m16(x=bf(x), t=t, cond=bf(cond), clip_feat=bf(clip), sync_feat=bf(sync))["x"]
    print(f"    forward: reference bf16 vs its own fp32: rel {rel(y16.float(), y32):.3e}")
    out["fwd_y32"], out["fwd_y16"] = y32, y16.float()      # inputs = golden g5's a_* tensors (same generator seed / order)
    # ---- 10-step CFG run, fp32 and bf16
    dac = build_ref_dac(C.DAC_TINY, dsd)
    cnd = synth.synth_conditioning(c, 1.0, t2a=False, sd=sd)
    gen = torch.Generator("cpu").manual_seed(1234)
    noise16 = torch.randn((2, 128, 50), generator=gen, dtype=torch.bfloat16)      # the bf16 draw of utils.py:114-121
    orig_prep = NS.utils.prepare_latents_with_generator
    for tag, m in (("b16", m16), ("f32", m32)):
        if hasattr(m, "_text_len_fixed"):
            del m._text_len_fixed
        md = ModelDict(foley_model=m, dac_model=dac, device=torch.device("cpu"))
        gen = torch.Generator("cpu").manual_seed(1234)
        if tag == "f32":     # the fp32 model on the SAME (bf16-drawn) noise: the two runs differ by arithmetic only
            NS.utils.prepare_latents_with_generator = lambda *a, **k: noise16.float().clone()
        try:
            with torch.inference_mode():
                audio, lat = run_ref_sampler(md, c, cnd, 1.0, 4.5, 10, 2, "euler", gen)
        finally:
            NS.utils.prepare_latents_with_generator = orig_prep
        out[f"cfg_{tag}_latents"], out[f"cfg_{tag}_wave_s5"] = lat, audio.float()[..., ::5]
    out["cfg_noise_b16"] = noise16.float()
    print(f"    10-step CFG on the same noise: reference bf16 vs fp32: latents rel {rel(out['cfg_b16_latents'], out['cfg_f32_latents']):.3e}, "
          f"waveform rel {rel(out['cfg_b16_wave_s5'], out['cfg_f32_wave_s5']):.3e}")
    # ---- C5 structure: bf16 + the reference's fp8 wrapper, 30 s shapes, one forward
    La, Lv, Ls = C.lengths(30.0, c)
    g = torch.Generator().manual_seed(55)
    x5 = torch.randn(1, 128, La, generator=g).repeat(2, 1, 1)                      # CFG pair shares the latents
    t5 = torch.tensor([620.0, 620.0])
    cn5 = synth.synth_conditioning(c, 30.0, t2a=True, sd=sd)
    pad = lambda a: F.pad(a, (0, 0, 0, 77 - a.shape[1]))
    text5 = torch.cat([pad(cn5["uncond_text"]), pad(cn5["text"])])
    clip5, sync5 = cn5["clip"].repeat(2, 1, 1), cn5["sync"].repeat(2, 1, 1)
    m8 = build_ref_dit(c, sd).to(torch.bfloat16)
    NS.utils._wrap_fp8_inplace(m8, quantization="fp8_e4m3fn", state_dict=None)
    with torch.inference_mode():
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y8 = m8(x=bf(x5), t=t5, cond=bf(text5), clip_feat=bf(clip5), sync_feat=bf(sync5))["x"]
            y16b = m16(x=bf(x5), t=t5, cond=bf(text5), clip_feat=bf(clip5), sync_feat=bf(sync5))["x"]
        y32b = m32(x=x5, t=t5, cond=text5, clip_feat=clip5, sync_feat=sync5)["x"]
    print(f"    C5 shapes (La={La}, Lv={Lv}, Ls={Ls}): fp8-wrapped bf16 vs fp32 rel {rel(y8.float(), y32b):.3e}, "
          f"plain bf16 vs fp32 {rel(y16b.float(), y32b):.3e}")
    out["c5_t"] = t5                                          # x: torch.Generator().manual_seed(55) -> randn(1,128,1500)
    out["c5_y8"], out["c5_y16"], out["c5_y32"] = y8.float()[1:, :, ::8], y16b.float()[1:, :, ::8], y32b[1:, :, ::8]   # cond half, every 8th frame
    save("g12_bf16_ref", **out)


def g13():
    """fp16 compute (the loader's `precision=fp16`, or `auto` on an fp16 checkpoint: nodes.py:89-106 loads the
    parameters as float16 and utils.py:229-234 runs the model under torch.autocast(float16)), pinned to the
    reference itself like g12 pins bf16: parameters `.to(float16)`, inputs in the parameter dtype, autocast fp16, on
    this container's CPU - one tiny-config forward on g5's inputs and the 10-step CFG 4.5 Euler run (noise drawn in
    fp16, utils.py:114-121), each next to the fp32 run on the same inputs."""
    c = C.TINY
    sd = synth.synth_dit_state_dict(c)
    dsd = synth.synth_dac_state_dict(C.DAC_TINY)
    out = {}
    m32 = build_ref_dit(c, sd)
    m16 = build_ref_dit(c, sd).to(torch.float16)
    hf = lambda t: t.to(torch.float16)
    g = torch.Generator().manual_seed(5)
    La, Lv, Ls = C.lengths(1.0, c)
    x = torch.randn(2, 128, La, generator=g)
    t = torch.tensor([980.0, 980.0])
    cond = torch.randn(2, 77, 768, generator=g)
    clip = torch.randn(2, Lv, 768, generator=g)
    sync = torch.randn(2, Ls, 768, generator=g)
    with torch.inference_mode():
        y32 = m32(x=x, t=t, cond=cond, clip_feat=clip, sync_feat=sync)["x"]
        with torch.autocast("cpu", dtype=torch.float16):
            y16 = m16(x=hf(x), t=t, cond=hf(cond), clip_feat=hf(clip), sync_feat=hf(sync))["x"]
    print(f"    forward: reference fp16 vs its own fp32: rel {rel(y16.float(), y32):.3e}")
    out["fwd_y32"], out["fwd_y16"] = y32, y16.float()
    dac = build_ref_dac(C.DAC_TINY, dsd)
    cnd = synth.synth_conditioning(c, 1.0, t2a=False, sd=sd)
    gen = torch.Generator("cpu").manual_seed(1234)
    noise16 = torch.randn((2, 128, 50), generator=gen, dtype=torch.float16)
    orig_prep = NS.utils.prepare_latents_with_generator
    for tag, m in (("f16", m16), ("f32", m32)):
        if hasattr(m, "_text_len_fixed"):
            del m._text_len_fixed
        md = ModelDict(foley_model=m, dac_model=dac, device=torch.device("cpu"))
        gen = torch.Generator("cpu").manual_seed(1234)
        if tag == "f32":
            NS.utils.prepare_latents_with_generator = lambda *a, **k: noise16.float().clone()
        try:
            with torch.inference_mode():
                audio, lat = run_ref_sampler(md, c, cnd, 1.0, 4.5, 10, 2, "euler", gen)
        finally:
            NS.utils.prepare_latents_with_generator = orig_prep
        out[f"cfg_{tag}_latents"], out[f"cfg_{tag}_wave_s5"] = lat, audio.float()[..., ::5]
    out["cfg_noise_f16"] = noise16.float()
    print(f"    10-step CFG on the same noise: reference fp16 vs fp32: latents rel {rel(out['cfg_f16_latents'], out['cfg_f32_latents']):.3e}, "
          f"waveform rel {rel(out['cfg_f16_wave_s5'], out['cfg_f32_wave_s5']):.3e}")
    save("g13_fp16_ref", **out)


class _StopAfterCapture(Exception):
    pass


def ref_forward_inside_sampler(m, c, cond, dur, guidance, steps, it, noise=None, seed=1234):
    """One model call of the reference's OWN sampler loop (utils.py:125-258) at loop iteration `it`: the
    sampler assembles its inputs itself ([uncond ; cond] concatenation, 77-token padding, the casts to the
    parameter dtype, the autocast region, utils.py:159-239) - a spy on `forward` answers iterations < it with
    zeros (an Euler step with v = 0 leaves the latents at the drawn noise), runs the real model at `it`,
    captures (x, t, y) and stops the loop.  `noise` overrides the draw (the fp32 run of a pair takes the
    16-bit run's noise, so the two differ by arithmetic only)."""
    dac = types.SimpleNamespace(sample_rate=48000, parameters=lambda: iter(()), decode=None)
    md = ModelDict(foley_model=m, dac_model=dac, device=torch.device("cpu"))
    if hasattr(m, "_text_len_fixed"):
        del m._text_len_fixed
    calls, cap = [0], {}
    orig_fwd = m.forward

    def spy(*a, **k):
        i = calls[0]
        calls[0] += 1
        if i < it:
            return {"x": torch.zeros_like(k["x"])}
        cap["x"], cap["t"] = k["x"].clone(), k["t"].clone()
        t0 = time.time()
        cap["y"] = orig_fwd(*a, **k)["x"].clone()
        cap["secs"] = time.time() - t0
        raise _StopAfterCapture()
    m.forward = spy
    orig_prep = NS.utils.prepare_latents_with_generator
    if noise is not None:
        NS.utils.prepare_latents_with_generator = lambda *a, **k: noise.clone()
    try:
        NS.utils.denoise_process_with_generator(
            {"siglip2_feat": cond["clip"], "syncformer_feat": cond["sync"]},
            {"text_feat": cond["text"], "uncond_text_feat": cond["uncond_text"]},
            dur, md, ref_cfg(c), guidance, steps, 1, "euler", generator=torch.Generator("cpu").manual_seed(seed))
    except _StopAfterCapture:
        pass
    finally:
        del m.forward
        NS.utils.prepare_latents_with_generator = orig_prep
    assert calls[0] == it + 1
    return cap


_FULL = {}


def g14():
    """G14 / G15 / G16 - the three benchmarked single-GPU configurations reference-checked AT FULL SIZE (round-4
    verdict item 2): one model call of the reference's own sampler loop, mid-schedule (iteration 25 of 50,
    CFG 4.5, so the batch is the [uncond ; cond] pair), at xxl width and FULL depth (18 + 36 blocks):
      g14  C2: text-to-audio 5 s  (La 250, Lv 40, Ls 112; empty visual features)      fp32, and bf16 as the
      g15  C3: video-to-audio 5 s (same lengths, dense SigLIP2 / Synchformer rows)     sampler runs a bf16 model
      g16  C5: 30 s (La 1500, Lv 240, Ls 736), negative-prompt CFG, bf16 parameters wrapped by the reference's
           `_wrap_fp8_inplace(fp8_e4m3fn)` (utils.py:316-485) under bf16 autocast - at depth 1+1 and full depth,
           each next to the fp32 run of the un-quantised model on the same inputs.
    The 16-bit runs draw their noise in the model dtype like the reference does; the fp32 runs take that
    noise.  Stored: y (every 2nd / 16th frame of both CFG halves; the 16-bit outputs as float32 images), the
    input noise is re-drawn by the tests from the same seed and verified against `x_sum`."""
    if "done" in _FULL:
        return
    _FULL["done"] = True
    it, steps, g = 25, 50, 4.5
    bf = torch.bfloat16

    def pair(tag, m32, m16, c, cond, dur, stride, out):
        c16 = ref_forward_inside_sampler(m16, c, cond, dur, g, steps, it)
        noise = c16["x"][:1].float()
        c32 = ref_forward_inside_sampler(m32, c, cond, dur, g, steps, it, noise=noise)
        assert torch.equal(c32["x"], c16["x"].float()) and torch.equal(c32["t"], c16["t"])
        d0 = rel(c16["y"].float(), c32["y"])
        print(f"    {tag}: reference 16-bit vs its own fp32: rel {d0:.3e}  (forward {c32['secs']:.0f}s fp32, {c16['secs']:.0f}s 16-bit)")
        out[tag + "_t"] = c32["t"]
        out[tag + "_x_sum"] = c32["x"].double().sum(dim=(0, 1)).float()      # per-frame checksum of the model input
        out[tag + "_y32"] = c32["y"][:, :, ::stride]
        out[tag + "_y16"] = c16["y"].float()[:, :, ::stride]
        return c32, c16

    def oracle_check(tag, sd, c, cond, c32, dur):
        if not CHECK:
            return
        La, Lv, Ls = C.lengths(dur, c)
        e_clip = sd["empty_clip_feat"].view(1, 1, -1).expand(1, Lv, -1)
        e_sync = sd["empty_sync_feat"].view(1, 1, -1).expand(1, Ls, -1)
        with torch.inference_mode():
            yo = O.dit_forward(sd, c.heads, c32["x"], c32["t"].float(),
                               torch.cat([O.pad_or_trim_text(cond["uncond_text"]), O.pad_or_trim_text(cond["text"])]),
                               torch.cat([e_clip, cond["clip"]]), torch.cat([e_sync, cond["sync"]]))
        check(tag + " forward", yo, c32["y"], 2e-5)

    # ---- depth 1+1 at C5 shapes (its own synthesised weights, like g4)
    c11 = C.DiTConfig(name="xxl-1-1", depth_triple=1, depth_single=1)
    sd11 = synth.synth_dit_state_dict(c11)
    cn11 = synth.synth_conditioning(c11, 30.0, t2a=True, sd=sd11)
    m32 = build_ref_dit(c11, sd11)
    m8 = build_ref_dit(c11, sd11).to(bf)
    NS.utils._wrap_fp8_inplace(m8, quantization="fp8_e4m3fn", state_dict=None)
    o16 = {}
    with torch.inference_mode():
        c32, _ = pair("d1", m32, m8, c11, cn11, 30.0, 16, o16)
    oracle_check("g16 depth 1+1", sd11, c11, cn11, c32, 30.0)
    del m32, m8, sd11
    # ---- full depth
    c = C.XXL
    t0 = time.time()
    sd = synth.synth_dit_state_dict(c)
    print(f"    synthesised xxl weights in {time.time() - t0:.0f}s")
    m32 = build_ref_dit(c, sd)
    m16 = build_ref_dit(c, sd).to(bf)
    with torch.inference_mode():
        o14, o15 = {}, {}
        c32, _ = pair("c2", m32, m16, c, synth.synth_conditioning(c, 5.0, t2a=True, sd=sd), 5.0, 2, o14)
        oracle_check("g14", sd, c, synth.synth_conditioning(c, 5.0, t2a=True, sd=sd), c32, 5.0)
        save("g14_c2_full", **o14)
        c32, _ = pair("c3", m32, m16, c, synth.synth_conditioning(c, 5.0, t2a=False, sd=sd), 5.0, 2, o15)
        oracle_check("g15", sd, c, synth.synth_conditioning(c, 5.0, t2a=False, sd=sd), c32, 5.0)
        save("g15_c3_full", **o15)
        NS.utils._wrap_fp8_inplace(m16, quantization="fp8_e4m3fn", state_dict=None)      # m16 becomes the fp8-wrapped model
        cn5 = synth.synth_conditioning(c, 30.0, t2a=True, sd=sd)
        c32, _ = pair("full", m32, m16, c, cn5, 30.0, 16, o16)
        oracle_check("g16 full depth", sd, c, cn5, c32, 30.0)
    save("g16_c5_full", **o16)


# ----------------------------------------------------------------------------- G17 / G18
def _traced_ref_sampler(md, c, cond, dur, g, steps, gen, noise=None):
    """The reference's denoise_process_with_generator (utils.py:125-258) with a spy on the scheduler step
    (scheduling_flow_match_discrete.py:262-297) that keeps the latents after EVERY iteration, and a spy on
    prepare_latents_with_generator (utils.py:111-121) that records (or overrides) the drawn noise."""
    trace, drawn = [], {}
    orig_step = NS.sched.FlowMatchDiscreteScheduler.step
    orig_prep = NS.utils.prepare_latents_with_generator

    def step_spy(self, *a, **k):
        r = orig_step(self, *a, **k)
        trace.append(r[0].float().clone())
        if len(trace) % 10 == 0:
            print(f"      iteration {len(trace)}", flush=True)
        return r

    def prep_spy(*a, **k):
        drawn["noise"] = (noise.clone() if noise is not None else orig_prep(*a, **k))
        return drawn["noise"].clone()
    NS.sched.FlowMatchDiscreteScheduler.step = step_spy
    NS.utils.prepare_latents_with_generator = prep_spy
    if hasattr(md.foley_model, "_text_len_fixed"):
        del md.foley_model._text_len_fixed
    t0 = time.time()
    try:
        with torch.inference_mode():
            audio, sr = NS.utils.denoise_process_with_generator(
                {"siglip2_feat": cond["clip"], "syncformer_feat": cond["sync"]},
                {"text_feat": cond["text"], "uncond_text_feat": cond["uncond_text"]},
                dur, md, ref_cfg(c), g, steps, 1, "euler", generator=gen)
    finally:
        NS.sched.FlowMatchDiscreteScheduler.step = orig_step
        NS.utils.prepare_latents_with_generator = orig_prep
    assert sr == 48000 and len(trace) == steps
    return audio.float(), torch.stack(trace), drawn["noise"], time.time() - t0


G17_CHECKPOINTS = (1, 10, 25, 40, 50)


def g17():
    """G17 / G18 - north_star's gate AT THE HEADLINE CONFIGURATION (round-5 verdict item 1): the reference's own
    sampler loop, full xxl model (18 + 36 blocks), C2 = text-to-audio 5 s, 50 Euler iterations, CFG 4.5, bs 1,
    seed 1234, through its DAC decoder (utils.py:125-258, scheduling_flow_match_discrete.py:210-297,
    dac_vae/model/dac.py:280-303).
      g17  fp32 parameters: the drawn noise, the latents after iterations {1, 10, 25, 40, 50}, the norm of the
           latents after every iteration, the 48 kHz waveform (every 5th sample).
      g18  the loop as the reference runs a bf16 model (parameters .to(bfloat16), autocast, noise drawn in
           bf16): final latents + waveform; next to it the fp32 model on THAT noise, so that
           d0 = |bf16 run - fp32 run| / |fp32 run| is the reference's own 50-iteration bf16 drift."""
    if "loop" in _FULL:          # "g17" and "g18" name the same generator
        return
    _FULL["loop"] = True
    c = C.XXL
    t0 = time.time()
    sd = synth.synth_dit_state_dict(c)
    dsd = synth.synth_dac_state_dict(C.DAC48K)
    print(f"    synthesised xxl + DAC weights in {time.time() - t0:.0f}s")
    dac = build_ref_dac(C.DAC48K, dsd)
    cond = synth.synth_conditioning(c, 5.0, t2a=True, sd=sd)
    m32 = build_ref_dit(c, sd)
    md32 = ModelDict(foley_model=m32, dac_model=dac, device=torch.device("cpu"))
    idx = [i - 1 for i in G17_CHECKPOINTS]
    # ---- g17: fp32, noise drawn in fp32 from seed 1234
    audio, trace, noise, secs = _traced_ref_sampler(md32, c, cond, 5.0, 4.5, 50, torch.Generator("cpu").manual_seed(1234))
    print(f"    reference C2 fp32 loop ran in {secs:.0f}s -> {tuple(audio.shape)}")
    chk = torch.randn((1, 128, 250), generator=torch.Generator("cpu").manual_seed(1234), dtype=torch.float32)
    assert torch.equal(chk, noise)
    save("g17_c2_loop", noise=noise, checkpoints=np.array(G17_CHECKPOINTS), latents=trace[idx],
         latent_norms=trace.double().flatten(1).norm(dim=1).float(), wave_s5=audio[..., ::5],
         wave_norm=np.float64(audio.double().norm()), ref_secs=np.float64(secs))
    # ---- g18: the bf16 loop, then the fp32 model on the bf16-drawn noise
    m16 = build_ref_dit(c, sd).to(torch.bfloat16)
    md16 = ModelDict(foley_model=m16, dac_model=dac, device=torch.device("cpu"))
    a16, tr16, noise16, secs16 = _traced_ref_sampler(md16, c, cond, 5.0, 4.5, 50, torch.Generator("cpu").manual_seed(1234))
    assert noise16.dtype == torch.bfloat16
    print(f"    reference C2 bf16 loop ran in {secs16:.0f}s")
    del m16, md16
    a32, tr32, _, secs32 = _traced_ref_sampler(md32, c, cond, 5.0, 4.5, 50, torch.Generator("cpu").manual_seed(1234),
                                               noise=noise16.float())
    d0_lat = [rel(tr16[i], tr32[i]) for i in idx]
    d0_wave = rel(a16, a32)
    print(f"    reference bf16 loop vs its own fp32 loop on the same noise: latents after {G17_CHECKPOINTS} rel "
          f"{['%.3e' % d for d in d0_lat]}, waveform rel {d0_wave:.3e}")
    save("g18_c2_bf16_loop", noise_b16=noise16.float(), checkpoints=np.array(G17_CHECKPOINTS),
         latents_b16=tr16[idx], latents_f32=tr32[idx], wave_b16_s5=a16[..., ::5], wave_f32_s5=a32[..., ::5],
         d0_latents=np.array(d0_lat), d0_wave=np.float64(d0_wave))


ALL = {"g17": g17, "g18": g17, "g14": g14, "g15": g14, "g16": g14, "g13": g13, "g12": g12, "g1": g1, "g2": g2, "g3": g3, "g4": g4, "g5": g5, "g6": g6, "g7": g7, "g8": g8, "g9": g9, "g10": g10, "g11": g11}


def main():
    global NS, CHECK
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--no-check-oracle", action="store_true")
    a = ap.parse_args()
    CHECK = not a.no_check_oracle
    NS = ref_harness.load_reference()
    torch.set_num_threads(os.cpu_count() or 8)
    names = [n for n in a.only.split(",") if n] or list(ALL)
    for n in names:
        print(f"[{n}]")
        t0 = time.time()
        ALL[n]()
        print(f"  done in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
