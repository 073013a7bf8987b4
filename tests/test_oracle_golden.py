"""Pins the CPU oracle (oracle/foley_oracle.py) against golden vectors frozen from the REFERENCE
itself (tests/golden/*.npz, produced by tests/golden/make_golden.py in the build container).
Runs anywhere - no GPU, no /root/reference.  G6 (xxl, ~3 min of weight synthesis) is opt-in
(FOLEY_SLOW=1); it was checked when the fixture was generated.
"""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import golden, rel_err
from foley_amd.host import config as C, synth
from oracle import foley_oracle as O

torch.set_num_threads(min(8, os.cpu_count() or 1))


def test_g1_scheduler():
    g = golden("g1_scheduler")
    for n in (10, 50):
        assert torch.equal(O.flow_sigmas(n), g[f"sigmas_{n}"])
        assert torch.equal(O.flow_timesteps(O.flow_sigmas(n)), g[f"timesteps_{n}"])
    assert rel_err(O.flow_sigmas(10, 3.0), g["sigmas_10_shift3"]) < 1e-7
    for solver in ("euler", "heun-2", "midpoint-2", "kutta-4"):
        st, x = O.SolverState(O.flow_sigmas(8), solver), g["x0"]
        for i in range(8):
            x = st.step(g["v"][i], x)
            assert rel_err(x, g["trace_" + solver.replace("-", "_")][i]) < 1e-7, (solver, i)


def test_g2_rope():
    g = golden("g2_rope")
    for n in (77, 250):
        cos, sin = O.rope_table(O.rope_positions(n))
        assert torch.equal(cos, g[f"cos_{n}"]) and torch.equal(sin, g[f"sin_{n}"])
    cos, sin = O.rope_table(O.rope_positions(6))
    assert rel_err(O.apply_rope(g["rope_x"], cos, sin), g["rope_y"]) < 1e-7
    for la, lv in ((250, 40), (50, 8), (1500, 240), (55, 8), (251, 40)):
        assert torch.equal(O.interleaved_positions(la, lv)[1], g[f"pos_v_{la}_{lv}"])


def test_g3_layers():
    g = golden("g3_layers")
    sd = {k: synth.synth_tensor("g3." + k, s, sc) for k, s, sc in
          (("w1", (768, 256, 3), 0.03), ("w2", (256, 768, 3), 0.02), ("w3", (768, 256, 3), 0.03))}
    x = g["convmlp_x"]
    y = O.conv1d_cl(F.silu(O.conv1d_cl(x, sd["w1"], None, 1)) * O.conv1d_cl(x, sd["w3"], None, 1), sd["w2"], None, 1)
    assert rel_err(y, g["convmlp_y"]) < 1e-6
    assert rel_err(O.snake(g["snake_x"], g["snake_alpha"]), g["snake_y"]) < 1e-7
    dc = C.DACConfig(decoder_dim=128, rates=(5, 2))
    taps = {}
    y = O.dac_decode(synth.synth_dac_state_dict(dc), g["dac52_z"], rates=(5, 2), taps=taps)
    assert rel_err(y, g["dac52_y"]) < 1e-5 and rel_err(taps["stage0"], g["dac52_stage0"]) < 1e-5


def test_g4_full_width_blocks():
    g = golden("g4_blocks")
    c = C.DiTConfig(name="xxl-1-1", depth_triple=1, depth_single=1)
    sd = synth.synth_dit_state_dict(c)
    taps = {}
    with torch.inference_mode():
        y = O.dit_forward(sd, c.heads, g["x"], g["t"], g["cond"], g["clip"], g["sync"], taps=taps)
    assert rel_err(taps["triple0"], g["triple_audio"]) < 1e-5
    assert rel_err(taps["single0"], g["single"]) < 1e-5
    assert rel_err(y, g["y"]) < 1e-5


def test_g5_tiny_forward():
    g = golden("g5_dit_tiny")
    sd = synth.synth_dit_state_dict(C.TINY)
    for tag in ("a", "b"):
        with torch.inference_mode():
            y = O.dit_forward(sd, C.TINY.heads, *(g[f"{tag}_{k}"] for k in ("x", "t", "cond", "clip", "sync")))
        assert rel_err(y, g[tag + "_y"]) < 1e-5


@pytest.mark.parametrize("tag,t2a,dur,guid,bs,solver,steps", [
    ("cfg_euler", False, 1.0, 4.5, 2, "euler", 10), ("t2a_nocfg", True, 1.0, 1.0, 1, "euler", 10),
    ("heun", False, 1.0, 4.5, 1, "heun-2", 10), ("midpoint", False, 1.0, 4.5, 1, "midpoint-2", 10),
    ("kutta", False, 1.0, 4.5, 1, "kutta-4", 12), ("v2a_2s", False, 2.0, 3.0, 2, "euler", 12)])
def test_g7_sampler(tag, t2a, dur, guid, bs, solver, steps):
    g = golden("g7_sampler")
    sd, dsd = synth.synth_dit_state_dict(C.TINY), synth.synth_dac_state_dict(C.DAC_TINY)
    cond = synth.synth_conditioning(C.TINY, dur, t2a=t2a, sd=sd)
    with torch.inference_mode():
        lat = O.sample_latents(sd, C.TINY.heads, g[tag + "_noise"], cond["text"], cond["uncond_text"], cond["clip"],
                               cond["sync"], steps, guid, solver)
        wav = O.dac_decode(dsd, lat)
    assert rel_err(lat, g[tag + "_latents"]) < 1e-5
    assert rel_err(wav[..., ::5], g[tag + "_wave_s5"]) < 3e-5


def test_g9_dac_full_width():
    g = golden("g9_dac")
    with torch.inference_mode():
        y = O.dac_decode(synth.synth_dac_state_dict(C.DAC48K), g["z"])
    assert rel_err(y, g["y"]) < 1e-5


@pytest.mark.slow
def test_g6_c1_xxl():
    g = golden("g6_c1_xxl")
    sd = synth.synth_dit_state_dict(C.XXL)
    cond = synth.synth_conditioning(C.XXL, 1.0, t2a=True, sd=sd)
    trace = []
    with torch.inference_mode():
        lat = O.sample_latents(sd, C.XXL.heads, g["noise"], cond["text"], cond["uncond_text"], cond["clip"],
                               cond["sync"], 10, 1.0, trace=trace)
        wav = O.dac_decode(synth.synth_dac_state_dict(C.DAC48K), lat)
    assert rel_err(torch.stack(trace), g["latents"]) < 1e-5 and rel_err(wav, g["waveform"]) < 3e-5


@pytest.mark.slow
def test_g17_c2_loop_tail():
    """The oracle against the reference's own HEADLINE run (g17: xxl, 5 s, 50 Euler iterations, CFG 4.5): resumed from the
    reference's latents after iteration 40, the last ten iterations + the DAC decode must reproduce the reference's final
    latents and waveform (~2 min on 8 cores)."""
    g = golden("g17_c2_loop")
    cps = [int(c) for c in g["checkpoints"]]
    sd = synth.synth_dit_state_dict(C.XXL)
    cond = synth.synth_conditioning(C.XXL, 5.0, t2a=True, sd=sd)
    with torch.inference_mode():
        lat = O.sample_latents(sd, C.XXL.heads, g["latents"][cps.index(40)], cond["text"], cond["uncond_text"], cond["clip"],
                               cond["sync"], 50, 4.5, start_iter=40)
        wav = O.dac_decode(synth.synth_dac_state_dict(C.DAC48K), lat)
    e_lat, e_wav = rel_err(lat, g["latents"][cps.index(50)]), rel_err(wav[..., ::5], g["wave_s5"])
    print("oracle vs reference, C2 iterations 41-50: latents %.2e, waveform %.2e" % (e_lat, e_wav))
    assert e_lat < 1e-5 and e_wav < 3e-5


def test_g8_fp8_weight_only_semantics():
    """Reference FP8WeightWrapper / _wrap_fp8_inplace (utils.py:316-485): the wrapped set, wrapped
    Linear / channels-last Conv1d forwards, and the TimestepEmbedder-under-autocast quirk."""
    from foley_amd import nodes
    g = golden("g8_fp8")
    c = C.TINY
    sd = synth.synth_dit_state_dict(c)
    for q, qd in (("fp8_e4m3fn", torch.float8_e4m3fn), ("fp8_e5m2", torch.float8_e5m2)):
        wrapped = set(str(g[q + "_wrapped"]).split("\n"))
        mine = {k[:-7] for k, v in sd.items() if nodes.fp8_wrapped_key(k, v)}
        assert mine == wrapped and len(wrapped) == 56
        rq = lambda w: w.to(qd).to(torch.float32)
        yl = F.linear(g["xl"], rq(sd["triple_blocks.0.audio_mlp.fc1.weight"]), sd["triple_blocks.0.audio_mlp.fc1.bias"])
        yc = O.conv1d_cl(g["xc"], rq(sd["single_blocks.0.linear1.weight"]), sd["single_blocks.0.linear1.bias"], 1)
        assert rel_err(yl, g[q + "_lin_y"]) < 1e-6 and rel_err(yc, g[q + "_conv_y"]) < 1e-6
        assert rel_err(O.time_embed_fp8_autocast(sd, g["t"], qd), g[q + "_time_y"]) < 1e-2
        # the loader's rounding rule reproduces exactly those tensors (+ the first time-embedding bias under autocast)
        sdq = nodes.fp8_round_state_dict(sd, q, autocast=True)
        changed = {k for k in sd if not torch.equal(sd[k], sdq[k])}
        assert changed == {k + ".weight" for k in wrapped} | {"time_in.mlp.0.bias"}
        assert torch.equal(sdq["time_in.mlp.0.bias"], rq(sd["time_in.mlp.0.bias"]))


def test_g10_dac_encoder():
    """DAC encoder half (dac.py:47-95, 225-278) + posterior (vae_utils.py:24-31): narrow codec on
    ragged-length audio and the real 4096-channel encoder on two latent frames."""
    g = golden("g10_dac_encode")
    for tag, dc in (("tiny", C.DAC_ENC_TINY), ("full", C.DAC48K)):
        dsd = synth.synth_dac_state_dict(dc, encoder=True)
        hop = 1
        for r in dc.encoder_rates:
            hop *= r
        a = O.dac_preprocess(g[tag + "_audio"], hop)
        assert a.shape[-1] % hop == 0 and a.shape[-1] - g[tag + "_audio"].shape[-1] < hop
        with torch.inference_mode():
            params = O.dac_encode(dsd, a, dc.encoder_rates)
        assert params.shape == g[tag + "_params"].shape == (a.shape[0], 2 * dc.latent_dim, a.shape[-1] // hop)
        assert rel_err(params, g[tag + "_params"]) < 1e-5
        mean, std = O.gaussian_posterior(params)
        assert torch.equal(mean, params[:, :dc.latent_dim]) and rel_err(std, g[tag + "_std"]) < 1e-5


RESIZE_CASES = (((2, 3, 96, 160), (224, 373)), ((1, 3, 480, 640), (512, 512)), ((1, 3, 480, 640), (224, 298)),
                ((2, 1, 300, 224), (300, 224)), ((1, 2, 270, 480), (128, 128)), ((1, 1, 7, 3), (3, 7)), ((1, 1, 1, 9), (5, 1)),
                ((1, 1, 33, 31), (32, 32)))


@pytest.mark.parametrize("shape,size", RESIZE_CASES)
def test_resize_oracle_is_the_uint8_kernel_of_the_reference_pipeline(shape, size):
    """The frames' resize (nodes.py:184-196: v2.Resize(bicubic, antialias) on uint8 CPU tensors) dispatches to ATen's native uint8
    kernel; torchvision is not in the image, so the pin is that kernel itself - F.interpolate on uint8.  The oracle's integer
    restatement (and the tap tables the HIP pass consumes, host/encoders.py::aa_tables) must equal it bit for bit: up- and
    down-scaling, identity axes, windows cut by the borders, one-sample axes."""
    import numpy as np
    from foley_amd.host import encoders as E
    g = torch.Generator().manual_seed(sum(shape) + sum(size))
    fr = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    fr[..., : shape[-2] // 2, :] = fr[..., : shape[-2] // 2, :] // 64 * 85      # flat areas + hard edges: overshoot, saturation
    ref = F.interpolate(fr, size=size, mode="bicubic", antialias=True).numpy()
    got = O.resize_u8(fr.numpy(), size)
    assert got.dtype == np.uint8 and got.shape == ref.shape and np.array_equal(got, ref)
    for n_in, n_out in ((shape[-1], size[1]), (shape[-2], size[0])):
        if n_in == n_out:
            continue
        xmin, xsize, w, prec = E.aa_tables(n_in, n_out)
        x = fr.numpy().astype(np.int64)[0, 0, 0] if n_in == shape[-1] else fr.numpy().astype(np.int64)[0, 0, :, 0]
        want = O.resize_u8_axis(x.astype(np.uint8), 0, n_out)
        mine = np.array([np.clip(((1 << (prec - 1)) + int((x[xmin[i]:xmin[i] + xsize[i]] * w[i, :xsize[i]].astype(np.int64)).sum())) >> prec,
                                 0, 255) for i in range(n_out)], dtype=np.uint8)
        assert np.array_equal(mine, want), (n_in, n_out)
