"""CPU tests of the host logic: packers (layouts the kernels assume), schedule/solver/RoPE tables,
the C-ABI surface, the node layer.  No GPU compute here; the GEMM addressing model
(kernels.h) is *emulated* in a few lines of torch so that packers + descriptors are validated
against the oracle before any kernel runs.
"""
import math
import os
import re

import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT, golden, rel_err
from foley_amd.host import config as C, packers, runtime as rt, sampler, synth, tables
from oracle import foley_oracle as O


def emulate_conv_gemm(x_rows, Wp, seg, taps, dil, tap0=None):
    """D[r, n] = sum_k A'[r, k] Wp[n, k] with A' the virtual-row matrix of kernels.h."""
    R, Cc = x_rows.shape
    tap0 = -((taps - 1) // 2) * dil if tap0 is None else tap0
    b, q = torch.arange(R) // seg, torch.arange(R) % seg
    cols = []
    for j in range(taps):
        st = q + tap0 + j * dil
        ok = (st >= 0) & (st < seg)
        src = (b * seg + st.clamp(0, seg - 1))
        cols.append(x_rows[src] * ok[:, None])
    return torch.cat(cols, dim=1) @ Wp.t()


def _g(s):
    return torch.Generator().manual_seed(s)


# ----------------------------------------------------------------------------- packers
def test_conv_to_gemm_matches_conv1d():
    B, L, Cin, Cout = 2, 9, 16, 24
    x, w = torch.randn(B, L, Cin, generator=_g(1)), torch.randn(Cout, Cin, 3, generator=_g(2))
    ref = O.conv1d_cl(x, w, None, 1).reshape(B * L, Cout)
    out = emulate_conv_gemm(x.reshape(B * L, Cin), packers.conv_to_gemm(w), L, 3, 1)
    assert rel_err(out, ref) < 1e-6


@pytest.mark.parametrize("dil", [1, 3, 9])
def test_dilated_conv7_as_gemm(dil):
    B, T, Cc = 2, 40, 8
    x, w = torch.randn(B, Cc, T, generator=_g(3)), torch.randn(Cc, Cc, 7, generator=_g(4))
    ref = F.conv1d(x, w, None, dilation=dil, padding=3 * dil).transpose(1, 2).reshape(B * T, Cc)
    out = emulate_conv_gemm(x.transpose(1, 2).reshape(B * T, Cc), packers.conv_to_gemm(w), T, 7, dil)
    assert rel_err(out, ref) < 1e-6


@pytest.mark.parametrize("s", [2, 3, 4, 5, 8])
def test_conv_transpose_as_gemm(s):
    """Virtual rows q in [0, Tin] = [x[q-1] ; x[q]], N = (phase, Cout), t_out = q*s + p - pad."""
    B, Tin, Cin, Cout = 2, 11, 6, 4
    x, w, bias = torch.randn(B, Cin, Tin, generator=_g(5)), torch.randn(Cin, Cout, 2 * s, generator=_g(6)), torch.randn(Cout, generator=_g(7))
    ref = F.conv_transpose1d(x, w, bias, stride=s, padding=math.ceil(s / 2), output_padding=s % 2)
    assert ref.shape[-1] == Tin * s
    Wt, bt = packers.convT_to_gemm(w, s), bias.repeat(s)
    xr = x.transpose(1, 2)                                             # [B, Tin, Cin]
    pad = (s + 1) // 2
    out = torch.full((B, Tin * s * Cout), float("nan"))
    for b in range(B):
        xp = F.pad(xr[b], (0, 0, 1, 1))                                # row q+? : xp[q] = x[q-1], xp[q+1] = x[q]
        for q in range(Tin + 1):
            d = torch.cat([xp[q], xp[q + 1]]) @ Wt.t() + bt            # [s*Cout]
            rel = q * s * Cout - pad * Cout + torch.arange(s * Cout)
            ok = (rel >= 0) & (rel < Tin * s * Cout)
            out[b, rel[ok]] = d[ok]
    assert not torch.isnan(out).any()
    assert rel_err(out.view(B, Tin * s, Cout), ref.transpose(1, 2)) < 1e-6


def test_interleave_gate_and_qkv_permutation():
    w1, w3 = torch.randn(64, 8, generator=_g(8)), torch.randn(64, 8, generator=_g(9))
    p = packers.interleave_gate(w1, w3)
    assert torch.equal(p[0:32], w1[0:32]) and torch.equal(p[32:64], w3[0:32]) and torch.equal(p[64:96], w1[32:64])
    H, hd = 2, 4
    w = torch.randn(3 * H * hd, 5, generator=_g(10))
    x = torch.randn(7, 5, generator=_g(11))
    y = (x @ w.t()).view(7, H, hd, 3)                                  # reference '(H D K)'
    yp = (x @ packers.qkv_hdk_to_khd(w, H).t()).view(7, 3, H, hd)      # packed '(K H D)'
    for k in range(3):
        assert torch.equal(yp[:, k], y[..., k])
    b = torch.arange(3 * H * hd, dtype=torch.float32)
    assert torch.equal(packers.qkv_hdk_to_khd(b, H).view(3, H, hd)[1, 1, 2], b.view(H, hd, 3)[1, 2, 1])


def test_weight_norm_fold_spellings():
    g, v = torch.rand(6, 1, 1, generator=_g(12)) + 0.5, torch.randn(6, 4, 7, generator=_g(13))
    ref = O.weight_norm_fold(g, v)
    a = packers.fold_weight_norm({"c.parametrizations.weight.original0": g, "c.parametrizations.weight.original1": v}, "c")
    b = packers.fold_weight_norm({"c.weight_g": g, "c.weight_v": v}, "c")
    c = packers.fold_weight_norm({"c.weight": ref}, "c")
    assert rel_err(a, ref) < 1e-7 and rel_err(b, ref) < 1e-7 and torch.equal(c, ref)


def test_pack_dit_and_arena_roundtrip():
    cfg = C.TINY
    sd = synth.synth_dit_state_dict(cfg)
    packed = packers.pack_dit(sd, cfg, torch.bfloat16)
    D, Hc = cfg.hidden, cfg.conv_hidden
    assert packed["s0.w13.w"].shape == (2 * Hc, 3 * D) and packed["s0.w13.w"].dtype == torch.bfloat16
    assert packed["s0.w2.w"].shape == (D, 3 * Hc) and packed["t1.a_mod.w"].shape == (9 * D, D)
    assert packed["smod_all.w"].shape == (cfg.depth_single * 6 * D, D)
    assert torch.equal(packed["smod_all.w"][6 * D:12 * D].float(),
                       sd["single_blocks.1.modulation.linear.weight"].to(torch.bfloat16).float())
    assert packed["t0.a_mod.b"].dtype == torch.float32 and "final_layer.adaLN_modulation.1.weight" not in packed
    arena = packers.Arena.from_packed(packed, "cpu")
    for k, v in packed.items():
        assert torch.equal(arena.view(k), v), k
        assert (arena.view(k).data_ptr() - arena.buffer.data_ptr()) % 256 == 0
    # a second arena built from the layout table alone (what a receiving rank does) has equal views
    total, table = packers.arena_layout(packed)
    other = packers.Arena(total, table, "cpu")
    other.buffer.copy_(arena.buffer)
    assert torch.equal(other.view("s1.qkv.w"), packed["s1.qkv.w"])


def test_pack_dac_shapes():
    dc = C.DAC_TINY
    p = packers.pack_dac(synth.synth_dac_state_dict(dc), dc)
    assert p["dac.in.w"].shape == (1024, 7 * 128) and p["dac.0.up.w"].shape == (8 * 512, 2 * 1024)
    assert p["dac.0.up.b"].shape == (8 * 512,) and p["dac.4.2.c1.w"].shape == (32, 32)
    assert p["dac.out.w"].shape == (7 * 32,) and p["dac.out.alpha"].shape == (32,)


# ----------------------------------------------------------------------------- tables
def test_schedule_tables_match_golden():
    g = golden("g1_scheduler")
    for n in (10, 50):
        assert torch.equal(tables.sigma_grid(n), g[f"sigmas_{n}"])
        assert torch.equal(tables.model_timesteps(tables.sigma_grid(n)), g[f"timesteps_{n}"])
    assert rel_err(tables.sigma_grid(10, 3.0), g["sigmas_10_shift3"]) < 1e-7


@pytest.mark.parametrize("solver", tables.SOLVERS)
def test_solver_table_reproduces_reference_trace(solver):
    """Emulate the device update rule on CPU and compare with the reference scheduler's trace (G1)."""
    g = golden("g1_scheduler")
    x, xs, acc = g["x0"].clone(), torch.zeros_like(g["x0"]), torch.zeros_like(g["x0"])
    coef = tables.solver_table(tables.sigma_grid(8), solver, 8)
    for i in range(8):
        v = g["v"][i]
        w_new, w_acc, dt, w_store, flags = [float(c) for c in coef[i, :5]]
        flags = int(flags)
        a = torch.zeros_like(acc) if flags & tables.STEP_ACC_RESET else acc
        deriv = w_new * v + w_acc * a if w_acc else w_new * v
        base = xs if flags & tables.STEP_USE_SAVED else x
        if flags & tables.STEP_SAVE_X:
            xs = x.clone()
        x = base + deriv * dt
        acc = a + w_store * v
        assert rel_err(x, g["trace_" + solver.replace("-", "_")][i]) < 1e-6, (solver, i)


def test_rope_and_position_tables_match_golden():
    g = golden("g2_rope")
    cos, sin = tables.rope_table(500)
    assert torch.equal(cos[:77].repeat_interleave(2, 1), g["cos_77"])
    assert torch.equal(sin[:250].repeat_interleave(2, 1), g["sin_250"])
    assert torch.equal(cos[::7].repeat_interleave(2, 1), g["cos_500_s7"])
    for la, lv in ((250, 40), (50, 8), (1500, 240), (55, 8), (251, 40)):
        pa, pv = tables.interleaved_positions(la, lv)
        assert torch.equal(pv, g[f"pos_v_{la}_{lv}"]) and torch.equal(pa, 2 * torch.arange(la))
    tb = tables.build_tables(250, 40, 112, 77, 50, "euler", 1.0)
    assert tb["t_feat"].shape == (50, 256) and tb["sync_gather"].shape == (250,)
    assert torch.equal(tb["sync_gather"].long(), O.nearest_exact_index(250, 112))
    assert rel_err(tb["t_feat"], O.timestep_embedding(tb["timesteps"])) == 0.0


def test_nearest_exact_index_matches_interpolate_for_every_widget_duration():
    """F.interpolate(mode='nearest-exact') evaluates floor((i+0.5)*in/out) in float32 (hifi_foley.py:44,
    58,761); a float64 formula picks a neighbouring row for 29 of the 591 durations the sampler widget
    allows (1.0 .. 60.0 s, step 0.1).  The host's float32 emulation must agree with the op itself for
    all of them, and the interleaved RoPE must stay a pure re-indexing."""
    def aten(out_len, in_len):
        src = torch.arange(in_len, dtype=torch.float32).view(1, 1, in_len)
        return F.interpolate(src, size=out_len, mode="nearest-exact").view(-1).long()
    n_f64_differs = 0
    for k in range(10, 601):
        d = k / 10
        la, lv, ls = C.lengths(d)
        for out_len, in_len in ((la, ls), (la, lv), (lv, la)):
            ref = aten(out_len, in_len)
            assert torch.equal(tables.nearest_exact_index(out_len, in_len), ref), (d, out_len, in_len)
            assert torch.equal(O.nearest_exact_index(out_len, in_len), ref)
            i = torch.arange(out_len, dtype=torch.float64)
            f64 = torch.clamp(torch.floor((i + 0.5) * (in_len / out_len)).long(), max=in_len - 1)
            n_f64_differs += int(not torch.equal(f64, ref))
        pa, pv = tables.interleaved_positions(la, lv)      # raises if not a pure re-indexing
        assert pv.shape == (lv,) and int(pv.max()) < 2 * la
    assert n_f64_differs > 0    # the sweep does cover the durations where the precision matters


def test_lengths_rule():
    assert C.lengths(5.0) == (250, 40, 112) and C.lengths(1.0) == (50, 8, 16) and C.lengths(30.0) == (1500, 240, 736)
    assert C.XXL.conv_hidden == 4096 and C.XL.conv_hidden == 3840 and C.TINY.conv_hidden == 768


# ----------------------------------------------------------------------------- C ABI surface
def test_library_exports_every_declared_symbol():
    """dlopen works without a GPU; every function declared in include/foley_hip.h must resolve."""
    hdr = open(os.path.join(ROOT, "include", "foley_hip.h")).read()
    declared = set(re.findall(r"\b(foley_[a-z0-9_]+)\s*\(", hdr)) - {"foley_progress_cb"}
    lib = rt.load_library()
    assert declared == set(rt.EXPORTED_SYMBOLS), declared ^ set(rt.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.foley_abi_version() == rt.ABI_VERSION == 12


def test_no_cpu_fallback():
    """The product path refuses CPU tensors / a missing library instead of silently falling back."""
    with pytest.raises(rt.FoleyRuntimeError):
        rt.op_latent_rows(torch.zeros(1, 128, 8), 1, torch.zeros(8, 128))
    with pytest.raises(rt.FoleyRuntimeError):
        rt.load_library("/nonexistent/libfoley_hip.so")
    if not torch.cuda.is_available():
        with pytest.raises(rt.FoleyRuntimeError):
            rt.FoleyContext(C.TINY, C.DAC_TINY, torch.float32, torch.device("cpu"))
    src = open(os.path.join(ROOT, "comfyui-hunyuanvideo-foley_amd", "host", "sampler.py")).read()
    assert "oracle" not in src.replace("# oracle", "")


# ----------------------------------------------------------------------------- node layer
def test_node_mappings_and_widgets():
    import foley_amd
    m = foley_amd.NODE_CLASS_MAPPINGS
    assert list(m) == ["HunyuanModelLoader", "HunyuanDependenciesLoader", "HunyuanFoleySampler",
                       "HunyuanFoleyTorchCompile", "HunyuanBlockSwap", "SelectAudioFromBatch"]
    assert foley_amd.NODE_DISPLAY_NAME_MAPPINGS["HunyuanFoleySampler"] == "Hunyuan-Foley Sampler"
    s = m["HunyuanFoleySampler"]
    it = s.INPUT_TYPES()
    assert list(it["required"]) == ["hunyuan_model", "hunyuan_deps", "frame_rate", "duration", "prompt",
                                    "negative_prompt", "cfg_scale", "steps", "sampler", "batch_size", "seed",
                                    "force_offload"]
    assert list(it["optional"]) == ["image", "torch_compile_cfg", "block_swap_args"]
    assert it["required"]["cfg_scale"][1]["default"] == 4.5 and it["required"]["steps"][1]["default"] == 50
    assert it["required"]["sampler"][0] == ["euler", "heun-2", "midpoint-2", "kutta-4"]
    assert it["required"]["batch_size"][1]["max"] == 6 and it["required"]["seed"][1]["max"] == 0xffffffffffffffff
    assert s.RETURN_TYPES == ("AUDIO", "AUDIO") and s.RETURN_NAMES == ("audio_first", "audio_batch")
    assert s.FUNCTION == "generate_audio" and s.CATEGORY == "audio/HunyuanFoley"
    ml = m["HunyuanModelLoader"].INPUT_TYPES()["required"]
    assert ml["precision"][0] == ["auto", "bf16", "fp16", "fp32"] and ml["precision"][1]["default"] == "bf16"
    assert ml["quantization"][0] == ["none", "fp8_e4m3fn", "fp8_e5m2", "auto"]
    assert m["HunyuanModelLoader"].RETURN_TYPES == ("HUNYUAN_MODEL",)
    assert m["HunyuanDependenciesLoader"].RETURN_TYPES == ("HUNYUAN_DEPS",)
    assert m["HunyuanFoleyTorchCompile"].RETURN_TYPES == ("TORCH_COMPILE_CFG",)
    assert m["HunyuanBlockSwap"].RETURN_TYPES == ("BLOCKSWAPARGS",)
    cfg, = m["HunyuanFoleyTorchCompile"]().make_config("inductor", "default", "None", False, 64)
    assert cfg["dynamic"] is None and cfg["dynamo_cache_limit"] == 64
    args, = m["HunyuanBlockSwap"]().set_args(blocks_to_swap=30, prefetch_blocks=1)
    assert args == {"blocks_to_swap": 30, "prefetch_blocks": 1}
    batch = {"waveform": torch.arange(6.0).view(3, 1, 2), "sample_rate": 48000}
    out, = m["SelectAudioFromBatch"]().select_audio(batch, 7)     # clamps with a warning
    assert torch.equal(out["waveform"], batch["waveform"][2:3]) and out["sample_rate"] == 48000


def test_example_workflow_loads_against_our_nodes():
    """Drop-in check: every Hunyuan node of the reference's example workflow exists
    here, its widget values validate against our INPUT_TYPES, and every link's socket type matches."""
    import json
    import foley_amd
    wf = json.load(open(os.path.join(ROOT, "tests", "golden", "example_workflow_nodes.json")))
    m = foley_amd.NODE_CLASS_MAPPINGS
    by_id = {n["id"]: n for n in wf["nodes"]}
    assert {n["type"] for n in wf["nodes"]} == {"HunyuanModelLoader", "HunyuanDependenciesLoader",
                                                "HunyuanFoleySampler", "HunyuanFoleyTorchCompile",
                                                "HunyuanBlockSwap", "SelectAudioFromBatch"}
    for n in wf["nodes"]:
        cls = m[n["type"]]
        it = cls.INPUT_TYPES()
        spec = {**it.get("required", {}), **it.get("optional", {})}
        for i in n["inputs"]:                       # every socket / converted widget the workflow names exists
            assert i["name"] in spec, (n["type"], i["name"])
        widgets = [k for k, v in spec.items() if isinstance(v[0], list) or v[0] in ("INT", "FLOAT", "STRING", "BOOLEAN")]
        vals = [v for v in (n["widgets_values"] or []) if v not in ("increment", "randomize", "fixed")]
        assert len(vals) == len(widgets), (n["type"], vals, widgets)
        for k, v in zip(widgets, vals):
            t, opts = spec[k][0], (spec[k][1] if len(spec[k]) > 1 else {})
            if isinstance(t, list):
                assert (not t) or v in t or k in ("model_name", "vae_name", "synchformer_name"), (k, v)
            elif t in ("INT", "FLOAT"):
                assert opts.get("min", v) <= v <= opts.get("max", v), (k, v)
    for l in wf["links"]:
        if l["from"] in by_id:
            src = m[by_id[l["from"]]["type"]]
            assert src.RETURN_TYPES[l["from_slot"]] == l["type"], l
        if l["to"] in by_id:
            dst = by_id[l["to"]]
            name = dst["inputs"][l["to_slot"]]["name"]
            it = m[dst["type"]].INPUT_TYPES()
            spec = {**it.get("required", {}), **it.get("optional", {})}
            assert spec[name][0] == l["type"], (l, spec[name][0])


def test_checkpoint_detection_helpers():
    from foley_amd import nodes
    sd = {"a": torch.zeros(4, 4, dtype=torch.bfloat16), "b": torch.zeros(2, dtype=torch.float32)}
    assert nodes.detect_ckpt_major_precision(sd) == torch.bfloat16 and nodes.detect_ckpt_fp8(sd) is None
    sd["c"] = torch.zeros(4, 4).to(torch.float8_e4m3fn)
    assert nodes.detect_ckpt_fp8(sd) == "fp8_e4m3fn"
    # quantization widget resolution (reference nodes.py:109-122): auto = checkpoint's flavour, else e4m3fn
    assert nodes.resolve_quantization("auto", None) == "fp8_e4m3fn" and nodes.resolve_quantization("auto", "fp8_e5m2") == "fp8_e5m2"
    assert nodes.resolve_quantization("none", "fp8_e5m2") == "none" and nodes.resolve_quantization("fp8_e5m2", None) == "fp8_e5m2"
    # rounding order: checkpoint -> parameter dtype -> fp8 (double rounding of fp32 checkpoints, re-rounding of the other flavour)
    w = torch.tensor([[1.0 + 2 ** -8 + 2 ** -12, 0.3]])
    r = nodes.fp8_round_state_dict({"a.weight": w}, "fp8_e4m3fn", param_dtype=torch.bfloat16)["a.weight"]
    assert torch.equal(r, w.to(torch.bfloat16).to(torch.float8_e4m3fn).float())
    r2 = nodes.fp8_round_state_dict({"a.weight": w.to(torch.float8_e5m2)}, "fp8_e4m3fn")["a.weight"]
    assert torch.equal(r2, w.to(torch.float8_e5m2).float().to(torch.float8_e4m3fn).float())
    f = nodes.select_frames(torch.rand(20, 4, 4, 3), 1.0, 16.0)
    assert f[0].shape == (8, 3, 4, 4) and f[1].shape == (25, 3, 4, 4) and f[0].dtype == torch.uint8


def test_text_padding_and_noise_draw():
    x = torch.randn(1, 12, 768)
    assert sampler.pad_or_trim_text(x, 77).shape == (1, 77, 768)
    assert float(sampler.pad_or_trim_text(x, 77)[:, 12:].abs().max()) == 0.0
    assert sampler.pad_or_trim_text(torch.randn(1, 90, 8), 77).shape == (1, 77, 8)
    g = golden("g6_c1_xxl")
    n = sampler.draw_noise(1, 128, 50, torch.float32, torch.Generator("cpu").manual_seed(1234))
    assert torch.equal(n, g["noise"])


def _mock_replicas(monkeypatch, n, behaviour):
    """N mocked replicas for `denoise_process_multi` on a box without a GPU: the stream / device context managers of
    torch.cuda are replaced by no-ops and `denoise_process_with_generator` by `behaviour(r, model, abort_event)`."""
    import contextlib
    import types

    class _Stream:
        def __init__(self, *_a, **_k): pass
        def wait_event(self, _ev): pass
        def synchronize(self): pass

    monkeypatch.setattr(torch.cuda, "Stream", _Stream)
    monkeypatch.setattr(torch.cuda, "device", lambda _d: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "stream", lambda _s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *_a: _Stream())
    cfg = types.SimpleNamespace(frame_rate=50, text_len=128, latent_dim=128)
    reps = []
    for r in range(n):
        ctx = types.SimpleNamespace(aborted=False)
        ctx.abort = (lambda c: (lambda: setattr(c, "aborted", True)))(ctx)
        reps.append((types.SimpleNamespace(cfg=cfg, dtype=torch.float32, device=torch.device("cpu"), _text_len_fixed=None, ctx=ctx, rank=r), None))

    def fake(visual, text, secs, model, dac, cfg_scale, steps, bs, solver, **kw):
        return behaviour(model.rank, model, kw["_abort_event"], kw.get("progress"), bs)
    monkeypatch.setattr(sampler, "denoise_process_with_generator", fake)
    return reps


@pytest.mark.parametrize("n", [2, 4])
def test_failing_replica_does_not_deadlock_the_others(monkeypatch, n):
    """a ComfyUI cancel - the progress callback of replica 0 raises - with other replicas inside their
    loops.  Every aborted replica raises too (foley_sample returns ABORTED -> FoleyRuntimeError); the first failing worker
    used to wait for `running[i]` of workers that were themselves spinning in their own `except` block: both waited forever.
    The call must return the FIRST error, promptly, and have asked every running replica to stop."""
    import threading
    import time

    class Cancel(Exception):
        pass

    inside = threading.Barrier(n)

    def behaviour(r, model, abort_event, progress, bs):
        inside.wait(timeout=5)                       # every replica is inside its loop before the cancel arrives
        if r == 0:
            progress(1, 10)                          # raises Cancel on the worker thread
        t0 = time.time()
        while not model.ctx.aborted:                 # the loop of the other replicas: stops when foley_abort() was called
            assert time.time() - t0 < 5, "replica was never asked to stop"
            time.sleep(0.001)
        raise rt.FoleyRuntimeError("foley_sample: aborted")

    reps = _mock_replicas(monkeypatch, n, behaviour)

    def cancel(_i, _n):
        raise Cancel()

    out = {}

    def call():
        try:
            sampler.denoise_process_multi({}, {"text_feat": torch.zeros(1, 12, 8)}, 1.0, reps, 4.5, 10, n, "euler",
                                          generator=torch.Generator("cpu").manual_seed(1), progress=cancel)
        except BaseException as e:                   # noqa: BLE001
            out["err"] = e

    t = threading.Thread(target=call, daemon=True)
    t.start()
    t.join(10)
    assert not t.is_alive(), "denoise_process_multi is stuck: failing workers wait on each other"
    assert isinstance(out.get("err"), Cancel), out                      # the first error is the one reported
    assert all(m.ctx.aborted for m, _ in reps[1:])


def test_replica_still_in_setup_does_not_start_after_a_failure(monkeypatch):
    """A replica that fails before the others have entered their loop: they see the shared event and never start."""
    started = []

    def behaviour(r, model, abort_event, progress, bs):
        started.append(r)
        raise rt.FoleyRuntimeError("boom")

    reps = _mock_replicas(monkeypatch, 3, behaviour)
    with pytest.raises(rt.FoleyRuntimeError, match="boom"):
        sampler.denoise_process_multi({}, {"text_feat": torch.zeros(1, 12, 8)}, 1.0, reps, 4.5, 10, 3, "euler",
                                      generator=torch.Generator("cpu").manual_seed(1))
    assert 1 <= len(started) <= 3


def test_checkpoint_file_formats(tmp_path):
    """On-disk formats either side of the loader boundary: .safetensors (incl. fp8
    tensors), .pth flat dict, .pth {"state_dict": ...} (reference utils.py:49-59, nodes.py:86)."""
    from safetensors.torch import save_file
    from foley_amd import nodes
    sd = synth.synth_dit_state_dict(C.TINY)
    keys = sorted(sd)[:40]
    small = {k: sd[k].contiguous() for k in keys}
    small8 = {k: (v.to(torch.float8_e4m3fn) if nodes.fp8_wrapped_key(k, v) else v) for k, v in small.items()}
    p1, p2, p3, p4 = (str(tmp_path / n) for n in ("a.safetensors", "b.pth", "c.pth", "d.safetensors"))
    save_file(small, p1)
    torch.save(small, p2)
    torch.save({"state_dict": small, "metadata": {"x": 1}}, p3)
    save_file(small8, p4)
    for p in (p1, p2, p3):
        got = nodes._load_state_dict(p)
        assert sorted(got) == keys and all(torch.equal(got[k], small[k]) for k in keys)
    got8 = nodes._load_state_dict(p4)
    assert nodes.detect_ckpt_fp8(got8) == "fp8_e4m3fn"
    # an fp8 checkpoint is honoured verbatim by quantization="auto": same values as rounding the fp32 one
    a = nodes.fp8_round_state_dict(got8, nodes.resolve_quantization("auto", nodes.detect_ckpt_fp8(got8)))
    b = nodes.fp8_round_state_dict(small, "fp8_e4m3fn")
    assert all(torch.equal(a[k], b[k]) for k in keys)


@pytest.mark.skipif(os.environ.get("FOLEY_SLOW") != "1", reason="set FOLEY_SLOW=1 to run (recompiles every kernel file, ~3 min)")
def test_no_kernel_keeps_accumulators_in_scratch():
    """tools/kernel_resources.py: no kernel instantiation may use more than a handful of scratch bytes per lane
    (round 2 found a GEMM tile whose accumulators lived in scratch - 576 bytes per lane - and, later, a
    150-register spill introduced by an unrelated one-line change)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True,
                         timeout=1800).stdout
    worst = 0
    for line in out.splitlines():
        if " scratch " in line:
            worst = max(worst, int(line.split(" scratch ")[1].split()[0]))
    assert worst <= 32, out
