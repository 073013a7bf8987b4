"""CPU oracle for the HunyuanVideo-Foley sampling path.   *** TEST INFRASTRUCTURE ***

A plain, functional fp32 restatement (torch CPU tensor ops: matmul / conv1d /
softmax / sin) of the reference algorithm for the hot path named by
BASELINE.json: Euler/CFG flow-match loop -> Foley DiT forward -> DAC-VAE decode.
It consumes a *reference-keyed* state dict, so it can be
driven by the same tensors as the reference classes.

Who may use this file: `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` - as the checker / reported baseline only.
The product (package `comfyui-hunyuanvideo-foley_amd`) never imports it and
has no CPU fallback; it fails loudly if the HIP library is missing.

Parity pin: the reference has no tests or golden vectors of its own,
so this oracle is pinned against the reference *itself*, imported in the build
container through `tests/golden/ref_harness.py`:
`tests/golden/make_golden.py` checks every function here against the
reference's modules (<=1e-5 relative) and freezes reference outputs into
`tests/golden/*.npz`; `tests/test_oracle_golden.py` re-checks the oracle
against those frozen reference outputs on any box.

Every function cites the reference lines (relative to /root/reference) it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# =============================================================================
# Scheduler  (hunyuanvideo_foley/utils/schedulers/scheduling_flow_match_discrete.py)
# =============================================================================
def flow_sigmas(num_steps: int, shift: float = 1.0) -> Tensor:
    """sigma_i = linspace(1, 0, N+1); SD3 shift when shift != 1 (:143-152, :207-208)."""
    s = torch.linspace(1, 0, num_steps + 1)
    if shift != 1.0:
        s = (shift * s) / (1 + (shift - 1) * s)
    return s


def flow_timesteps(sigmas: Tensor) -> Tensor:
    """Model-facing t_i = 1000 * sigma_i, fp32 (:157)."""
    return (sigmas[:-1] * 1000).to(torch.float32)


class SolverState:
    """State machine of `FlowMatchDiscreteScheduler.step` for all four solvers (:210-373).

    Faithful to the reference including its quirk that for multi-stage solvers
    every loop iteration is one *stage* (the caller loops over `timesteps`, so an
    N-iteration heun-2 run takes N/2 real steps and evaluates the model at the
    loop's t_i rather than the stage time).
    """

    def __init__(self, sigmas: Tensor, solver: str = "euler"):
        assert solver in ("euler", "heun-2", "midpoint-2", "kutta-4")
        self.sigmas, self.solver = sigmas, solver
        self.idx = 0
        self.d1 = self.d2 = self.d3 = None
        self.dt = None
        self.sample = None

    def step(self, v: Tensor, x: Tensor) -> Tensor:
        x = x.float()
        v = v.float()
        sigma, sigma_next = self.sigmas[self.idx], self.sigmas[self.idx + 1]
        last = True
        if self.solver == "euler":                      # :299-302
            deriv, dt = v, sigma_next - sigma
        elif self.solver in ("heun-2", "midpoint-2"):   # :304-341
            if self.d1 is None:
                self.d1, self.dt, self.sample = v, sigma_next - sigma, x
                deriv = v
                dt = self.dt if self.solver == "heun-2" else self.dt / 2
                last = False
            else:
                deriv = 0.5 * (self.d1 + v) if self.solver == "heun-2" else v
                dt, x = self.dt, self.sample
                self.d1 = self.dt = self.sample = None
        else:                                           # kutta-4 :343-384
            if self.d1 is None:
                self.d1, self.dt, self.sample = v, sigma_next - sigma, x
                deriv, dt, last = v, self.dt / 2, False
            elif self.d2 is None:
                self.d2 = v
                deriv, dt, last = v, self.dt / 2, False
            elif self.d3 is None:
                self.d3 = v
                deriv, dt, last = v, self.dt, False
            else:
                deriv = 1 / 6 * self.d1 + 1 / 3 * self.d2 + 1 / 3 * self.d3 + 1 / 6 * v
                dt, x = self.dt, self.sample
                self.d1 = self.d2 = self.d3 = self.dt = self.sample = None
        out = x + deriv * dt                            # :280
        if last:
            self.idx += 1
        return out


# =============================================================================
# RoPE  (models/nn/posemb_layers.py, models/nn/attn_layers.py)
# =============================================================================
def rope_table(positions: Tensor, dim: int = 128, theta: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """cos/sin [P, dim], each frequency repeated twice (posemb_layers.py:117-172).

    freq_k = theta^(-2k/dim) (Tensor-Tensor pow in fp32), angle = pos * freq.
    """
    pos = positions.to(torch.float32)
    idx = torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2]
    freqs = torch.pow(torch.tensor(theta, dtype=torch.float32).expand_as(idx),
                      -(idx / torch.tensor(float(dim))))
    ang = torch.outer(pos, freqs)
    return ang.cos().repeat_interleave(2, dim=1), ang.sin().repeat_interleave(2, dim=1)


def rope_positions(n: int) -> Tensor:
    """get_meshgrid_nd: linspace(0, n, n+1)[:n] (posemb_layers.py:47-55)."""
    return torch.linspace(0.0, float(n), n + 1, dtype=torch.float32)[:n]


def nearest_exact_index(out_len: int, in_len: int) -> Tensor:
    """Source index of F.interpolate(x, size=out_len, mode='nearest-exact') as the reference calls it
    (hifi_foley.py:44,58,761): obtained from the op itself on an index ramp, so ATen's float32
    evaluation of floor((i+0.5)*in/out) is reproduced exactly (a float64 formula differs for ~5 % of
    the durations the sampler widget allows)."""
    src = torch.arange(in_len, dtype=torch.float32).view(1, 1, in_len)
    return F.interpolate(src, size=out_len, mode="nearest-exact").view(-1).long()


def interleaved_positions(la: int, lv: int) -> Tuple[Tensor, Tensor]:
    """Positions seen by audio / visual tokens under the interleaved RoPE.

    interleave_two_sequences + apply_rotary_emb + decouple_interleaved_two_sequences
    (hifi_foley.py:35-60, 236-251): the visual sequence is nearest-exact up-sampled
    to `la`, interleaved [a0 v0 a1 v1 ...], rotated with positions 0..2la-1, split,
    and the visual half nearest-exact down-sampled back to `lv`.  Net effect:
    audio token i is rotated with position 2i, visual token j with position
    2*src(j)+1 where src(j) = floor((j+0.5)*la/lv) is the up-sampled slot that
    the down-sampling picks - and that slot holds v_j itself iff up(src(j)) == j
    (true for all (la, lv) the sampler produces; asserted here).
    """
    up = nearest_exact_index(la, lv)      # slot -> visual token
    down = nearest_exact_index(lv, la)    # visual token -> slot
    assert torch.equal(up[down], torch.arange(lv)), "interleaved RoPE is not a pure re-indexing"
    return 2 * torch.arange(la), 2 * down + 1


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """x [..., S, H, 128] with cos/sin [S, 128] (attn_layers.py:112-146, head_first=False).

    out = x*cos + rotate_half(x)*sin, rotate_half: (x0,x1) -> (-x1, x0) per pair.
    """
    xf = x.float()
    x0, x1 = xf[..., 0::2], xf[..., 1::2]
    rot = torch.stack((-x1, x0), dim=-1).flatten(-2)
    return xf * cos[:, None, :] + rot * sin[:, None, :]


# =============================================================================
# Small layers  (models/nn/*.py)
# =============================================================================
def rms_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """norm_layers.py:36-52 / nn.RMSNorm: x * rsqrt(mean(x^2) + eps) * w."""
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def layer_norm(x: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def timestep_embedding(t: Tensor, dim: int = 256, max_period: int = 10000) -> Tensor:
    """embed_layers.py:76-101: [cos(t f) | sin(t f)], f_i = exp(-ln(P) i / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def time_embed_fp8_autocast(sd: Dict[str, Tensor], t: Tensor, qdtype: torch.dtype) -> Tensor:
    """TimestepEmbedder when its Linears are fp8-wrapped and the model runs under bf16 autocast
    (the only way the reference runs fp8 checkpoints, utils.py:229-234): embed_layers.py:134 casts
    the sinusoid features to `mlp[0].weight.dtype` = fp8, FP8WeightWrapper.forward (utils.py:361-366)
    then keeps the weight in fp8 and casts the first BIAS to fp8 as well; autocast runs both Linears
    in bf16.  The second layer sees a bf16 activation and is an ordinary wrapped Linear."""
    r8 = lambda v: v.to(qdtype).to(torch.bfloat16)
    f = r8(timestep_embedding(t, sd["time_in.mlp.0.weight"].shape[1]))
    h = F.linear(f, r8(sd["time_in.mlp.0.weight"]), r8(sd["time_in.mlp.0.bias"]))
    h = F.silu(h)
    y = F.linear(h, r8(sd["time_in.mlp.2.weight"]), sd["time_in.mlp.2.bias"].to(torch.bfloat16))
    return y.float()


def conv1d_cl(x: Tensor, w: Tensor, b: Optional[Tensor], pad: int) -> Tensor:
    """ChannelLastConv1d (mlp_layers.py:104-110): x [B, L, C] -> [B, L, C_out]."""
    return F.conv1d(x.transpose(1, 2), w, b, padding=pad).transpose(1, 2)


def sdpa(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """softmax(q k^T / sqrt(d)) v with q,k,v [B, H, S, d]; no mask (attn_layers.py:419-422)."""
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    return torch.matmul(torch.softmax(s, dim=-1), v)


# =============================================================================
# DiT blocks  (models/hifi_foley.py)
# =============================================================================
def _heads(x: Tensor, n: int, H: int) -> List[Tensor]:
    """'B L (K H D) -> K B L H D' (hifi_foley.py:219)."""
    B, L, _ = x.shape
    return list(x.view(B, L, n, H, -1).unbind(2))


def triple_block(sd: SD, p: str, H: int, audio: Tensor, cond: Tensor, v_cond: Tensor, vec: Tensor,
                 rope_a: Tuple[Tensor, Tensor], rope_v: Tuple[Tensor, Tensor],
                 rope_aq: Tuple[Tensor, Tensor], rope_vq: Tuple[Tensor, Tensor],
                 rope_t: Tuple[Tensor, Tensor]) -> Tuple[Tensor, Tensor]:
    """TwoStreamCABlock.forward (hifi_foley.py:179-333); cond passes through unchanged."""
    W = lambda k: sd[p + k]
    La, Lv = audio.shape[1], v_cond.shape[1]
    sv = F.silu(vec)
    am = F.linear(sv, W("audio_mod.linear.weight"), W("audio_mod.linear.bias")).chunk(9, dim=-1)
    vm = F.linear(sv, W("v_cond_mod.linear.weight"), W("v_cond_mod.linear.bias")).chunk(9, dim=-1)
    mod = lambda x, sh, sc: x * (1 + sc[:, None]) + sh[:, None]     # modulate_layers.py:19-30

    # 1. joint self attention over [v_cond ; audio]                         (:215-269)
    aq, ak, av = _heads(F.linear(mod(layer_norm(audio, 1e-6), am[0], am[1]),
                                 W("audio_self_attn_qkv.weight"), W("audio_self_attn_qkv.bias")), 3, H)
    aq = rms_norm(aq, W("audio_self_q_norm.weight"), 1e-6)
    ak = rms_norm(ak, W("audio_self_k_norm.weight"), 1e-6)
    vq, vk, vv = _heads(F.linear(mod(layer_norm(v_cond, 1e-6), vm[0], vm[1]),
                                 W("v_cond_attn_qkv.weight"), W("v_cond_attn_qkv.bias")), 3, H)
    vq = rms_norm(vq, W("v_cond_attn_q_norm.weight"), 1e-6)
    vk = rms_norm(vk, W("v_cond_attn_k_norm.weight"), 1e-6)
    aq, ak = apply_rope(aq, *rope_a), apply_rope(ak, *rope_a)
    vq, vk = apply_rope(vq, *rope_v), apply_rope(vk, *rope_v)
    q = torch.cat((vq, aq), dim=1).transpose(1, 2)
    k = torch.cat((vk, ak), dim=1).transpose(1, 2)
    v = torch.cat((vv, av), dim=1).transpose(1, 2)
    att = sdpa(q, k, v).transpose(1, 2).flatten(2)
    v_att, a_att = att[:, :Lv], att[:, Lv:]
    audio = audio + F.linear(a_att, W("audio_self_proj.weight"), W("audio_self_proj.bias")) * am[2][:, None]
    v_cond = v_cond + F.linear(v_att, W("v_cond_self_proj.weight"), W("v_cond_self_proj.bias")) * vm[2][:, None]

    # 2. cross attention, query = [v_cond ; audio], key/value = text            (:271-319)
    aq = _heads(F.linear(mod(layer_norm(audio, 1e-6), am[3], am[4]),
                         W("audio_cross_q.weight"), W("audio_cross_q.bias")), 1, H)[0]
    aq = apply_rope(rms_norm(aq, W("audio_cross_q_norm.weight"), 1e-6), *rope_aq)
    vq = _heads(F.linear(mod(layer_norm(v_cond, 1e-6), vm[3], vm[4]),
                         W("v_cond_cross_q.weight"), W("v_cond_cross_q.bias")), 1, H)[0]
    vq = apply_rope(rms_norm(vq, W("v_cond_cross_q_norm.weight"), 1e-6), *rope_vq)
    tk, tv = _heads(F.linear(cond, W("text_cross_kv.weight"), W("text_cross_kv.bias")), 2, H)
    tk = apply_rope(rms_norm(tk, W("text_cross_k_norm.weight"), 1e-6), *rope_t)
    q = torch.cat((vq, aq), dim=1).transpose(1, 2)
    att = sdpa(q, tk.transpose(1, 2), tv.transpose(1, 2)).transpose(1, 2).flatten(2)
    v_att, a_att = att[:, :Lv], att[:, Lv:]
    audio = audio + F.linear(a_att, W("audio_cross_proj.weight"), W("audio_cross_proj.bias")) * am[5][:, None]
    v_cond = v_cond + F.linear(v_att, W("v_cond_cross_proj.weight"), W("v_cond_cross_proj.bias")) * vm[5][:, None]

    # 3. GELU-tanh MLPs                                                      (:321-331)
    def mlp(x, pre):
        h = F.gelu(F.linear(x, W(pre + ".fc1.weight"), W(pre + ".fc1.bias")), approximate="tanh")
        return F.linear(h, W(pre + ".fc2.weight"), W(pre + ".fc2.bias"))
    audio = audio + mlp(mod(layer_norm(audio, 1e-6), am[6], am[7]), "audio_mlp") * am[8][:, None]
    v_cond = v_cond + mlp(mod(layer_norm(v_cond, 1e-6), vm[6], vm[7]), "v_cond_mlp") * vm[8][:, None]
    return audio, v_cond


def single_block(sd: SD, p: str, H: int, x: Tensor, cond: Tensor, rope: Tuple[Tensor, Tensor]) -> Tensor:
    """SingleStreamBlock.forward (hifi_foley.py:364-390); cond is per-token [B, L, D]."""
    W = lambda k: sd[p + k]
    B, L, D = x.shape
    m = F.linear(F.silu(cond), W("modulation.linear.weight"), W("modulation.linear.bias")).chunk(6, dim=-1)
    xn = layer_norm(x, 1e-5) * (1 + m[1]) + m[0]
    qkv = F.linear(xn, W("linear_qkv.weight"), W("linear_qkv.bias"))
    qkv = qkv.view(B, L, H, D // H, 3)                 # 'B L (H D K)' (:362)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]   # [B, L, H, d]
    eps = torch.finfo(torch.float32).eps               # nn.RMSNorm(eps=None) (:360-361)
    q = apply_rope(rms_norm(q, W("q_norm.weight"), eps), *rope)
    k = apply_rope(rms_norm(k, W("k_norm.weight"), eps), *rope)
    out = sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).flatten(2)
    x = x + conv1d_cl(out, W("linear1.weight"), W("linear1.bias"), 1) * m[2]
    xn = layer_norm(x, 1e-5) * (1 + m[4]) + m[3]
    h = F.silu(conv1d_cl(xn, W("linear2.w1.weight"), None, 1)) * conv1d_cl(xn, W("linear2.w3.weight"), None, 1)
    return x + conv1d_cl(h, W("linear2.w2.weight"), None, 1) * m[5]


def dit_forward(sd: SD, heads: int, x: Tensor, t: Tensor, cond: Tensor, clip_feat: Tensor,
                sync_feat: Tensor, n_triple: Optional[int] = None, n_single: Optional[int] = None,
                taps: Optional[dict] = None, fp8_time: Optional[torch.dtype] = None) -> Tensor:
    """HunyuanVideoFoley.forward (hifi_foley.py:707-924) for the shipped configuration:
    add_sync_feat_to_audio, interleaved_audio_visual_rope, no attention mask.

    x [B,128,La], t [B], cond [B,Lt,768], clip_feat [B,Lv,768], sync_feat [B,Ls,768] -> [B,128,La]
    """
    H = heads
    B, _, La = x.shape
    if n_triple is None:
        n_triple = 1 + max((int(k.split(".")[1]) for k in sd if k.startswith("triple_blocks.")), default=-1)
    if n_single is None:
        n_single = 1 + max((int(k.split(".")[1]) for k in sd if k.startswith("single_blocks.")), default=-1)
    D = sd["time_in.mlp.2.weight"].shape[0]
    hd = D // H
    # time embedding (:744, embed_layers.py:104-136)
    # fp8_time: fp8-wrapped model under autocast - the features and the first bias pass through fp8
    # (see time_embed_fp8_autocast; `sd` is expected to hold the fp8-rounded weights already)
    tf = timestep_embedding(t, sd["time_in.mlp.0.weight"].shape[1])
    b0 = sd["time_in.mlp.0.bias"]
    if fp8_time is not None:
        tf, b0 = tf.to(fp8_time).to(tf.dtype), b0.to(fp8_time).to(b0.dtype)
    vec = F.linear(F.silu(F.linear(tf, sd["time_in.mlp.0.weight"], b0)),
                   sd["time_in.mlp.2.weight"], sd["time_in.mlp.2.bias"])
    # sync features (:755-762)
    Ls = sync_feat.shape[1]
    assert Ls % 8 == 0
    sf = (sync_feat.view(B, Ls // 8, 8, -1) + sd["sync_pos_emb"]).view(B, Ls, -1)
    sf = F.silu(F.linear(sf, sd["sync_in.0.weight"], sd["sync_in.0.bias"]))
    sf = conv1d_cl(F.silu(conv1d_cl(sf, sd["sync_in.2.w1.weight"], None, 0))
                   * conv1d_cl(sf, sd["sync_in.2.w3.weight"], None, 0), sd["sync_in.2.w2.weight"], None, 0)
    add_sync = sf[:, nearest_exact_index(La, Ls)]
    # text / audio / visual embedders (:765-770)
    cond = F.linear(F.silu(F.linear(cond, sd["cond_in.linear_1.weight"], sd["cond_in.linear_1.bias"])),
                    sd["cond_in.linear_2.weight"], sd["cond_in.linear_2.bias"])
    audio = F.conv1d(x, sd["audio_embedder.proj.weight"], sd["audio_embedder.proj.bias"]).transpose(1, 2)
    v_cond = F.linear(F.silu(F.linear(clip_feat, sd["visual_proj.w1.weight"]))
                      * F.linear(clip_feat, sd["visual_proj.w3.weight"]), sd["visual_proj.w2.weight"])
    Lv, Lt = v_cond.shape[1], cond.shape[1]
    # RoPE tables (:797-799, :295-308, :865)
    pa, pv = interleaved_positions(La, Lv)
    full = rope_table(rope_positions(2 * La), hd)
    rope_a = (full[0][pa], full[1][pa])
    rope_v = (full[0][pv], full[1][pv])
    rope_aq = rope_table(rope_positions(La), hd)
    rope_vq = rope_table(rope_positions(Lv), hd)
    rope_t = rope_table(rope_positions(Lt), hd)
    audio = audio + add_sync                                       # layer 0 (:838-839)
    if taps is not None:
        taps["audio_in"], taps["v_cond_in"], taps["cond_in"], taps["vec"] = audio, v_cond, cond, vec
    for b in range(n_triple):
        audio, v_cond = triple_block(sd, f"triple_blocks.{b}.", H, audio, cond, v_cond, vec,
                                     rope_a, rope_v, rope_aq, rope_vq, rope_t)
        if taps is not None:
            taps[f"triple{b}"] = audio
    vec3 = add_sync + vec[:, None]                                 # (:866-867)
    xs = audio
    for b in range(n_single):
        xs = single_block(sd, f"single_blocks.{b}.", H, xs, vec3, rope_aq)
        if taps is not None:
            taps[f"single{b}"] = xs
    # FinalLayer1D with 3-D conditioning: the adaLN shift/scale are dropped by
    # modulate() (modulate_layers.py:20-24) => linear(LayerNorm(x)).
    out = F.linear(layer_norm(xs, 1e-6), sd["final_layer.linear.weight"], sd["final_layer.linear.bias"])
    return out.transpose(1, 2)                                     # unpatchify1d (:926-936)


# =============================================================================
# DAC-VAE decoder  (models/dac_vae/model/dac.py, nn/layers.py)
# =============================================================================
def weight_norm_fold(g: Tensor, v: Tensor) -> Tensor:
    """torch weight_norm parametrization, dim=0: w = g * v / ||v|| over all dims but 0
    (nn/layers.py:9-14).  For ConvTranspose1d dim 0 is the *input* channel."""
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def _wn_weight(sd: SD, key: str) -> Tensor:
    if key + ".weight" in sd:                       # already folded / plain
        return sd[key + ".weight"]
    if key + ".weight_g" in sd:                     # legacy spelling
        return weight_norm_fold(sd[key + ".weight_g"], sd[key + ".weight_v"])
    return weight_norm_fold(sd[key + ".parametrizations.weight.original0"],
                            sd[key + ".parametrizations.weight.original1"])


def snake(x: Tensor, alpha: Tensor) -> Tensor:
    """x + (alpha + 1e-9)^-1 * sin(alpha x)^2, alpha [1, C, 1] (nn/layers.py:18-24)."""
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def dac_decode(sd: SD, z: Tensor, rates: Sequence[int] = (8, 5, 4, 3, 2),
               dilations: Sequence[int] = (1, 3, 9), taps: Optional[dict] = None) -> Tensor:
    """DAC.decode, continuous=True (dac.py:280-303): z [B,128,T] -> audio [B,1,T*hop]."""
    x = F.conv1d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])          # :297
    x = F.conv1d(x, _wn_weight(sd, "decoder.model.0"), sd["decoder.model.0.bias"], padding=3)
    for i, s in enumerate(rates):                                                       # :98-117
        p = f"decoder.model.{i + 1}.block."
        x = snake(x, sd[p + "0.alpha"])
        x = F.conv_transpose1d(x, _wn_weight(sd, p + "1"), sd[p + "1.bias"], stride=s,
                               padding=math.ceil(s / 2), output_padding=s % 2)
        for j, d in enumerate(dilations):                                               # :28-44
            q = p + f"{j + 2}.block."
            y = snake(x, sd[q + "0.alpha"])
            y = F.conv1d(y, _wn_weight(sd, q + "1"), sd[q + "1.bias"], dilation=d, padding=3 * d)
            y = snake(y, sd[q + "2.alpha"])
            y = F.conv1d(y, _wn_weight(sd, q + "3"), sd[q + "3.bias"])
            x = x + y
        if taps is not None:
            taps[f"stage{i}"] = x
    n = len(rates)
    x = snake(x, sd[f"decoder.model.{n + 1}.alpha"])
    x = F.conv1d(x, _wn_weight(sd, f"decoder.model.{n + 2}"), sd[f"decoder.model.{n + 2}.bias"], padding=3)
    return torch.tanh(x)


def dac_preprocess(audio: Tensor, hop: int) -> Tensor:
    """DAC.preprocess (dac.py:225-234): right-pad the waveform to a multiple of the hop."""
    T = audio.shape[-1]
    return F.pad(audio, (0, math.ceil(T / hop) * hop - T))


def dac_encode(sd: SD, audio: Tensor, rates: Sequence[int] = (2, 3, 4, 5, 8),
               dilations: Sequence[int] = (1, 3, 9)) -> Tensor:
    """DAC.encode, continuous=True (dac.py:236-278): audio [B,1,T] (T a multiple of the hop) ->
    posterior parameters [B, 2*latent, T/hop] = quant_conv(encoder(audio)); rows [:latent] are the
    mean, rows [latent:] the log-variance of the DiagonalGaussianDistribution (vae_utils.py:24-31).
    Encoder (dac.py:47-95): conv7 -> per rate s {3 residual units (dil 1/3/9) -> snake -> conv k=2s,
    stride s, pad ceil(s/2), doubling the channels} -> snake -> conv3 to the latent width."""
    x = F.conv1d(audio, _wn_weight(sd, "encoder.block.0"), sd["encoder.block.0.bias"], padding=3)
    for i, s in enumerate(rates):
        p = f"encoder.block.{i + 1}.block."
        for j, d in enumerate(dilations):                                               # dac.py:28-44
            q = p + f"{j}.block."
            y = snake(x, sd[q + "0.alpha"])
            y = F.conv1d(y, _wn_weight(sd, q + "1"), sd[q + "1.bias"], dilation=d, padding=3 * d)
            y = snake(y, sd[q + "2.alpha"])
            y = F.conv1d(y, _wn_weight(sd, q + "3"), sd[q + "3.bias"])
            x = x + y
        x = snake(x, sd[p + "3.alpha"])
        x = F.conv1d(x, _wn_weight(sd, p + "4"), sd[p + "4.bias"], stride=s, padding=math.ceil(s / 2))
    n = len(rates)
    x = snake(x, sd[f"encoder.block.{n + 1}.alpha"])
    x = F.conv1d(x, _wn_weight(sd, f"encoder.block.{n + 2}"), sd[f"encoder.block.{n + 2}.bias"], padding=1)
    return F.conv1d(x, sd["quant_conv.weight"], sd["quant_conv.bias"])                  # :274


def gaussian_posterior(params: Tensor):
    """DiagonalGaussianDistribution (vae_utils.py:24-31): (mean, std) with logvar clamped to [-30, 20];
    mode() = mean, sample() = mean + std * N(0, 1)."""
    mean, logvar = torch.chunk(params, 2, dim=1)
    return mean, torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))


# =============================================================================
# Sampler  (/utils.py:125-258)
# =============================================================================
def pad_or_trim_text(x: Tensor, t_fixed: int = 77) -> Tensor:
    """_pad_or_trim_time (utils.py:103-111): zero-pad / trim the token axis to 77."""
    T = x.shape[1]
    if T == t_fixed:
        return x
    if T > t_fixed:
        return x[:, :t_fixed]
    return F.pad(x, (0, 0, 0, t_fixed - T))


def sample_latents(sd: SD, heads: int, noise: Tensor, text: Tensor, uncond_text: Tensor, clip: Tensor,
                   sync: Tensor, steps: int, guidance: float, solver: str = "euler",
                   shift: float = 1.0, text_len: int = 77, trace: Optional[list] = None,
                   max_iters: Optional[int] = None, start_iter: int = 0) -> Tensor:
    """The denoising loop of denoise_process_with_generator (utils.py:144-247).
    `start_iter` > 0 (Euler only) resumes the loop from latents `noise` = the state after that many iterations.

    noise [bs,128,La] (already drawn, utils.py:151-156); text/uncond_text [1,T,768];
    clip [1,Lv,768]; sync [1,Ls,768].  CFG batch order is [uncond ; cond]
    (utils.py:193-195); CFG is skipped iff guidance <= 1.0.
    """
    bs = noise.shape[0]
    sig = flow_sigmas(steps, shift)
    ts = flow_timesteps(sig)
    st = SolverState(sig, solver)
    rep = lambda a: a.repeat(bs, 1, 1)
    text_r = pad_or_trim_text(rep(text), text_len)
    unc_r = pad_or_trim_text(rep(uncond_text), text_len)
    clip_r, sync_r = rep(clip), rep(sync)
    if guidance > 1.0:
        e_clip = sd["empty_clip_feat"].unsqueeze(0).expand(bs, clip.shape[1], -1)   # hifi_foley.py:620-632
        e_sync = sd["empty_sync_feat"].unsqueeze(0).expand(bs, sync.shape[1], -1)
        clip_in, sync_in = torch.cat([e_clip, clip_r]), torch.cat([e_sync, sync_r])
        text_in = torch.cat([unc_r, text_r])
    else:
        clip_in, sync_in, text_in = clip_r, sync_r, text_r
    x = noise.float()
    assert start_iter == 0 or solver == "euler"
    st.idx = start_iter
    for i, t in enumerate(ts):
        if i < start_iter:
            continue
        if max_iters is not None and i >= max_iters:
            break
        xin = torch.cat([x, x]) if guidance > 1.0 else x
        v = dit_forward(sd, heads, xin, t.expand(xin.shape[0]), text_in, clip_in, sync_in)
        if guidance > 1.0:
            vu, vc = v.chunk(2)
            v = vu + guidance * (vc - vu)
        x = st.step(v, x)
        if trace is not None:
            trace.append(x.clone())
    return x


def sample_waveform(dit_sd: SD, dac_sd: SD, heads: int, noise: Tensor, cond: Dict[str, Tensor], steps: int,
                    guidance: float, solver: str = "euler", rates: Sequence[int] = (8, 5, 4, 3, 2),
                    trace: Optional[list] = None) -> Tensor:
    """Full path: loop + DAC decode; returns audio [bs,1,La*hop] fp32 (utils.py:249-258).
    (The reference's 'trim to exact length' slices the size-1 channel axis: a no-op)"""
    lat = sample_latents(dit_sd, heads, noise, cond["text"], cond["uncond_text"], cond["clip"],
                         cond["sync"], steps, guidance, solver, trace=trace)
    return dac_decode(dac_sd, lat.float(), rates)


# ----------------------------------------------------------------------------- frame pre-processing (V2A conditioning)
def _keys_cubic(x: float, a: float = -0.5) -> float:
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


def resize_u8_axis(x, axis: int, n_out: int):
    """One axis of torchvision v2.Resize(interpolation=BICUBIC, antialias=True) on a uint8 array, the way the reference's frames
    go through it on the CPU (nodes.py:184-196 builds the two pipelines, utils.py:262-283 runs them before `.to(device)`): for
    uint8 CPU tensors v2.Resize dispatches to ATen's native uint8 kernel - PIL's scheme: per output sample a window of
    `2*2*max(scale,1)` input samples around centre = scale*(i+0.5), Keys cubic (a = -0.5) weights normalised in double
    precision, rounded half away from zero to int16 at the largest precision whose biggest weight stays below 2^15, an int32
    accumulator preset to half an output step, arithmetic shift, saturation.  Plain loops over numpy rows.
    Pinned against that very kernel (F.interpolate on uint8) in tests/test_oracle_golden.py - torchvision is not in the image."""
    import numpy as np
    x = np.moveaxis(np.asarray(x), axis, -1)
    n_in = x.shape[-1]
    if n_in == n_out:
        return np.moveaxis(x, -1, axis)
    scale = n_in / n_out
    support = 2.0 * scale if scale >= 1.0 else 2.0
    inv = 1.0 / scale if scale >= 1.0 else 1.0
    kmax = int(math.ceil(support)) * 2 + 1
    rows = []
    wmax = 0.0
    for i in range(n_out):
        centre = scale * (i + 0.5)
        lo = max(int(centre - support + 0.5), 0)
        n = min(max(min(int(centre + support + 0.5), n_in) - lo, 0), kmax)
        w = [_keys_cubic((j + lo - centre + 0.5) * inv) for j in range(n)]
        tot = 0.0
        for v in w:
            tot += v
        if tot != 0.0:
            w = [v / tot for v in w]
        wmax = max([wmax] + w)
        rows.append((lo, w))
    prec = 0
    while prec < 22 and int(0.5 + wmax * (1 << (prec + 1))) < (1 << 15):
        prec += 1
    out = np.zeros(x.shape[:-1] + (n_out,), dtype=np.uint8)
    for i, (lo, w) in enumerate(rows):
        acc = np.full(x.shape[:-1], 1 << (prec - 1), dtype=np.int64)
        for j, v in enumerate(w):
            s = v * (1 << prec)
            wi = int(s - 0.5) if s < 0 else int(s + 0.5)
            acc += x[..., lo + j].astype(np.int64) * wi
        out[..., i] = np.clip(acc >> prec, 0, 255).astype(np.uint8)
    return np.moveaxis(out, -1, axis)


def resize_u8(frames, size: Tuple[int, int]):
    """uint8 [..., H, W] -> [..., size[0], size[1]]: the horizontal pass first, its uint8 image through the vertical pass."""
    return resize_u8_axis(resize_u8_axis(frames, -1, size[1]), -2, size[0])
