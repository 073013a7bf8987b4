"""Host-built schedule / index / trig tables for one sampling run (tiny, built once with torch CPU).

These are the step-invariant lookups the reference rebuilds inside its hot loop:
  * flow-match sigma grid + model timesteps (scheduling_flow_match_discrete.py:131-170),
  * per-iteration solver coefficients - the `step()` state machine for euler / heun-2 /
    midpoint-2 / kutta-4 (:262-373) flattened into a table the device kernel indexes,
  * sinusoidal timestep features (embed_layers.py:76-101),
  * RoPE cos/sin table (posemb_layers.py:117-172; one row per position, one column per pair),
  * token -> RoPE position maps incl. the interleaved audio/visual scheme (hifi_foley.py:35-60,
    236-251), and the nearest-exact up-sampling index of the sync features (hifi_foley.py:761).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

SOLVERS = ("euler", "heun-2", "midpoint-2", "kutta-4")
STEP_SAVE_X, STEP_USE_SAVED, STEP_ACC_RESET = 1, 2, 4


def sigma_grid(steps: int, shift: float = 1.0) -> torch.Tensor:
    s = torch.linspace(1, 0, steps + 1)
    if shift != 1.0:
        s = (shift * s) / (1 + (shift - 1) * s)
    return s


def model_timesteps(sigmas: torch.Tensor) -> torch.Tensor:
    return (sigmas[:-1] * 1000).to(torch.float32)


def solver_table(sigmas: torch.Tensor, solver: str, n_iter: int) -> torch.Tensor:
    """[n_iter, 8] rows {w_new, w_acc, dt, w_store, flags, 0, 0, 0}.

    Per iteration the device computes  deriv = w_new*v + w_acc*acc ; x' = base + deriv*dt ;
    acc' = (reset ? 0 : acc) + w_store*v, with base = saved sample or current sample.
    Like the reference, every loop iteration is one solver *stage* and the sigma index only
    advances after the last stage of a step.
    """
    if solver not in SOLVERS:
        raise ValueError(f"Solver {solver} not supported. Supported solvers: {list(SOLVERS)}")
    rows = torch.zeros(n_iter, 8, dtype=torch.float32)
    idx, stage, dt = 0, 0, None
    for i in range(n_iter):
        if stage == 0:
            dt = sigmas[idx + 1] - sigmas[idx]          # fp32, like the reference
        if solver == "euler":
            rows[i, :5] = torch.tensor([1.0, 0.0, float(dt), 0.0, 0.0])
            idx += 1
            continue
        if solver in ("heun-2", "midpoint-2"):
            heun = solver == "heun-2"
            if stage == 0:
                rows[i, :5] = torch.tensor([1.0, 0.0, float(dt if heun else dt / 2), 0.5 if heun else 0.0,
                                            float(STEP_SAVE_X | STEP_ACC_RESET)])
                stage = 1
            else:
                rows[i, :5] = torch.tensor([0.5 if heun else 1.0, 1.0 if heun else 0.0, float(dt), 0.0,
                                            float(STEP_USE_SAVED)])
                stage, idx = 0, idx + 1
            continue
        # kutta-4
        if stage == 0:
            rows[i, :5] = torch.tensor([1.0, 0.0, float(dt / 2), 1.0 / 6.0, float(STEP_SAVE_X | STEP_ACC_RESET)])
        elif stage == 1:
            rows[i, :5] = torch.tensor([1.0, 0.0, float(dt / 2), 1.0 / 3.0, 0.0])
        elif stage == 2:
            rows[i, :5] = torch.tensor([1.0, 0.0, float(dt), 1.0 / 3.0, 0.0])
        else:
            rows[i, :5] = torch.tensor([1.0 / 6.0, 1.0, float(dt), 0.0, float(STEP_USE_SAVED)])
        stage = (stage + 1) % 4
        if stage == 0:
            idx += 1
    return rows


def timestep_features(t: torch.Tensor, dim: int = 256, max_period: int = 10000) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def rope_table(n_pos: int, dim: int = 128, theta: float = 10000.0):
    """cos/sin [n_pos, dim/2] for positions 0..n_pos-1 (pair k uses theta^(-2k/dim))."""
    pos = torch.linspace(0.0, float(n_pos), n_pos + 1, dtype=torch.float32)[:n_pos]
    idx = torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2]
    freqs = torch.pow(torch.tensor(theta, dtype=torch.float32).expand_as(idx), -(idx / torch.tensor(float(dim))))
    ang = torch.outer(pos, freqs)
    return ang.cos().contiguous(), ang.sin().contiguous()


def nearest_exact_index(out_len: int, in_len: int) -> torch.Tensor:
    """Source index of F.interpolate(x, size=out_len, mode='nearest-exact') (hifi_foley.py:44,58,761).
    ATen evaluates it in *float32*: scale = float(in)/float(out), idx = min(int(floor((i + 0.5f) * scale)),
    in-1) - a float64 evaluation picks the neighbouring row for ~5 % of the widget-reachable
    durations (1.1 s, 2.7 s, 30.1 s ...), so the float32 arithmetic is reproduced operation by operation."""
    scale = torch.tensor(float(in_len), dtype=torch.float32) / torch.tensor(float(out_len), dtype=torch.float32)
    i = torch.arange(out_len, dtype=torch.float32)
    return torch.clamp(torch.floor((i + 0.5) * scale).long(), max=in_len - 1)


def interleaved_positions(la: int, lv: int):
    """audio token i -> 2i ; visual token j -> 2*floor((j+0.5)*la/lv)+1."""
    down = nearest_exact_index(lv, la)
    up = nearest_exact_index(la, lv)
    if not torch.equal(up[down], torch.arange(lv)):
        raise ValueError(f"interleaved RoPE is not a pure re-indexing for La={la}, Lv={lv}")
    return 2 * torch.arange(la), 2 * down + 1


def build_tables(la: int, lv: int, ls: int, lt: int, steps: int, solver: str, shift: float,
                 time_freq_dim: int = 256, fp8_time: Optional[torch.dtype] = None) -> Dict[str, torch.Tensor]:
    """fp8_time: round the sinusoid timestep features through this fp8 type (fp8-wrapped model under
    autocast, embed_layers.py:134 - golden g8)."""
    sig = sigma_grid(steps, shift)
    ts = model_timesteps(sig)
    n_iter = steps
    pa, pv = interleaved_positions(la, lv)
    rope_len = max(2 * la, lt, lv) + 1
    cos, sin = rope_table(rope_len)
    return {
        "sigmas": sig,
        "timesteps": ts,
        "t_feat": (timestep_features(ts, time_freq_dim) if fp8_time is None
                   else timestep_features(ts, time_freq_dim).to(fp8_time).to(torch.float32)).contiguous(),
        "solver_coef": solver_table(sig, solver, n_iter),
        "rope_cos": cos, "rope_sin": sin,
        "pos_audio_self": pa.to(torch.int32), "pos_visual_self": pv.to(torch.int32),
        "pos_linear": torch.arange(max(la, lv, lt), dtype=torch.int32),
        "sync_gather": nearest_exact_index(la, ls).to(torch.int32),
    }
