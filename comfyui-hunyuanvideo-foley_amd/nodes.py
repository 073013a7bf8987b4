"""ComfyUI node layer of the MI355X-native Foley path.

Drop-in boundary: the six node keys, display names, socket type strings, widget
names / order / defaults / ranges and return tuples below are those of the reference's
`nodes.py` (node classes at :57-70, :156-168, :211-242, :433-457, :609-631, :636-663; mappings
:668-683), so `example_workflows/HunyuanVideoFoleyExample.json` loads unchanged.  What sits
behind them is new: the loaders pack checkpoints into a device arena for libfoley_hip.so, the
sampler enqueues the whole denoising loop + DAC decode on the GPU.  The `TORCH_COMPILE_CFG` and
`BLOCKSWAPARGS` sockets are accepted and ignored: there is no tracing compiler on this path and
the 288 GB of HBM make block swapping pointless.

ComfyUI modules (`folder_paths`, `comfy.*`) are imported lazily so the package also imports in
test environments without ComfyUI.
"""
from __future__ import annotations

import logging
import os

import torch

from .host import config as _cfg
from .host import encoders as _enc
from .host import sampler as _sampler

log = logging.getLogger("HunyuanVideo-Foley[MI355X]")

_SOLVERS = ["euler", "heun-2", "midpoint-2", "kutta-4"]
_PKG_DIR = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- ComfyUI glue
def _folder_paths():
    try:
        import folder_paths  # type: ignore
    except Exception:
        return None
    foley_dir = os.path.join(folder_paths.models_dir, "foley")
    if "foley" not in folder_paths.folder_names_and_paths:      # models/foley/ like the reference
        folder_paths.folder_names_and_paths["foley"] = ([foley_dir], folder_paths.supported_pt_extensions)
    return folder_paths


def _foley_files(substr=None):
    fp = _folder_paths()
    names = fp.get_filename_list("foley") if fp is not None else []
    return [f for f in names if substr is None or substr in f]


def _torch_device():
    try:
        import comfy.model_management as mm  # type: ignore
        return mm.get_torch_device()
    except Exception:
        return torch.device("cuda:0")


def _load_state_dict(path):
    try:
        from comfy.utils import load_torch_file  # type: ignore
    except ImportError:
        load_torch_file = None
    if load_torch_file is not None:
        obj = load_torch_file(path, device=torch.device("cpu"))
    elif str(path).endswith(".safetensors"):
        from safetensors.torch import load_file
        obj = load_file(path)
    else:
        # tensors only: never unpickle arbitrary objects from a downloaded .pth (comfy's loader does the same)
        obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and isinstance(obj.get("state_dict"), dict):   # utils.py:49-59
        obj = obj["state_dict"]
    return {k: v for k, v in obj.items() if isinstance(v, torch.Tensor)}


def fp8_wrapped_key(key: str, tensor) -> bool:
    """Does the reference's _wrap_fp8_inplace (utils.py:408-485) store this tensor in fp8?  Its deny
    list never matches, so: the weight of EVERY nn.Linear / nn.Conv1d and nothing else -
    learned feature rows, position tables, norm gains and biases stay full precision
    (tests/golden/g8_fp8.npz lists the 56 wrapped modules of the tiny config)."""
    return key.endswith(".weight") and tensor.dim() >= 2 and tensor.is_floating_point()


def fp8_round_state_dict(state_dict, qmode: str, autocast: bool = True, param_dtype=torch.float32):
    """Weight-only fp8 storage = plain cast, no scales; values are rounded once here and the kernels
    run on the exactly representable bf16/fp32 images.  Order as in the reference loader
    (nodes.py:96-124): checkpoint -> parameters of the requested precision (`param_dtype`; fp8
    checkpoint tensors are upcast exactly) -> cast to the fp8 storage type, so an fp32 checkpoint is
    rounded twice and an fp8 checkpoint of the other flavour is re-rounded.  Under autocast
    (bf16/fp16 compute, the only way the reference runs fp8 models) the first TimestepEmbedder bias
    is rounded too: the wrapper casts it to the activation dtype, which embed_layers.py:134 has made
    fp8 (golden g8, "Q14"); the matching feature rounding lives in host/sampler.py::build_plan."""
    out = {}
    qd = {"fp8_e4m3fn": torch.float8_e4m3fn, "fp8_e5m2": torch.float8_e5m2}.get(qmode)
    for k, v in state_dict.items():
        if v.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
            v = v.to(torch.float32)
        if qd is not None and v.is_floating_point() and (fp8_wrapped_key(k, v) or (autocast and k == "time_in.mlp.0.bias")):
            v = v.to(torch.float32).to(param_dtype).to(qd).to(torch.float32)
        out[k] = v
    return out


def round_params(state_dict, param_dtype):
    """The reference loads the checkpoint into parameters of the requested precision
    (`foley_model.to(dtype)`, nodes.py:96-106): EVERY floating tensor - biases, norm gains, the sync
    position table, the learned empty-feature rows - is rounded to bf16 / fp16, not only the matrices.
    The packed arena keeps those small tensors in fp32 storage, so the rounding is applied to their
    values here (exact for fp8 / already-rounded checkpoints)."""
    if param_dtype == torch.float32:
        return state_dict
    out = {}
    for k, v in state_dict.items():
        if v.is_floating_point() and v.dtype not in (torch.float8_e4m3fn, torch.float8_e5m2):
            v = v.to(param_dtype).to(torch.float32)
        out[k] = v
    return out


def resolve_quantization(quantization: str, detected):
    """Reference nodes.py:109-122: 'auto' honours fp8 tensors found in the checkpoint, else e4m3fn
    (its e5m2 fallback is for compute capability < 9; gfx950 reports 9.x and implements OCP e4m3fn /
    e5m2 natively) - i.e. the DEFAULT widget value rounds every Linear / Conv weight through fp8,
    exactly like the reference does on a capable device."""
    if quantization == "none":
        return "none"
    if quantization == "auto":
        return detected or "fp8_e4m3fn"
    return quantization


def detect_ckpt_fp8(state_dict):
    """'fp8_e5m2' / 'fp8_e4m3fn' if the checkpoint stores such tensors, else None (utils.py:492-504)."""
    for v in state_dict.values():
        if v.dtype == torch.float8_e5m2:
            return "fp8_e5m2"
        if v.dtype == torch.float8_e4m3fn:
            return "fp8_e4m3fn"
    return None


def detect_ckpt_major_precision(state_dict):
    """Dominant dtype among bf16 / fp16 / fp32 by element count (utils.py:507-515)."""
    counts = {torch.bfloat16: 0, torch.float16: 0, torch.float32: 0}
    for v in state_dict.values():
        if v.dtype in counts:
            counts[v.dtype] += v.numel()
    if not any(counts.values()):
        return torch.bfloat16
    return max(counts, key=counts.get)


class AttributeDict(dict):
    """dict with attribute access - the HUNYUAN_DEPS payload type."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


# ----------------------------------------------------------------------------- NODE 1: model loader
class HunyuanModelLoader:
    @classmethod
    def INPUT_TYPES(cls):
        return {
            "required": {
                "model_name": (_foley_files(),),
                "precision": (["auto", "bf16", "fp16", "fp32"], {"default": "bf16", "tooltip": "Compute dtype of the GEMM operands (fp32 = parity mode on fp32 MFMA; auto = detect from checkpoint)"}),
                "quantization": (["none", "fp8_e4m3fn", "fp8_e5m2", "auto"], {"default": "auto", "tooltip": "FP8 weight-only storage of the checkpoint (values are rounded through fp8 like the reference, compute stays bf16 / fp16)"}),
            },
        }

    RETURN_TYPES = ("HUNYUAN_MODEL",)
    FUNCTION = "build_model"
    CATEGORY = "audio/HunyuanFoley"

    @staticmethod
    def pack_state_dict(state_dict, precision="bf16", quantization="auto", device=None, cfg=None, dac_cfg=None):
        """state dict -> FoleyModel.  precision=fp16 (and `auto` on an fp16 checkpoint) computes in fp16 like the
        reference does (parameters .to(float16), torch.autocast(float16): nodes.py:89-106, utils.py:229-234) - the
        v_mfma_f32_32x32x16_f16 instantiation of the same kernels."""
        cfg = cfg or _cfg.load_yaml_config(os.path.join(_PKG_DIR, "configs", "hunyuanvideo-foley-xxl.yaml"))
        dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}.get(precision)
        detected = detect_ckpt_fp8(state_dict)
        if precision == "auto" or dtype is None:
            major = detect_ckpt_major_precision(state_dict)
            dtype = major       # the checkpoint's dominant dtype: fp32, bf16 or fp16 (utils.py:507-515)
        qmode = resolve_quantization(quantization, detected)
        param_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}.get(precision, dtype)
        sd = round_params(state_dict, param_dtype)
        sd = fp8_round_state_dict(sd, qmode, autocast=dtype != torch.float32, param_dtype=param_dtype)
        return _sampler.FoleyModel(cfg, sd, dtype, device or _torch_device(), dac_cfg=dac_cfg or _cfg.DAC48K,
                                   quantization=qmode)

    def build_model(self, model_name, precision, quantization):
        fp = _folder_paths()
        if fp is None:
            raise RuntimeError("ComfyUI's folder_paths module is required to resolve model files")
        sd = _load_state_dict(fp.get_full_path("foley", model_name))
        model = self.pack_state_dict(sd, precision, quantization)
        log.info("Loaded HunyuanVideoFoley main model: %s", model_name)
        return (model,)


# ----------------------------------------------------------------------------- NODE 2: dependencies loader
class HunyuanDependenciesLoader:
    @classmethod
    def INPUT_TYPES(cls):
        return {
            "required": {
                "vae_name": (_foley_files("vae"),),
                "synchformer_name": (_foley_files("synch"),),
            }
        }

    RETURN_TYPES = ("HUNYUAN_DEPS",)
    FUNCTION = "load_dependencies"
    CATEGORY = "audio/HunyuanFoley"

    def load_dependencies(self, vae_name, synchformer_name):
        fp = _folder_paths()
        if fp is None:
            raise RuntimeError("ComfyUI's folder_paths module is required to resolve model files")
        device = _torch_device()
        deps = AttributeDict()
        deps["dac_model"] = _sampler.FoleyDAC(_load_state_dict(fp.get_full_path("foley", vae_name)), device)
        deps["synchformer_path"] = fp.get_full_path("foley", synchformer_name)
        # Conditioning encoders stay on PyTorch-ROCm and are created on first use: the
        # Synchformer visual extractor is this repo's functional restatement over the checkpoint's
        # tensors (host/encoders.py), SigLIP2 and CLAP come from `transformers` like in the reference
        # (nodes.py:198-201).
        deps["syncformer_model"] = None      # state dict of the visual extractor once loaded
        deps["siglip2_model"] = None
        deps["clap_tokenizer"] = None
        deps["clap_model"] = None
        deps["device"] = device
        return (deps,)


SIGLIP2_REPO, CLAP_REPO = "google/siglip2-base-patch16-512", "laion/larger_clap_general"


def _ensure_text_encoder(deps):
    if deps.get("clap_model") is None:
        from transformers import AutoTokenizer, ClapTextModelWithProjection
        deps["clap_tokenizer"] = AutoTokenizer.from_pretrained(CLAP_REPO)
        deps["clap_model"] = ClapTextModelWithProjection.from_pretrained(CLAP_REPO).eval()
    return deps


def _ensure_visual_encoders(deps, device, dtype):
    """SigLIP2 through transformers, Synchformer from the file the Dependencies Loader resolved.  Both
    live on the GPU in the model's dtype like in the reference sampler (nodes.py:283-284)."""
    if deps.get("siglip2_model") is None:
        from transformers import AutoModel
        deps["siglip2_model"] = AutoModel.from_pretrained(SIGLIP2_REPO).eval()
    deps["siglip2_model"].to(device=device, dtype=dtype)
    sync = deps.get("syncformer_model")
    if sync is None:
        path = deps.get("synchformer_path")
        if not path:
            raise RuntimeError("HUNYUAN_DEPS carries no Synchformer checkpoint (synchformer_path)")
        sync = _load_state_dict(path)
    if not isinstance(sync, dict):
        raise RuntimeError("HUNYUAN_DEPS['syncformer_model'] must be the Synchformer state dict")
    first = next(iter(sync.values()))
    if first.device != torch.device(device) or first.dtype != dtype or any(not k.startswith("vfeat_extractor.") for k in sync):
        sync = _enc.load_synchformer_state(sync, device, dtype)
    deps["syncformer_model"] = sync
    return deps


@torch.inference_mode()
def encode_text_feat(prompts, deps, device, dtype=None):
    """CLAP last_hidden_state for [negative, positive] (feature_utils.py:133-138)."""
    _ensure_text_encoder(deps)
    deps["clap_model"].to(device=device, dtype=dtype) if dtype is not None else deps["clap_model"].to(device)
    return _enc.encode_text_feat(deps["clap_tokenizer"], deps["clap_model"], prompts, device)


select_frames = _enc.select_frames


# ----------------------------------------------------------------------------- NODE 3: sampler
class HunyuanFoleySampler:
    SAMPLER_NAMES = _SOLVERS

    @classmethod
    def INPUT_TYPES(cls):
        return {
            "required": {
                "hunyuan_model": ("HUNYUAN_MODEL",),
                "hunyuan_deps": ("HUNYUAN_DEPS",),
                "frame_rate": ("FLOAT", {"default": 16, "min": 1, "max": 120, "step": 0.1, "tooltip": "The framerate of the input image sequence"}),
                "duration": ("FLOAT", {"default": 5.0, "min": 1, "max": 60.0, "step": 0.1, "tooltip": "Duration of the audio to generate in seconds"}),
                "prompt": ("STRING", {"multiline": True, "default": "A person walks on frozen ice"}),
                "negative_prompt": ("STRING", {"multiline": True, "default": "noisy, harsh"}),
                "cfg_scale": ("FLOAT", {"default": 4.5, "min": 1.0, "max": 10.0, "step": 0.1, "tooltip": "Classifier-Free Guidance scale"}),
                "steps": ("INT", {"default": 50, "min": 10, "max": 100, "step": 1, "tooltip": "Number of denoising steps"}),
                "sampler": (cls.SAMPLER_NAMES, {"default": "euler", "tooltip": "Flow-match ODE solver"}),
                "batch_size": ("INT", {"default": 1, "min": 1, "max": 6, "step": 1, "tooltip": "Number of audio variations to generate at once"}),
                "seed": ("INT", {"default": 0, "min": 0, "max": 0xffffffffffffffff}),
                "force_offload": ("BOOLEAN", {"default": True, "tooltip": "Kept for workflow compatibility; weights stay resident in HBM"}),
            },
            "optional": {
                "image": ("IMAGE",),
                "torch_compile_cfg": ("TORCH_COMPILE_CFG", {"tooltip": "Accepted for workflow compatibility and ignored (no tracing compiler on this path)."}),
                "block_swap_args": ("BLOCKSWAPARGS", {"tooltip": "Accepted for workflow compatibility and ignored (288 GB HBM: nothing to swap)."}),
            }
        }

    RETURN_TYPES = ("AUDIO", "AUDIO")
    RETURN_NAMES = ("audio_first", "audio_batch")
    FUNCTION = "generate_audio"
    CATEGORY = "audio/HunyuanFoley"

    def generate_audio(self, hunyuan_model, hunyuan_deps, frame_rate, duration, prompt, negative_prompt, cfg_scale,
                       steps, sampler, batch_size, seed, force_offload, image=None, torch_compile_cfg=None,
                       block_swap_args=None, features=None):
        """`features` (not a ComfyUI socket) lets callers inject precomputed conditioning
        {'siglip2_feat','syncformer_feat','text_feat','uncond_text_feat'} - used by tests/bench."""
        model, deps = hunyuan_model, hunyuan_deps
        device = model.device
        rng = torch.Generator(device="cpu").manual_seed(seed)          # nodes.py:273
        audio_len_in_s = duration
        if features is not None:
            visual = {k: features[k] for k in ("siglip2_feat", "syncformer_feat")}
            text = {k: features[k] for k in ("text_feat", "uncond_text_feat")}
            audio_len_in_s = features.get("audio_len_in_s", duration)
        elif image is not None:
            visual, text, audio_len_in_s = self._video_features(image, duration, frame_rate, prompt,
                                                                negative_prompt, deps, device, model.dtype)
        else:
            # text-to-audio: learned "empty" visual rows (nodes.py:326-333)
            clip_len = int(duration * 8)
            sync_len = int(((int(duration * 25) - 16) // 8 + 1) * 8)
            visual = {"siglip2_feat": model.get_empty_clip_sequence(bs=1, len=clip_len),
                      "syncformer_feat": model.get_empty_sync_sequence(bs=1, len=sync_len)}
            res = encode_text_feat([negative_prompt, prompt], deps, device, model.dtype)
            text = {"text_feat": res[1:], "uncond_text_feat": res[:1]}
        pbar = None
        try:
            import comfy.utils  # type: ignore
            pbar = comfy.utils.ProgressBar(steps)
        except Exception:
            pass
        progress = (lambda i, n: pbar.update_absolute(i, n)) if pbar is not None else None
        n_dev = torch.cuda.device_count() if os.environ.get("FOLEY_DATA_PARALLEL", "0") == "1" else 1
        if batch_size > 1 and n_dev > 1 and model.arena is not None:
            # clips are independent: shard them over the node's GPUs (host/sampler.py::denoise_process_multi; the widget
            # list is the reference's, so the switch is an environment variable - INTEGRATION.md)
            devs = [model.device] + [torch.device("cuda", i) for i in range(n_dev) if i != model.device.index]
            reps = _sampler.replicate(model, deps["dac_model"], devs[:batch_size])
            audio, sr = _sampler.denoise_process_multi(visual, text, audio_len_in_s, reps, cfg_scale, steps, batch_size,
                                                       sampler, generator=rng, progress=progress)
        else:
            audio, sr = _sampler.denoise_process_with_generator(
                visual, text, audio_len_in_s, model, deps["dac_model"], guidance_scale=cfg_scale,
                num_inference_steps=steps, batch_size=batch_size, sampler=sampler, generator=rng, progress=progress)
        waveform_batch = audio.float().cpu()
        first = {"waveform": waveform_batch[0].unsqueeze(0), "sample_rate": sr}
        return (first, {"waveform": waveform_batch, "sample_rate": sr})

    @staticmethod
    @torch.inference_mode()
    def _video_features(image, duration, frame_rate, prompt, negative_prompt, deps, device, dtype=torch.float32):
        """Video-to-Audio conditioning (nodes.py:290-320 + utils.py:262-292): resample the IMAGE batch to
        8 fps / 25 fps, run SigLIP2 and the Synchformer visual extractor, CLAP for the two prompts.  The
        audio length follows the sync stream: len(frames_25fps) / 25."""
        _ensure_visual_encoders(deps, device, dtype)
        f8, f25 = _enc.select_frames(image, duration, frame_rate, device=device if torch.device(device).type == "cuda" else None)
        visual, audio_len_in_s = _enc.video_features(f8, f25, deps["siglip2_model"], deps["syncformer_model"], device,
                                                     model_dtype=dtype)
        res = encode_text_feat([negative_prompt, prompt], deps, device, dtype)
        return visual, {"text_feat": res[1:], "uncond_text_feat": res[:1]}, audio_len_in_s


# ----------------------------------------------------------------------------- compat nodes
class HunyuanFoleyTorchCompile:
    """Kept so existing workflows load; the resulting config is ignored by the sampler."""

    @classmethod
    def INPUT_TYPES(cls):
        return {
            "required": {
                "backend": (["inductor"], {"default": "inductor"}),
                "fullgraph": ("BOOLEAN", {"default": False}),
                "mode": (["default", "reduce-overhead", "max-autotune"], {"default": "default"}),
                "dynamic": (["true", "false", "None"], {"default": "false"}),
                "dynamo_cache_limit": ("INT", {"default": 64, "min": 64, "max": 8192, "step": 64}),
            }
        }

    RETURN_TYPES = ("TORCH_COMPILE_CFG",)
    FUNCTION = "make_config"
    CATEGORY = "audio/HunyuanFoley"

    def make_config(self, backend, mode, dynamic, fullgraph, dynamo_cache_limit):
        dyn = {"true": True, "false": False, "None": None}.get(str(dynamic), False)
        return ({"backend": backend, "mode": mode, "dynamic": dyn, "fullgraph": fullgraph,
                 "dynamo_cache_limit": int(dynamo_cache_limit)},)


class HunyuanBlockSwap:
    @classmethod
    def INPUT_TYPES(cls):
        return {
            "required": {
                "blocks_to_swap": ("INT", {"default": 30, "min": 0, "max": 57, "step": 1}),
            },
            "optional": {
                "use_non_blocking": ("BOOLEAN", {"default": False}),
                "prefetch_blocks": ("INT", {"default": 1, "min": 0, "max": 10, "step": 1}),
                "block_swap_debug": ("BOOLEAN", {"default": False}),
            },
        }

    RETURN_TYPES = ("BLOCKSWAPARGS",)
    RETURN_NAMES = ("block_swap_args",)
    FUNCTION = "set_args"
    CATEGORY = "audio/HunyuanFoley"
    DESCRIPTION = "Accepted for compatibility; the MI355X path keeps all 57 blocks resident in HBM."

    def set_args(self, **kwargs):
        return (kwargs,)


class SelectAudioFromBatch:
    @classmethod
    def INPUT_TYPES(cls):
        return {
            "required": {
                "audio_batch": ("AUDIO", {"tooltip": "An audio object containing a batch of waveforms."}),
                "index": ("INT", {"default": 0, "min": 0, "max": 63, "tooltip": "The 0-based index of the audio to select from the batch."}),
            }
        }

    RETURN_TYPES = ("AUDIO",)
    FUNCTION = "select_audio"
    CATEGORY = "audio/utils"

    def select_audio(self, audio_batch, index):
        wave, sr = audio_batch["waveform"], audio_batch["sample_rate"]
        if index >= wave.shape[0]:
            log.warning("Index %d is out of bounds for audio batch of size %d. Clamping to last item.", index,
                        wave.shape[0])
            index = wave.shape[0] - 1
        return ({"waveform": wave[index].unsqueeze(0), "sample_rate": sr},)


NODE_CLASS_MAPPINGS = {
    "HunyuanModelLoader": HunyuanModelLoader,
    "HunyuanDependenciesLoader": HunyuanDependenciesLoader,
    "HunyuanFoleySampler": HunyuanFoleySampler,
    "HunyuanFoleyTorchCompile": HunyuanFoleyTorchCompile,
    "HunyuanBlockSwap": HunyuanBlockSwap,
    "SelectAudioFromBatch": SelectAudioFromBatch,
}
NODE_DISPLAY_NAME_MAPPINGS = {
    "HunyuanModelLoader": "Hunyuan-Foley Model Loader",
    "HunyuanDependenciesLoader": "Hunyuan-Foley Dependencies Loader",
    "HunyuanFoleySampler": "Hunyuan-Foley Sampler",
    "HunyuanFoleyTorchCompile": "Hunyuan-Foley Torch Compile",
    "HunyuanBlockSwap": "Hunyuan-Foley BlockSwap Settings",
    "SelectAudioFromBatch": "Select Audio From Batch",
}
