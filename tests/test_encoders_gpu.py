"""The conditioning encoders ON THE HIP ENGINE (SURVEY 8f N2 stage 2, host/encoders_hip.py): Synchformer against the
golden frozen from the reference's own MotionFormer (tests/golden/g11_v2a.npz - NOT against torch running the same
restatement), the SigLIP vision tower against `transformers`' SiglipVisionModel on the CPU, and the GPU branch of the
uint8 resize against the CPU uint8 kernel torchvision's v2.Resize dispatches to."""
import pytest
import torch

from conftest import golden, rel_err
from foley_amd.host import encoders as E, encoders_hip as EH, synth

pytestmark = pytest.mark.gpu


def test_synchformer_on_the_engine_matches_reference_golden(dev):
    """Every GEMM / attention / LayerNorm of the Synchformer visual extractor through libfoley_hip.so (12 divided
    space-time blocks, final norm, spatial aggregation layer) on golden g11's two overlapping segments.  fp32 operands
    (parity mode) against the reference's fp32 CPU output: 2e-5; fp16 operands - what the reference's GPU path computes
    in (feature_utils.py:99-104, autocast fp16) - and bf16 to their rounding levels."""
    g = golden("g11_v2a")
    sd = {k: v.to(dev) for k, v in synth.materialize(E.synchformer_schema()).items()}
    frames = synth.synth_tensor("g11.frames", (24, 3, 224, 224), 0.55).to(dev)
    f32 = EH.encode_video_with_sync_hip(sd, frames, torch.float32)
    assert f32.shape == (1, 16, 768)
    e32 = rel_err(f32, g["sync_feat"])
    f16 = EH.encode_video_with_sync_hip(sd, frames, torch.float16)
    b16 = EH.encode_video_with_sync_hip(sd, frames, torch.bfloat16)
    e16, eb = rel_err(f16, g["sync_feat"]), rel_err(b16, g["sync_feat"])
    print("Synchformer on the HIP engine vs the reference's MotionFormer: fp32 %.2e, fp16 %.2e, bf16 %.2e" % (e32, e16, eb))
    assert e32 < 2e-5 and e16 < 5e-3 and eb < 4e-2
    # batches of segments are independent: one segment alone equals its rows of the pair
    one = EH.encode_video_with_sync_hip(sd, frames[:16].contiguous(), torch.float32)
    assert rel_err(one, f32[:, :8]) < 1e-5
    with pytest.raises(ValueError):
        EH.encode_video_with_sync_hip(sd, frames[:15], torch.float32)


def _small_siglip(layers=2, image=128, patch=16):
    from transformers import SiglipVisionConfig, SiglipVisionModel
    torch.manual_seed(0)
    cfg = SiglipVisionConfig(hidden_size=768, num_hidden_layers=layers, num_attention_heads=12, intermediate_size=3072,
                             image_size=image, patch_size=patch)
    m = SiglipVisionModel(cfg).eval()
    with torch.no_grad():      # default init leaves several tensors at zero / one: give every parameter a value
        gen = torch.Generator().manual_seed(1)
        for n, p_ in m.named_parameters():
            if p_.dim() == 1 and ("norm" in n and n.endswith("weight")):
                p_.copy_(1 + 0.1 * torch.randn(p_.shape, generator=gen))
            elif p_.dim() == 1:
                p_.copy_(0.05 * torch.randn(p_.shape, generator=gen))
    return m


def test_siglip_vision_tower_on_the_engine(dev):
    """`get_image_features` of transformers' SigLIP vision model (what feature_utils.py:63-78 calls per 8 fps frame) as
    restated on the engine over the model's state dict - ViT-B width (768, 12 heads of 64, MLP 3072, GELU-tanh), pre-norm
    encoder, attention-pooling head - against the HF module itself on the CPU in fp32."""
    m = _small_siglip()
    sd = {k: v.detach().to(dev) for k, v in m.state_dict().items()}
    px = torch.randn(3, 3, 128, 128, generator=torch.Generator().manual_seed(2))
    with torch.inference_mode():
        ref = m(pixel_values=px).pooler_output
    f32 = EH.siglip_image_features_hip(sd, px.to(dev), torch.float32)
    f16 = EH.siglip_image_features_hip(sd, px.to(dev), torch.float16)
    b16 = EH.siglip_image_features_hip(sd, px.to(dev), torch.bfloat16)
    e32, e16, eb = rel_err(f32, ref), rel_err(f16, ref), rel_err(b16, ref)
    print("SigLIP vision tower on the HIP engine vs transformers (CPU fp32): fp32 %.2e, fp16 %.2e, bf16 %.2e" % (e32, e16, eb))
    assert f32.shape == ref.shape and e32 < 2e-5 and e16 < 5e-3 and eb < 4e-2


def test_siglip_full_depth_benchmarked_tower(dev):
    """The tower bench.py times (google/siglip2-base-patch16-512's architecture: 12 layers, 512 px, 1024 tokens per frame)
    at FULL depth and resolution, one frame, against transformers' module on the CPU in fp32 - the benchmarked encoder is
    parity-checked, not only finite; a second frame checks batch independence at this size."""
    m = _small_siglip(layers=12, image=512, patch=16)
    sd = {k: v.detach().to(dev) for k, v in m.state_dict().items()}
    px = torch.randn(2, 3, 512, 512, generator=torch.Generator().manual_seed(6))
    with torch.inference_mode():
        ref = m(pixel_values=px[:1]).pooler_output
    f32 = EH.siglip_image_features_hip(sd, px.to(dev), torch.float32)
    b16 = EH.siglip_image_features_hip(sd, px.to(dev), torch.bfloat16)
    e32, eb = rel_err(f32[:1], ref), rel_err(b16[:1], ref)
    print("SigLIP2 ViT-B/16-512, 12 layers, on the HIP engine vs transformers (CPU fp32): fp32 %.2e, bf16 %.2e" % (e32, eb))
    assert f32.shape == (2, 768) and e32 < 2e-5 and eb < 6e-2
    one = EH.siglip_image_features_hip(sd, px[1:].to(dev).contiguous(), torch.float32)
    assert rel_err(one, f32[1:]) < 1e-5


def test_clap_full_depth_benchmarked_encoder(dev):
    """The CLAP text encoder bench.py times (ClapTextConfig defaults = laion/larger_clap_general's RoBERTa-base: 12 layers,
    MLP 3072, vocabulary 50265) at full depth, one prompt of 12 tokens + <s> </s>, against transformers on the CPU in fp32."""
    from transformers import ClapTextConfig, ClapTextModelWithProjection
    torch.manual_seed(5)
    m = ClapTextModelWithProjection(ClapTextConfig()).eval()
    with torch.no_grad():
        gen = torch.Generator().manual_seed(6)
        for n, p_ in m.named_parameters():
            if p_.dim() == 1:
                p_.copy_((1.0 if "LayerNorm.weight" in n else 0.0) + 0.05 * torch.randn(p_.shape, generator=gen))
    g = torch.Generator().manual_seed(7)
    ids = torch.cat([torch.tensor([0]), torch.randint(3, 50000, (12,), generator=g), torch.tensor([2])])[None]
    mask = torch.ones_like(ids)
    with torch.inference_mode():
        ref = m(input_ids=ids, attention_mask=mask, return_dict=True).last_hidden_state
    sd = {k: v.detach().to(dev) for k, v in m.state_dict().items() if k.startswith("text_model.") and v.is_floating_point()}
    f32 = EH.clap_text_hidden_hip(sd, ids.to(dev), mask.to(dev), torch.float32)
    b16 = EH.clap_text_hidden_hip(sd, ids.to(dev), mask.to(dev), torch.bfloat16)
    e32, eb = rel_err(f32, ref), rel_err(b16, ref)
    print("CLAP RoBERTa-base, 12 layers, on the HIP engine vs transformers (CPU fp32): fp32 %.2e, bf16 %.2e" % (e32, eb))
    assert f32.shape == ref.shape and e32 < 2e-5 and eb < 6e-2


def test_clap_text_encoder_on_the_engine(dev):
    """`last_hidden_state` of transformers' ClapTextModelWithProjection (the text tokens of feature_utils.py:133-138) as restated
    on the engine - RoBERTa embeddings with pad-aware positions, post-norm layers, exact GELU, LayerNorm eps 1e-12 - for two
    prompts of different length (right-padded: the shorter one's pad rows are part of the output, like in the reference),
    against the HF module on the CPU in fp32."""
    from transformers import ClapTextConfig, ClapTextModelWithProjection
    torch.manual_seed(3)
    m = ClapTextModelWithProjection(ClapTextConfig(hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=256,
                                                   vocab_size=300, max_position_embeddings=90, projection_dim=64)).eval()
    with torch.no_grad():
        gen = torch.Generator().manual_seed(4)
        for n, p_ in m.named_parameters():
            if p_.dim() == 1:
                p_.copy_((1.0 if "LayerNorm.weight" in n else 0.0) + 0.05 * torch.randn(p_.shape, generator=gen))
    ids = torch.tensor([[0, 17, 45, 99, 120, 7, 33, 250, 2], [0, 88, 5, 2, 1, 1, 1, 1, 1]])
    mask = (ids != 1).long()
    with torch.inference_mode():
        ref = m(input_ids=ids, attention_mask=mask, return_dict=True).last_hidden_state
    sd = {k: v.detach().to(dev) for k, v in m.state_dict().items() if v.is_floating_point()}
    f32 = EH.clap_text_hidden_hip(sd, ids.to(dev), mask.to(dev), torch.float32)
    f16 = EH.clap_text_hidden_hip(sd, ids.to(dev), mask.to(dev), torch.float16)
    b16 = EH.clap_text_hidden_hip(sd, ids.to(dev), mask.to(dev), torch.bfloat16)
    e32, e16, eb = rel_err(f32, ref), rel_err(f16, ref), rel_err(b16, ref)
    print("CLAP text encoder on the HIP engine vs transformers (CPU fp32): fp32 %.2e, fp16 %.2e, bf16 %.2e" % (e32, e16, eb))
    assert f32.shape == ref.shape and e32 < 2e-5 and e16 < 5e-3 and eb < 4e-2
    with pytest.raises(Exception):
        EH.clap_text_hidden_hip(sd, ids.to(dev), torch.tensor([[0, 1, 1, 1, 1, 1, 1, 1, 1], [1] * 9]).to(dev), torch.float32)   # left padding


def test_gpu_uint8_resize_matches_the_cpu_uint8_kernel(dev):
    """torchvision's v2.Resize(bicubic, antialias) runs the native uint8 kernel on the CPU (where the reference
    pre-processes, utils.py:262-283); encoders._resize_u8 on the GPU runs the same two fixed-point passes on
    libfoley_hip.so (foley_op_resize_aa_u8).  torchvision is not in the image, so the reference is the very CPU kernel
    v2.Resize dispatches to: F.interpolate on uint8 - separable, uint8 intermediate.  Integer arithmetic on both sides:
    EQUAL, byte for byte (the float32 two-pass branch this replaced was one grey level off on 0.6 % of the pixels; a single
    2-D float interpolation is off by up to 22 levels on 20 % of them).  Shapes: the two pipelines' (SigLIP2 512x512,
    Synchformer short edge 224), up-scaling, an identity axis, odd widths (1- and 2-byte vertical lanes), tiny axes."""
    g = torch.Generator().manual_seed(5)
    for shape, size in (((4, 3, 96, 160), (224, 373)), ((2, 3, 480, 640), (512, 512)), ((2, 3, 300, 224), (300, 224)),
                        ((3, 3, 480, 640), (224, 298)), ((1, 3, 360, 641), (224, 399)), ((2, 2, 1080, 1920), (512, 512)),
                        ((1, 1, 7, 3), (3, 7)), ((1, 1, 1, 9), (5, 1))):
        fr = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
        fr[..., : shape[-2] // 2, :] = fr[..., : shape[-2] // 2, :] // 64 * 85       # flat areas + hard edges
        cpu = E._resize_u8(fr, size)
        gpu = E._resize_u8(fr.to(dev), size).cpu()
        assert gpu.dtype == torch.uint8 and gpu.shape == cpu.shape
        assert torch.equal(gpu, cpu), (shape, size, int((gpu.int() - cpu.int()).abs().max()), float((gpu != cpu).float().mean()))


def test_resize_pass_against_the_oracle(dev):
    """foley_op_resize_aa_u8 through the C ABI, one axis at a time, against oracle.resize_u8_axis on every axis position
    (outer / inner extents of 1, odd inner extents, unaligned views are made contiguous by the caller)."""
    import numpy as np
    from foley_amd.host import runtime as rt
    from oracle import foley_oracle as O
    g = torch.Generator().manual_seed(11)
    for shape, axis, n_out in (((5, 37, 12), 1, 19), ((5, 37, 13), 1, 64), ((1, 50, 1), 1, 7), ((6, 9, 40), 2, 33), ((3, 4, 5, 6), 2, 11),
                               ((64, 3), 0, 100)):
        x = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
        xmin, xsize, w, prec = E.aa_tables(shape[axis], n_out)
        got = rt.op_resize_aa_u8(x.to(dev), axis, n_out, torch.from_numpy(xmin).to(dev), torch.from_numpy(xsize).to(dev),
                                 torch.from_numpy(w).to(dev), prec).cpu().numpy()
        assert np.array_equal(got, O.resize_u8_axis(x.numpy(), axis, n_out)), (shape, axis, n_out)
    with pytest.raises(rt.FoleyRuntimeError):
        rt.op_resize_aa_u8(torch.zeros(4, 4, device=dev), 1, 2, torch.zeros(2, dtype=torch.int32, device=dev),
                           torch.zeros(2, dtype=torch.int32, device=dev), torch.zeros(2, 5, dtype=torch.int16, device=dev), 14)
