"""CPU tests of the Video-to-Audio conditioning path (nodes.py `_video_features`):
frame resampling, the two pre-processing pipelines, the Synchformer visual extractor against the
golden frozen from the reference's own MotionFormer (tests/golden/g11_v2a.npz, make_golden.py g11),
and the node-level feature extraction with small stand-in HF encoders (no checkpoints / network in
the image; the real SigLIP2 / CLAP weights are fetched by `from_pretrained` in production)."""
import pytest
import torch

from conftest import golden, rel_err
from foley_amd import nodes
from foley_amd.host import config as C, encoders as E, synth


def tiny_siglip():
    from transformers import SiglipConfig, SiglipModel
    cfg = SiglipConfig(vision_config=dict(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=128,
                                          image_size=512, patch_size=64),
                       text_config=dict(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=128,
                                        vocab_size=64, max_position_embeddings=16, bos_token_id=1, eos_token_id=2, pad_token_id=0))
    torch.manual_seed(0)
    return SiglipModel(cfg).eval()


def tiny_clap():
    from transformers import ClapTextConfig, ClapTextModelWithProjection
    torch.manual_seed(1)
    model = ClapTextModelWithProjection(ClapTextConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12,
                                                       intermediate_size=128, vocab_size=300, max_position_embeddings=90,
                                                       projection_dim=64)).eval()

    class Tok:
        """Byte-level stand-in for the CLAP (Roberta) tokenizer: <s> bytes </s>, right-padded with <pad>=1."""

        def __call__(self, texts, padding=True, return_tensors="pt"):
            ids = [[0] + [3 + (b % 250) for b in t.encode()][:75] + [2] for t in texts]
            n = max(len(i) for i in ids)
            batch = {"input_ids": torch.tensor([i + [1] * (n - len(i)) for i in ids]),
                     "attention_mask": torch.tensor([[1] * len(i) + [0] * (n - len(i)) for i in ids])}

            class B(dict):
                def to(self, dev):
                    return B({k: v.to(dev) for k, v in self.items()})
            return B(batch)
    return Tok(), model


def test_frame_selection_matches_reference_rule():
    """nodes.py:293-317: hold the last frame / cut to int(duration*frame_rate) frames, then
    linspace(0, n-1, int(duration*8 | *25)).long() - pinned by golden g11 for four (clip, duration, fps) cases."""
    g = golden("g11_v2a")
    for tag in "abcd":
        total, dur, fps = (float(v) for v in g["case_" + tag])
        total = int(total)
        ar = torch.arange(total)
        img = torch.zeros(total, 2, 3, 3)
        img[..., 0] = ((ar % 200).float() / 255.0).view(-1, 1, 1)      # frame i carries its own index in two channels
        img[..., 1] = ((ar // 200).float() / 255.0).view(-1, 1, 1)
        f8, f25 = E.select_frames(img, dur, fps)
        assert f8.dtype == torch.uint8 and f8.shape == (int(dur * 8), 3, 2, 3) and f25.shape[0] == int(dur * 25)
        n = int(dur * fps)
        got8 = f8[:, 0, 0, 0].long() + 200 * f8[:, 1, 0, 0].long()
        got25 = f25[:, 0, 0, 0].long() + 200 * f25[:, 1, 0, 0].long()
        last = torch.tensor(total - 1)                                  # the last frame is held beyond the clip's end
        assert torch.equal(got8, torch.minimum(g["idx8_" + tag], last)) and torch.equal(got25, torch.minimum(g["idx25_" + tag], last))
        assert int(g["idx25_" + tag].max()) == n - 1


def test_frame_selection_equals_whole_clip_conversion():
    """select_frames touches only the frame range the two index sets span (one contiguous copy, converted on the device in
    the GPU path); the reference converts the whole padded clip and then selects (nodes.py:293-317).  Same result, bit for
    bit, for clips shorter / longer than duration * frame_rate and fractional rates."""
    g = torch.Generator().manual_seed(9)
    for total, dur, fps in ((30, 2.0, 24.0), (200, 5.0, 16.0), (61, 2.5, 29.97), (10, 1.0, 8.0), (400, 3.3, 60.0)):
        img = torch.rand(total, 6, 7, 3, generator=g)
        n = int(dur * fps)
        full = torch.cat((img, img[-1:].repeat(n - total, 1, 1, 1)), dim=0) if n > total else img[:n]
        frames = (full * 255.0).byte().permute(0, 3, 1, 2)
        r8 = frames.index_select(0, torch.linspace(0, n - 1, int(dur * 8)).long())
        r25 = frames.index_select(0, torch.linspace(0, n - 1, int(dur * 25)).long())
        f8, f25 = E.select_frames(img, dur, fps)
        assert torch.equal(f8, r8) and torch.equal(f25, r25), (total, dur, fps)
        # the span is converted in bounded slabs (float32 frames are 12 bytes per pixel): any slab size gives the same frames
        saved = E._CHUNK_BYTES
        try:
            for cb in (1, 3 * img[0].numel() * 4, 7 * img[0].numel() * 4):
                E._CHUNK_BYTES = cb
                c8, c25 = E.select_frames(img, dur, fps)
                assert torch.equal(c8, r8) and torch.equal(c25, r25), (total, dur, fps, cb)
        finally:
            E._CHUNK_BYTES = saved


def test_preprocess_pipelines():
    """v2.Resize(bicubic, antialias) -> /255 -> Normalize(0.5, 0.5) (nodes.py:184-196): shapes, range, the
    short-edge / centre-crop geometry, constants stay constant, identity when no resize is needed."""
    g = torch.Generator().manual_seed(0)
    fr = torch.randint(0, 256, (3, 3, 96, 160), generator=g, dtype=torch.uint8)
    a = E.siglip2_preprocess(fr)
    assert a.shape == (3, 3, 512, 512) and a.dtype == torch.float32 and float(a.min()) >= -1.0 and float(a.max()) <= 1.0
    b = E.synchformer_preprocess(fr)
    assert b.shape == (3, 3, 224, 224)
    # geometry: short edge 96 -> 224, long edge 160 -> int(224*160/96) = 373, crop offset round((373-224)/2) = 74
    full = torch.nn.functional.interpolate(fr, size=(224, 373), mode="bicubic", antialias=True)
    assert torch.equal(b, (full[..., 74:74 + 224].float() / 255.0 - 0.5) / 0.5)
    const = torch.full((1, 3, 50, 70), 200, dtype=torch.uint8)
    assert torch.allclose(E.synchformer_preprocess(const), torch.full((1, 3, 224, 224), (200 / 255 - 0.5) / 0.5))
    same = torch.randint(0, 256, (2, 3, 224, 224), generator=g, dtype=torch.uint8)
    assert torch.equal(E.synchformer_preprocess(same), (same.float() / 255 - 0.5) / 0.5)
    tall = torch.randint(0, 256, (1, 3, 300, 224), generator=g, dtype=torch.uint8)     # portrait: crop along H
    assert torch.equal(E.synchformer_preprocess(tall), (tall[..., 38:38 + 224, :].float() / 255 - 0.5) / 0.5)


def test_synchformer_matches_reference_golden():
    """host/encoders.py::synchformer_segments vs the reference's MotionFormer (divided space-time attention x12,
    LayerNorm, spatial aggregation layer) on two overlapping segments - golden g11, generated by running
    the reference's own module with the same synthesised weights."""
    g = golden("g11_v2a")
    sd = synth.materialize(E.synchformer_schema())
    frames = synth.synth_tensor("g11.frames", (24, 3, 224, 224), 0.55)
    with torch.inference_mode():
        feat = E.encode_video_with_sync(sd, frames)
    assert feat.shape == (1, 16, 768)
    assert rel_err(feat, g["sync_feat"]) < 2e-5
    with pytest.raises(ValueError):
        E.encode_video_with_sync(sd, frames[:15])                      # fewer than one 16-frame segment
    with pytest.raises(ValueError):
        E.load_synchformer_state({"afeat_extractor.x": torch.zeros(1)}, "cpu", torch.float32)


def test_video_features_shapes_and_text(monkeypatch):
    """_video_features end to end on CPU (1 s clip): SigLIP2 features [1, 8, 768], Synchformer features
    [1, 16, 768] = the (Lv, Ls) the DiT expects for 1 s (config.lengths), audio length from the 25 fps stream,
    [negative, positive] text order."""
    tok, clap = tiny_clap()
    sync_full = synth.materialize(E.synchformer_schema())
    sync_full["afeat_extractor.dummy.weight"] = torch.zeros(4)          # checkpoint carries more than the visual branch
    deps = nodes.AttributeDict(siglip2_model=tiny_siglip(), syncformer_model=sync_full, clap_tokenizer=tok, clap_model=clap)
    g = torch.Generator().manual_seed(3)
    image = torch.rand(20, 72, 96, 3, generator=g)                      # 20 frames at 16 fps -> 1.25 s of video
    visual, text, alen = nodes.HunyuanFoleySampler._video_features(image, 1.0, 16.0, "a dog barks", "noise", deps,
                                                                   torch.device("cpu"), torch.float32)
    la, lv, ls = C.lengths(1.0)
    assert visual["siglip2_feat"].shape == (1, lv, 768) and visual["syncformer_feat"].shape == (1, ls, 768)
    assert alen == 1.0
    assert text["text_feat"].shape[0] == 1 and text["text_feat"].shape[2] == 768
    assert text["text_feat"].shape[1] == len("a dog barks") + 2           # padded to the longer prompt
    assert all(k.startswith("vfeat_extractor.") for k in deps["syncformer_model"])
    assert torch.isfinite(visual["siglip2_feat"]).all() and torch.isfinite(visual["syncformer_feat"]).all()
    # the positive prompt is row 1 of the [negative, positive] batch
    res = E.encode_text_feat(tok, clap, ["noise", "a dog barks"], torch.device("cpu"))
    assert torch.equal(text["text_feat"], res[1:]) and torch.equal(text["uncond_text_feat"], res[:1])
    # feature lengths for the 5 s headline clip follow the same rule: [1,40,768] / [1,112,768]
    assert C.lengths(5.0)[1:] == (40, 8 * ((125 - E.SYNC_SEGMENT) // E.SYNC_STRIDE + 1)) == (40, 112)


def test_encoder_state_cache_follows_the_weights():
    """The device copy of an HF encoder's state dict (what the HIP engine stages its matrices from) is cached on the module
    with a signature of the weights it was taken from: same weights -> same dict object (the engine keeps its staged
    matrices), weights edited in place or replaced -> a fresh copy, `release_encoder_caches` drops it."""
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.LayerNorm(3))
    keep = lambda k, v: v.is_floating_point()
    a = E._cached_state(m, "_foley_text_sd", "cpu", keep)
    assert E._cached_state(m, "_foley_text_sd", "cpu", keep) is a and torch.equal(a["0.weight"], m[0].weight)
    with torch.no_grad():
        m[0].weight.add_(1.0)                                  # in-place edit: version counter moves
    b = E._cached_state(m, "_foley_text_sd", "cpu", keep)
    assert b is not a and torch.equal(b["0.weight"], m[0].weight)
    m[0].weight = torch.nn.Parameter(torch.zeros(3, 4))        # reloaded weights: new storage
    c = E._cached_state(m, "_foley_text_sd", "cpu", keep)
    assert c is not b and float(c["0.weight"].abs().sum()) == 0.0
    E.release_encoder_caches(m)
    assert not hasattr(m, "_foley_text_sd")
