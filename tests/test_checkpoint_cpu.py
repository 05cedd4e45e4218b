"""Checkpoint key compatibility with the reference trainer (SURVEY.md section 8b state-dict keys, 8f rank 3).  CPU only."""
from types import SimpleNamespace

import torch

from titok_video_amd import checkpoint as CK
from titok_video_amd.model.losses import ReconstructionLoss
from titok_video_amd.model.titok import TiTok
from titok_video_amd.synthetic import seeded_titok_state, seeded_tower_state


def _cfg():
    return SimpleNamespace(
        tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny"),
                                  losses=SimpleNamespace(disc_weight=0.4, perceptual_weight=0.0, gram_weight=0.0, perceptual_samples_per_step=24,
                                                         perceptual_sampling_size=128)),
        discriminator=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], model_size="tiny"),
                                      losses=SimpleNamespace(gp_weight=0.1, gp_noise=0.1, centering_weight=0.01)),
        training=SimpleNamespace(main=SimpleNamespace(torch_compile=False, max_steps=10)))


def _reference_shaped_checkpoint():
    """A trainer state dict as the reference writes it: tokenizer under 'model.', discriminator under 'loss_module.disc_model.',
    plus the kind of entries its state_dict() filter drops (train.py:218-220)."""
    sd = {"model." + k: v for k, v in seeded_titok_state(3).items()}
    sd.update({"loss_module.disc_model." + k: v for k, v in seeded_tower_state("encoder", "tiny", (4, 8, 8), 3, 1, seed=5).items()})
    sd["loss_module.perceptual_model.net.slice1.0.weight"] = torch.zeros(3)
    sd["eval_metrics.fvd.sum"] = torch.zeros(1)
    return sd


def test_reference_trainer_state_dict_loads_strict_and_round_trips(tmp_path):
    cfg = _cfg()
    model, lm = TiTok(cfg), ReconstructionLoss(cfg)
    sd = _reference_shaped_checkpoint()
    CK.load_trainer_state_dict(sd, model, lm, strict=True)                 # every key consumed, none missing
    out = CK.trainer_state_dict(model, lm)
    kept = {k: v for k, v in sd.items() if "perceptual_model" not in k and "eval_metrics" not in k}
    assert list(out.keys()) == list(kept.keys()) or set(out.keys()) == set(kept.keys())
    for k, v in kept.items():
        assert torch.equal(out[k], v), k
    assert len([k for k in out if k.startswith("model.")]) == len(seeded_titok_state(3))
    assert sum(v.numel() for k, v in out.items() if k.startswith("model.")) == 6_828_295      # tiny/tiny tokenizer
    path = str(tmp_path / "ck.pt")
    CK.save_checkpoint(path, model, lm, global_step=123)
    model2, lm2 = TiTok(cfg), ReconstructionLoss(cfg)
    assert CK.load_checkpoint(path, model2, lm2) == 123
    for (k, a), (_, b) in zip(CK.trainer_state_dict(model, lm).items(), CK.trainer_state_dict(model2, lm2).items()):
        assert torch.equal(a, b), k


def test_bare_tokenizer_state_dict_and_partial_checkpoint():
    cfg = _cfg()
    model = TiTok(cfg)
    CK.load_trainer_state_dict(seeded_titok_state(4), model, strict=True)   # no prefixes: plain TiTok state dict
    sd = {k: v for k, v in _reference_shaped_checkpoint().items() if k.startswith("model.")}
    lm = ReconstructionLoss(cfg)
    before = {k: v.clone() for k, v in lm.state_dict().items()}
    CK.load_trainer_state_dict(sd, model, lm, strict=False)                 # init_from_checkpoint semantics (strict=False)
    for k, v in lm.state_dict().items():
        assert torch.equal(v, before[k])
