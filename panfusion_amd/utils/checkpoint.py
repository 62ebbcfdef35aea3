"""Checkpoint key plumbing of the reference's LightningModule (``models/pano/PanoGenerator.py:85-111``) for
the HIP denoiser: the same ``convert_state_dict`` renaming, and a loader that fits a reference checkpoint
(``mv_base_model.{unet,pano_unet}._orig_mod.<diffusers names>``, LoRA under either of diffusers' two layouts,
EPA blocks under ``mv_base_model.cp_blocks_*``) onto whatever module tree ``MultiViewBaseModel`` was given.
"""
import re

_LORA_PAIRS = (("to_q.lora_layer", "processor.to_q_lora"), ("to_k.lora_layer", "processor.to_k_lora"),
               ("to_v.lora_layer", "processor.to_v_lora"), ("to_out.0.lora_layer", "processor.to_out_lora"))


def convert_state_dict(state_dict):
    """In-place rename ``<attn>.to_q.lora_layer.* -> <attn>.processor.to_q_lora.*`` (and k, v, out), exactly the
    reference's ``PanoGenerator.convert_state_dict`` (PanoGenerator.py:101-107): checkpoints are SAVED with the
    LoRA matrices where diffusers 0.24 migrated them, and LOADED into a model whose LoRAAttnProcessor still holds them."""
    for old_k in list(state_dict.keys()):
        new_k = old_k
        for a, b in _LORA_PAIRS:
            new_k = new_k.replace(a, b)
        state_dict[new_k] = state_dict.pop(old_k)
    return state_dict


def _variants(key):
    """The spellings one tensor can have: with / without torch.compile's ``_orig_mod.`` after the branch name,
    LoRA in the linear or in the processor."""
    out = {key}
    for k in list(out):
        out.add(re.sub(r"^((?:pano_)?unet|(?:pers|pano)_cn)\._orig_mod\.", r"\1.", k))
        out.add(re.sub(r"^((?:pano_)?unet|(?:pers|pano)_cn)\.(?!_orig_mod\.)", r"\1._orig_mod.", k))
    for k in list(out):
        for a, b in _LORA_PAIRS:
            out.add(k.replace(a, b))
            out.add(k.replace(b, a))
    return out


def load_reference_state_dict(model, state_dict, prefix="mv_base_model.", strict=True):
    """Load a reference checkpoint's ``state_dict`` into a ``MultiViewBaseModel``.  Keys outside ``prefix`` (text
    encoder, VAE, eval metrics) are ignored; every remaining key must find its tensor in the model under one of
    its spellings (``_orig_mod.`` or not, LoRA layout) and -- with ``strict`` -- every model tensor must be fed."""
    own = model.state_dict()
    feed, unknown = {}, []
    for k, v in state_dict.items():
        if not k.startswith(prefix):
            continue
        k = k[len(prefix):]
        hit = [c for c in _variants(k) if c in own]
        if len(hit) != 1:
            unknown.append(k)
            continue
        feed[hit[0]] = v
    missing = [k for k in own if k not in feed]
    if unknown or (strict and missing):
        raise KeyError("checkpoint does not fit the model: %d unexpected (e.g. %s), %d missing (e.g. %s)"
                       % (len(unknown), unknown[:3], len(missing), missing[:3]))
    result = model.load_state_dict(feed, strict=False)
    model.repack()
    return result
