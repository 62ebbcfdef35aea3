"""Parameter tree of the SD-2-base ``UNet2DConditionModel`` with diffusers' attribute and
state-dict names, WITHOUT any forward code.

The denoiser only needs the weights of the UNets it is given (it sequences its own kernels,
like the reference sequences the sub-modules: models/pano/MVGenModel.py:98-294).  When diffusers
itself is installed, pass its ``UNet2DConditionModel`` straight to ``MultiViewBaseModel``; this
container exists for environments without diffusers (load a diffusers ``state_dict`` into it, or
fill it with synthetic weights for benchmarking).
"""
import math

import torch
import torch.nn as nn

SD2_BASE = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                layers_per_block=2, num_heads=(5, 10, 20, 20), cross_attention_dim=1024,
                norm_num_groups=32, cross_attn_blocks=(True, True, True, False))


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("parameter container only: the HIP denoiser (MultiViewBaseModel) runs the kernels")


class _LoRALinear(nn.Linear):
    """Linear that may carry ``lora_layer.{down,up}`` (diffusers LoRACompatibleLinear layout)."""

    def __init__(self, i, o, bias=True):
        super().__init__(i, o, bias=bias)
        self.lora_layer = None

    def set_lora(self, rank):
        self.lora_layer = _Holder()
        self.lora_layer.down = nn.Linear(self.in_features, rank, bias=False)
        self.lora_layer.up = nn.Linear(rank, self.out_features, bias=False)


def _resnet(cin, cout, temb, groups):
    r = _Holder()
    r.in_channels, r.out_channels = cin, cout
    r.norm1 = nn.GroupNorm(groups, cin, eps=1e-5)
    r.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
    r.time_emb_proj = nn.Linear(temb, cout)
    r.norm2 = nn.GroupNorm(groups, cout, eps=1e-5)
    r.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
    r.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None
    return r


def _attention(dim, ctx, heads):
    a = _Holder()
    a.heads = heads
    a.to_q = _LoRALinear(dim, dim, bias=False)
    a.to_k = _LoRALinear(ctx, dim, bias=False)
    a.to_v = _LoRALinear(ctx, dim, bias=False)
    a.to_out = nn.ModuleList([_LoRALinear(dim, dim), nn.Dropout(0.0)])
    return a


def _transformer(ch, heads, ctx, groups):
    t = _Holder()
    t.norm = nn.GroupNorm(groups, ch, eps=1e-6)
    t.proj_in = nn.Linear(ch, ch)
    blk = _Holder()
    blk.norm1, blk.norm2, blk.norm3 = nn.LayerNorm(ch), nn.LayerNorm(ch), nn.LayerNorm(ch)
    blk.attn1 = _attention(ch, ch, heads)
    blk.attn2 = _attention(ch, ctx, heads)
    ff = _Holder()
    geglu = _Holder()
    geglu.proj = nn.Linear(ch, ch * 8)
    ff.net = nn.ModuleList([geglu, nn.Dropout(0.0), nn.Linear(ch * 4, ch)])
    blk.ff = ff
    t.transformer_blocks = nn.ModuleList([blk])
    t.proj_out = nn.Linear(ch, ch)
    return t


def _sampler(ch, stride):
    s = _Holder()
    s.channels = s.out_channels = ch
    s.conv = nn.Conv2d(ch, ch, 3, stride=stride, padding=1)
    return s


class UNetParams(_Holder):
    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, num_heads=(5, 10, 20, 20), cross_attention_dim=1024,
                 norm_num_groups=32, cross_attn_blocks=(True, True, True, False)):
        super().__init__()
        boc, g, ctx = tuple(block_out_channels), norm_num_groups, cross_attention_dim
        temb = boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = _Holder()
        self.time_embedding.linear_1 = nn.Linear(boc[0], temb)
        self.time_embedding.linear_2 = nn.Linear(temb, temb)

        def block(cross):
            b = _Holder()
            b.has_cross_attention = cross
            b.resnets = nn.ModuleList()
            if cross:
                b.attentions = nn.ModuleList()
            b.downsamplers = b.upsamplers = None
            return b

        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i, oc in enumerate(boc):
            b = block(cross_attn_blocks[i])
            for j in range(layers_per_block):
                b.resnets.append(_resnet(ch if j == 0 else oc, oc, temb, g))
                if b.has_cross_attention:
                    b.attentions.append(_transformer(oc, num_heads[i], ctx, g))
            if i != len(boc) - 1:
                b.downsamplers = nn.ModuleList([_sampler(oc, 2)])
            self.down_blocks.append(b)
            ch = oc
        self.mid_block = block(True)
        self.mid_block.resnets.append(_resnet(boc[-1], boc[-1], temb, g))
        self.mid_block.attentions.append(_transformer(boc[-1], num_heads[-1], ctx, g))
        self.mid_block.resnets.append(_resnet(boc[-1], boc[-1], temb, g))
        self.up_blocks = nn.ModuleList()
        rev, rheads, rcross = boc[::-1], tuple(num_heads)[::-1], tuple(cross_attn_blocks)[::-1]
        prev = rev[0]
        for i, oc in enumerate(rev):
            tail_skip = rev[min(i + 1, len(rev) - 1)]
            b = block(rcross[i])
            for j in range(layers_per_block + 1):
                skip = tail_skip if j == layers_per_block else oc
                b.resnets.append(_resnet((prev if j == 0 else oc) + skip, oc, temb, g))
                if b.has_cross_attention:
                    b.attentions.append(_transformer(oc, rheads[i], ctx, g))
            if i != len(rev) - 1:
                b.upsamplers = nn.ModuleList([_sampler(oc, 1)])
            self.up_blocks.append(b)
            prev = oc
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    def attention_modules(self):
        for blk in [*self.down_blocks, self.mid_block, *self.up_blocks]:
            for t in getattr(blk, "attentions", []) or []:
                yield t.transformer_blocks[0].attn1
                yield t.transformer_blocks[0].attn2

    def add_lora(self, rank=4, layout="lora_layer"):
        """Rank-`rank` LoRA on the attention projections (q, k, v, out), PanoGenerator.py:132-151.
        layout "lora_layer": ``<linear>.lora_layer.{down,up}`` (diffusers' post-migration / saved layout);
        layout "processor":  ``<attn>.processor.to_{q,k,v,out}_lora.{down,up}`` -- what
        ``unet.set_attn_processor(LoRAAttnProcessor(...))`` creates and what the keys of a reference
        checkpoint address after ``convert_state_dict`` (PanoGenerator.py:101-111)."""
        for a in self.attention_modules():
            if layout == "lora_layer":
                for lin in (a.to_q, a.to_k, a.to_v, a.to_out[0]):
                    lin.set_lora(rank)
            elif layout == "processor":
                proc = _Holder()
                for name, lin in (("to_q_lora", a.to_q), ("to_k_lora", a.to_k), ("to_v_lora", a.to_v),
                                  ("to_out_lora", a.to_out[0])):
                    lora = _Holder()
                    lora.down = nn.Linear(lin.in_features, rank, bias=False)
                    lora.up = nn.Linear(rank, lin.out_features, bias=False)
                    nn.init.zeros_(lora.up.weight)          # diffusers LoRALinearLayer: up starts at zero
                    setattr(proc, name, lora)
                a.processor = proc
            else:
                raise ValueError("layout must be 'lora_layer' or 'processor'")


class ControlNetParams(_Holder):
    """Parameter tree of diffusers ``ControlNetModel.from_unet(unet)`` (PanoGenerator.py:153-157):
    encoder half of the UNet + conditioning embedding + 12 + 1 zero-convs, diffusers names."""

    def __init__(self, conditioning_embedding_out_channels=(16, 32, 96, 256), **unet_cfg):
        super().__init__()
        enc = UNetParams(**unet_cfg)
        self.conv_in, self.time_embedding = enc.conv_in, enc.time_embedding
        self.down_blocks, self.mid_block = enc.down_blocks, enc.mid_block
        boc = tuple(unet_cfg.get("block_out_channels", SD2_BASE["block_out_channels"]))
        lpb = unet_cfg.get("layers_per_block", 2)
        ce = _Holder()
        coc = tuple(conditioning_embedding_out_channels)
        ce.conv_in = nn.Conv2d(3, coc[0], 3, padding=1)
        ce.blocks = nn.ModuleList()
        for i in range(len(coc) - 1):
            ce.blocks.append(nn.Conv2d(coc[i], coc[i], 3, padding=1))
            ce.blocks.append(nn.Conv2d(coc[i], coc[i + 1], 3, padding=1, stride=2))
        ce.conv_out = nn.Conv2d(coc[-1], boc[0], 3, padding=1)
        self.controlnet_cond_embedding = ce
        self.controlnet_down_blocks = nn.ModuleList([nn.Conv2d(boc[0], boc[0], 1)])
        for i, oc in enumerate(boc):
            for _ in range(lpb):
                self.controlnet_down_blocks.append(nn.Conv2d(oc, oc, 1))
            if i != len(boc) - 1:
                self.controlnet_down_blocks.append(nn.Conv2d(oc, oc, 1))
        self.controlnet_mid_block = nn.Conv2d(boc[-1], boc[-1], 1)


@torch.no_grad()
def fill_synthetic(module, seed, device=None):
    """Seeded synthetic weights generated ON the module's device (fan-in scaled normals, norm gains
    ~1, small biases).  For benchmarking only -- no SD-2 checkpoint is reachable offline."""
    dev = device or next(module.parameters()).device
    gen = torch.Generator(device=dev).manual_seed(seed)
    for name, p in module.named_parameters():
        if p.dim() >= 2:
            std = (0.3 if "lora_layer" in name else 1.0) / math.sqrt(p[0].numel())
            p.copy_(torch.randn(p.shape, generator=gen, device=dev) * std)
        elif name.endswith("weight"):
            p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=gen, device=dev))
        else:
            p.copy_(0.05 * torch.randn(p.shape, generator=gen, device=dev))
    return module
