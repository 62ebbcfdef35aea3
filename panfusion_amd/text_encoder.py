"""CLIP text encoding of the prompts on the HIP kernels -- the step right before the denoising loop (SURVEY.md
§8f row 2; reference ``models/pano/PanoGenerator.py:197-211`` ``encode_text``, ``PanFusion.py:45-62``
``embed_prompt``, ``:134-138`` the null prompt of the CFG pair).

``TextEncoder(clip_text_model)`` takes a module tree with transformers' ``CLIPTextModel`` attribute names (the
object the reference loads, PanoGenerator.py:116-118; both the 4.x layout ``.text_model.{embeddings, encoder,
final_layer_norm}`` and the flattened 5.x one), repacks the weights once and runs ``last_hidden_state`` as C-ABI
calls: embedding gather, then per layer LayerNorm -> q / k / v projections -> causal self-attention on the
flash kernel (heads of 64; the causal mask and the padding of the 77 tokens to 80 keys travel as its additive bias
table) -> output projection + residual -> LayerNorm -> fc1 + GELU -> fc2 + residual, and the final LayerNorm.
Tokenisation is host string processing and stays with the caller's ``CLIPTokenizer``.

Weight packing notes: the value bias moves behind the output projection (softmax rows sum to one: P (V + 1 b^T) =
P V + 1 b^T); GELU(fc1) runs on the GEMM's fused GEGLU epilogue with a constant-one value row per gate row
(value weights 0, value bias 1: value * gelu(gate) = gelu(gate)).
"""
from types import SimpleNamespace as NS

import torch

from . import engine, ops


def _text_model(m):
    return getattr(m, "text_model", m)


def pack_text_encoder(model, dev, dtype, mixed):
    tm = _text_model(model)
    emb = tm.embeddings
    p = NS(dtype=dtype, stream=torch.float32 if mixed else dtype)
    p.tok, p.pos = engine._f32(emb.token_embedding.weight, dev), engine._f32(emb.position_embedding.weight, dev)
    p.C = p.tok.shape[1]
    p.layers = []
    for layer in tm.encoder.layers:
        a = layer.self_attn
        L = NS()
        L.heads = getattr(a, "num_heads", None) or model.config.num_attention_heads
        L.ln1, L.ln2 = engine._norm(layer.layer_norm1, dev), engine._norm(layer.layer_norm2, dev)
        L.wq, L.bq = engine._w16(a.q_proj.weight, dev, dtype), engine._bias(a.q_proj, dev)
        L.wk, L.bk = engine._w16(a.k_proj.weight, dev, dtype), engine._bias(a.k_proj, dev)
        L.wv = engine._w16(a.v_proj.weight, dev, dtype)
        wo = a.out_proj.weight.detach().float()
        L.wo = engine._w16(wo, dev, dtype)
        L.bo = (a.out_proj.bias.detach().float() + wo @ a.v_proj.bias.detach().float()).to(dev).contiguous()
        w1, b1 = layer.mlp.fc1.weight.detach().float(), layer.mlp.fc1.bias.detach().float()
        inter = w1.shape[0]
        wg = torch.zeros(2 * inter, w1.shape[1])
        wg[1::2] = w1                                      # rows (value_j = 0, gate_j = fc1 row j)
        bg = torch.ones(2 * inter)
        bg[1::2] = b1                                      # value bias 1: 1 * gelu(gate)
        L.w1, L.b1 = engine._w16(wg, dev, dtype), bg.to(dev).contiguous()
        L.w2, L.b2 = engine._w16(layer.mlp.fc2.weight, dev, dtype), engine._bias(layer.mlp.fc2, dev)
        p.layers.append(L)
    p.ln_f = engine._norm(tm.final_layer_norm, dev)
    act = getattr(model.config, "hidden_act", "gelu")
    if act != "gelu":
        raise NotImplementedError("text encoder activation %r (the SD-2 OpenCLIP-H text model uses exact GELU)" % act)
    p.masks = {}
    return p


def _causal_bias(p, L, Lp, dev):
    """Additive bias [Lp, Lp] of the causal mask over L real tokens (key j visible to query i iff j <= i and j < L;
    padding queries see key 0 so that no row is fully masked) + the all-ones 32x32 tile map the kernel wants."""
    key = (L, Lp, str(dev))
    if key not in p.masks:
        i = torch.arange(Lp)[:, None]
        j = torch.arange(Lp)[None, :]
        vis = (j <= i) & (j < L)
        vis[L:, 0] = True
        bias = torch.where(vis, 0.0, -1.0e4).float().contiguous().to(dev)
        flags = torch.ones((Lp + 31) // 32, (Lp + 31) // 32, dtype=torch.uint8, device=dev)
        p.masks[key] = (bias, flags)
    return p.masks[key]


class TextEncoder:
    def __init__(self, model, compute_dtype=torch.float16, precision=None):
        self.model, self.compute_dtype = model, compute_dtype
        self.precision = precision or engine.default_precision(compute_dtype)
        self._packed = {}

    def packed(self, device):
        key = (str(device), self.compute_dtype, self.precision)
        if key not in self._packed:
            self._packed[key] = pack_text_encoder(self.model, device, self.compute_dtype, self.precision == "mixed")
        return self._packed[key]

    def repack(self):
        self._packed.clear()

    @torch.no_grad()
    def encode_ids(self, input_ids):
        """``text_encoder(input_ids)[0]`` (last_hidden_state): int64 [B, L] on the GPU -> fp32 [B, L, C]."""
        p = self.packed(input_ids.device)
        B, L = input_ids.shape
        Lp = (L + 3) // 4 * 4
        dt, Cc = p.dtype, p.C
        x = ops.embed_tokens(input_ids, p.tok, p.pos, Lp, p.stream).view(B * Lp, Cc)
        bias, flags = _causal_bias(p, L, Lp, input_ids.device)
        for Ly in p.layers:
            D = Cc // Ly.heads
            ln = ops.layernorm(x, Ly.ln1.g, Ly.ln1.b, Ly.ln1.eps, out_dtype=dt)
            q = ops.linear(ln, Ly.wq, bias=Ly.bq)
            k = ops.linear(ln, Ly.wk, bias=Ly.bk)
            vt = ops.linear_t(ln.view(B, Lp, Cc), Ly.wv)                              # [B, C, ld]
            o = ops.attention(q, k, vt, B, Ly.heads, D, Lp, Lp, q_ld=Cc, k_ld=Cc, vt_ld=vt.shape[-1], q_bs=Lp * Cc,
                              k_bs=Lp * Cc, vt_bs=vt.shape[1] * vt.shape[2], bias=bias, flags=flags)
            x = ops.linear(o.view(B * Lp, Cc), Ly.wo, bias=Ly.bo, residual=x)
            ln = ops.layernorm(x, Ly.ln2.g, Ly.ln2.b, Ly.ln2.eps, out_dtype=dt)
            h = ops.linear(ln, Ly.w1, bias=Ly.b1, geglu=True)                         # gelu(fc1(ln))
            x = ops.linear(h, Ly.w2, bias=Ly.b2, residual=x)
        # final LayerNorm: the 16-bit kernel output is the only rounding of the result; callers get fp32 like the
        # reference's `.to(self.dtype)` (PanoGenerator.py:211)
        y = ops.layernorm(x, p.ln_f.g, p.ln_f.b, p.ln_f.eps, out_dtype=dt)
        return y.view(B, Lp, Cc)[:, :L].float()

    def encode_text(self, tokenizer, text):
        """``PanoGenerator.encode_text`` (PanoGenerator.py:197-211) with the caller's CLIPTokenizer."""
        t = tokenizer(text, padding="max_length", max_length=tokenizer.model_max_length, truncation=True, return_tensors="pt")
        dev = next(iter(self._packed.values())).tok.device if self._packed else torch.device("cuda")
        return self.encode_ids(t.input_ids.to(dev))


def embed_prompt(encoder, tokenizer, pers_prompts, pano_prompt, num_cameras):
    """``PanFusion.embed_prompt`` + the null prompt of ``inference`` (PanFusion.py:45-62,134-138): per-view prompts
    (a list of b * m strings, or '' for none), the panorama prompt, and '' for the unconditional half ->
    (pers_prompt_embd [2b, m, L, C], pano_prompt_embd [2b, 1, L, C]) = [null ; prompt] like the reference's cat."""
    if pers_prompts:
        pe = encoder.encode_text(tokenizer, pers_prompts)
        pe = pe.unflatten(0, (-1, num_cameras))
    else:
        pe = encoder.encode_text(tokenizer, "")[:, None].repeat(1, num_cameras, 1, 1)
    pano = encoder.encode_text(tokenizer, pano_prompt or "")[:, None]
    null = encoder.encode_text(tokenizer, "")[:, None]
    return torch.cat([null.repeat(pe.shape[0], num_cameras, 1, 1), pe]), torch.cat([null.repeat(pano.shape[0], 1, 1, 1), pano])
