"""Dual-branch denoiser with the reference's constructor, attributes and
forward signature (reference ``models/pano/MVGenModel.py:9-297``), executed on
the HIP kernels.

``MultiViewBaseModel(unet, pano_unet, pers_cn=None, pano_cn=None, pano_pad=True)``
accepts the same diffusers ``UNet2DConditionModel`` objects the reference
passes (``PanFusion.instantiate_model``, PanFusion.py:23-28; ``unet=None`` is the
PanoOnly shape, PanoOnly.py:13).  Their weights are repacked once per device
into kernel layouts; ``forward`` then sequences C-ABI kernel calls and never
touches the modules' own ``forward``.
"""
import contextlib
import os

import torch
import torch.nn as nn

from ... import engine
from .modules import WarpAttn, camera_groups


def _repack_hook(module, incompatible):
    """load_state_dict post hook (module-level: a lambda would make the module unpicklable -- torch.save(model),
    mp.spawn / Lightning ddp_spawn passing the module)."""
    module.repack()


class MultiViewBaseModel(nn.Module):
    def __init__(self, unet, pano_unet, pers_cn=None, pano_cn=None, pano_pad=True,
                 compute_dtype=torch.float16, precision=None, differentiable=False):
        """compute_dtype: the 16-bit MFMA operand type; precision: "mixed" (fp32 residual streams +
        split-precision stream-path GEMMs: the scheme that meets the 1e-3 parity bar, default for fp16)
        or "fast" (everything 16-bit); None -> engine.default_precision (PF_PRECISION overrides).
        differentiable: with autograd enabled, forward() returns outputs that back-propagate into the EPA blocks and
        the LoRA matrices of both UNets (the reference's training_step, PanFusion.py:64-98) -- same forward kernels,
        backward in train_engine.py.  Under torch.no_grad() (validation, predict) nothing changes."""
        super().__init__()
        self.differentiable = differentiable
        self.precision = precision or engine.default_precision(compute_dtype)
        self.unet = unet
        self.pano_unet = pano_unet
        self.pers_cn = pers_cn
        self.pano_cn = pano_cn
        self.pano_pad = pano_pad
        self.compute_dtype = compute_dtype
        self._packed = {}
        # the panorama branch runs on its own HIP stream between EPA fusions (its layers are too small
        # to fill 256 CUs; side by side with the view branch they fill each other's tails)
        self.two_streams = os.environ.get("PF_STREAMS", "2") != "1"
        self._side = None
        # packed 16-bit weights go stale when a checkpoint is loaded -- also through a parent module (the
        # reference's LightningModule), where an overridden load_state_dict would never run
        self.register_load_state_dict_post_hook(_repack_hook)

        if self.unet is not None:      # EPA block widths, reference MVGenModel.py:19-32
            self.cp_blocks_encoder = nn.ModuleList(
                [WarpAttn(blk.downsamplers[-1].out_channels, compute_dtype, self.precision)
                 for blk in self.unet.down_blocks if blk.downsamplers is not None])
            self.cp_blocks_mid = WarpAttn(self.unet.mid_block.resnets[-1].out_channels, compute_dtype, self.precision)
            self.cp_blocks_decoder = nn.ModuleList(
                [WarpAttn(blk.upsamplers[0].channels, compute_dtype, self.precision)
                 for blk in self.unet.up_blocks if blk.upsamplers is not None])
            self.trainable_parameters = [(list(self.cp_blocks_mid.parameters())
                                          + list(self.cp_blocks_decoder.parameters())
                                          + list(self.cp_blocks_encoder.parameters()), 1.0)]

    # ------------------------------------------------------------------ weights
    def packed(self, which, device):
        key = (which, str(device), self.compute_dtype, self.precision)
        if which.endswith("_cn") and key in self._packed and self._packed[key].param_key != self._cn_versions(which):
            del self._packed[key]                   # a ControlNet that trains: its weights moved since the pack was built
        if key not in self._packed:
            pack = engine.pack_controlnet if which.endswith("_cn") else engine.pack_unet
            self._packed[key] = pack(getattr(self, which), device, self.compute_dtype, self.precision == "mixed")
            # the LoRA state this pack was folded from (refold_lora compares against it; ADVICE r2: without it the
            # first call after an in-place parameter change saw "no change")
            self._packed[key].lora_key = self._lora_versions()
            if which.endswith("_cn"):
                self._packed[key].param_key = self._cn_versions(which)
        return self._packed[key]

    def _cn_versions(self, which):
        """Version counters of a ControlNet's parameters (optimizer steps and in-place copies bump them): the whole pack is
        rebuilt when one moved -- every weight of a training ControlNet changes in a step, there is nothing to re-fold."""
        return tuple(t._version for t in getattr(self, which).parameters())

    def repack(self):
        """Drop the packed 16-bit weights (call after changing parameters / loading a checkpoint)."""
        self._packed.clear()
        self.__dict__.pop("_trainable_cache", None)
        if self.unet is not None:
            for blk in [*self.cp_blocks_encoder, self.cp_blocks_mid, *self.cp_blocks_decoder]:
                blk.compute_dtype, blk.precision = self.compute_dtype, self.precision
                blk.repack()

    def _prompt16(self, prompt):
        """(b, m, L, D) prompt embeddings -> (b*m, L, D) in the 16-bit operand type.  The cast re-read 25 MB per step
        for data that does not change inside a sampling loop: done once per (storage, version) of the caller's tensor."""
        key = (prompt.data_ptr(), prompt._version, tuple(prompt.shape), prompt.dtype, str(prompt.device), self.compute_dtype)
        cache = self.__dict__.setdefault("_prompt_cache", {})
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 8:
                cache.pop(next(iter(cache)))
            hit = (prompt.flatten(0, 1).to(self.compute_dtype).contiguous(), prompt)     # (cast, keep-alive of the source)
            cache[key] = hit
        return hit[0]

    # ------------------------------------------------------------------ forward
    def trainable_tensors(self):
        """EPA parameters and the LoRA matrices of both UNets: what the reference optimises (PanoGenerator.py:129-160,
        MVGenModel.py:34-36), in the order DenoiserFunction returns their gradients."""
        from ... import train_engine
        # cache keyed on the LoRA modules found in the tree (ids): attaching / replacing LoRA layers after the first call
        # (set_attn_processor, lora_layer assignment) rebuilds the list instead of silently training the old tensors
        fp = self._lora_fingerprint()
        cached = getattr(self, "_trainable_cache", None)
        if cached is not None and cached[0] == fp:
            return cached[1]
        out, seen = [], set()

        def add(t):
            if id(t) not in seen:
                seen.add(id(t))
                out.append(t)
        if self.unet is not None:
            for blk in [*self.cp_blocks_encoder, self.cp_blocks_mid, *self.cp_blocks_decoder]:
                for t in blk.transformer.parameters():
                    add(t)
        self._n_epa = len(out)
        for unet in (self.unet, self.pano_unet):
            if unet is None:
                continue
            for mod in unet.modules():
                if all(hasattr(mod, a) for a in ("to_q", "to_k", "to_v", "to_out")):
                    for name, lin in (("to_q", mod.to_q), ("to_k", mod.to_k), ("to_v", mod.to_v), ("to_out", mod.to_out[0])):
                        ref = train_engine.lora_of(mod, lin, name + "_lora")
                        if ref is not None:
                            add(ref.down)
                            add(ref.up)
        # the ControlNets: all of their parameters (PanoGenerator.py:153-157 get_cn -> list(cn.parameters())); which of them
        # actually train is the caller's requires_grad (the reference freezes nothing of a ControlNet it added)
        self._n_lora_epa = len(out)
        for cn in (self.pers_cn, self.pano_cn):
            if cn is not None:
                for t in cn.parameters():
                    add(t)
        self._trainable_cache = (fp, out)
        return out

    def _attn_projections(self):
        """(attention module, name, linear) of every attention projection of both UNets; the set of attention MODULES is
        fixed by the UNet architecture, only their lora_layer / processor attributes can change: listed once."""
        hit = self.__dict__.get("_attn_proj_cache")
        if hit is None:
            hit = []
            for unet in (self.unet, self.pano_unet):
                if unet is None:
                    continue
                for mod in unet.modules():
                    if all(hasattr(mod, a) for a in ("to_q", "to_k", "to_v", "to_out")):
                        hit += [(mod, "to_q", mod.to_q), (mod, "to_k", mod.to_k), (mod, "to_v", mod.to_v), (mod, "to_out", mod.to_out[0])]
            self.__dict__["_attn_proj_cache"] = hit
        return hit

    def _lora_fingerprint(self):
        """ids of the LoRA modules currently attached (either diffusers location)."""
        fp = []
        for mod, name, lin in self._attn_projections():
            lora = getattr(lin, "lora_layer", None) or engine._processor_lora(mod, name + "_lora")
            fp.append(id(lora) if lora is not None else 0)
        return tuple(fp)

    def _lora_versions(self):
        # the LoRA matrices only: the EPA blocks track their own parameters (WarpAttn.packed), the ControlNets theirs (packed()) --
        # a layout-conditioned run (EPA + ControlNet train, LoRA frozen) must not re-fold 256 unchanged projections per step
        tensors = self.trainable_tensors()
        return tuple((id(t), t._version) for t in tensors[self._n_epa:self._n_lora_epa])

    def forward(self, latents, pano_latent, timestep, prompt_embd, pano_prompt_embd, cameras,
                pers_layout_cond=None, pano_layout_cond=None):
        """Reference signature (MVGenModel.py:38-39).  Inference unless the model was built ``differentiable`` and
        autograd is recording."""
        args = (latents, pano_latent, timestep, prompt_embd, pano_prompt_embd, cameras, pers_layout_cond, pano_layout_cond)
        if self.differentiable and torch.is_grad_enabled():
            from ... import train_engine
            params = [t for t in self.trainable_tensors() if t.requires_grad]
            if params:
                out = train_engine.DenoiserFunction.apply(self, args, *params)
                return out if isinstance(out, tuple) else (None, out)       # PanoOnly: (sample = None, pano_sample) as the reference
        return self._forward(*args)

    @torch.no_grad()
    def refold_lora(self):
        """Bring the packed attention projections up to date when a LoRA matrix changed since the pack was built / last
        re-folded (optimizer steps and in-place copies bump the version counters): the forward kernels read
        W + up @ down folded into one 16-bit weight.  Only the 4 x 32 projections per UNet are re-folded, IN PLACE
        (engine.fold_attention: one kernel launch each, which also refreshes the backward's operands once a training step has
        attached them); the frozen weights stay.  Called at the top of EVERY forward --
        also the no_grad ones (validation / predict after fit, DenoiseLoop): the key is a tuple of ~600 ints."""
        if not self._packed:
            return
        key = self._lora_versions()
        for (which, *_), u in self._packed.items():
            if which.endswith("_cn") or getattr(u, "lora_key", key) == key:
                continue
            for t in engine.all_transformers(u):
                engine.fold_attention(t.attn1)          # in place: one launch per LoRA-carrying projection
                engine.fold_attention(t.attn2)
            # K / V^T of the cached prompts depend on to_k / to_v: recomputed INTO the cached buffers -- a DenoiseLoop graph
            # captured before the optimizer step reads them (like the folded weights) by address, and a replay must not mix
            # the new to_q / to_out with K / V^T of the old to_k / to_v
            # Only the entries a captured graph can still read (DenoiseLoop._graph_keepalive pins them) are refreshed; the others are
            # dropped and recomputed on their next use (up to 4 prompts x 16 cross-attentions x 2 GEMMs per optimizer step
            # for tensors nobody would read again).  The refresh runs on the CURRENT stream: a caller that replays a DenoiseLoop graph
            # on another stream while training on this one must order the two itself (wait_stream), as for any shared weight.
            cache = getattr(u, "text_kv_cache", {})
            for ck in [ck for ck, hit in cache.items() if not hit.get("pins")]:     # (an empty WeakSet: every graph that read the entry is gone)
                del cache[ck]
            for hit in cache.values():
                for t in engine.all_transformers(u):
                    k, vt = engine.text_kv(t.attn2, hit["text"])
                    hit[id(t)][0].copy_(k)
                    hit[id(t)][1].copy_(vt)
            u.lora_key = key

    @torch.no_grad()
    def _forward(self, latents, pano_latent, timestep, prompt_embd, pano_prompt_embd, cameras,
                 pers_layout_cond=None, pano_layout_cond=None, tape=None):
        """tape: a list that receives (layer, inputs) records for train_engine.backward (training forward)."""
        self.refold_lora()
        if self.pers_cn is None:
            pers_layout_cond = None                 # reference MVGenModel.py:62-65
        if self.pano_cn is None:
            pano_layout_cond = None
        dev = pano_latent.device
        dt = self.compute_dtype
        two = self.unet is not None
        branches = []
        cn_res = {}                                 # id(branch) -> (12 skip residuals, mid residual)
        shard = getattr(self, "shard", None)      # set by sharding.ShardedDenoiseLoop: latents hold only
        from ... import train_engine
        if tape is not None:
            if shard is not None:
                raise NotImplementedError("the training path covers the un-sharded denoiser")
            # Layout conditions under training (PanFusion.py:85-89).  A ControlNet with parameters that require gradients
            # (the reference's layout_cond=True: every ControlNet parameter trains at lr x 0.1, PanoGenerator.py:153-157)
            # runs on a tape of its own (train_engine.controlnet_forward); the tape of the branch it feeds notes where its
            # 12 + 1 residuals were added, and the backward hands the skip / mid gradients found there to
            # train_engine.controlnet_backward.  A frozen ControlNet runs as in inference: its residuals are constants,
            # gradients flow THROUGH the additions into the EPA blocks and the LoRA matrices.
            from ... import train_engine
            make_branch = lambda *a, **k: train_engine.TrainBranch(tape, *a, **k)
        else:
            make_branch = engine.Branch
        cn_recs = {}                                # id(branch) -> train_engine record of a trainable ControlNet's forward

        def controlnet(which, br, latent, t, cond):
            c = self.packed(which, dev)
            if tape is not None and any(p_.requires_grad for p_ in getattr(self, which).parameters()):
                cn_res[id(br)], cn_recs[id(br)] = train_engine.controlnet_forward(c, latent, t, br.text, cond)
            else:
                cn_res[id(br)] = engine.run_controlnet(c, latent, t, br.text, cond)
        if two:                                   # this rank's views, cameras all m of them
            b, m = latents.shape[:2]
            flat_cams = {k: v.reshape(-1) for k, v in cameras.items()}
            m_total, groups = camera_groups(flat_cams, b)
            if shard is None and m_total != m:
                raise ValueError("cameras describe %d views but latents hold %d" % (m_total, m))
            pano_t = timestep[:, 0]
        # panorama owner that was given no views (sharding split 0, ...): no view branch at all, the EPA blocks
        # compute their panorama-query half from the gathered view tokens
        pano_only = two and shard is not None and m == 0
        pers = None
        if two and not pano_only:
            pers = make_branch(self.packed("unet", dev), latents.flatten(0, 1), timestep.reshape(-1),
                               self._prompt16(prompt_embd), pano=False, pad=False)
            branches.append(pers)
            if shard is not None and shard.pano_g is not None and not shard.has_pano and tape is None and b == 1:
                # a view rank of the panorama-rank layout: after each of its own self-attentions it computes its share of the
                # owner's panorama self-attention at the same UNet position, where that one is split (sharding.splits_pano_attention)
                from ... import sharding as _sh
                lat_h, pano_tokens = latents.shape[-2], pano_latent.shape[-2] * pano_latent.shape[-1]
                # (the PANORAMA UNet's attention weights: packed on this rank too -- its self-attentions are met in call order)
                pano_ts = list(engine.all_transformers(self.packed("pano_unet", dev))) if _sh.splits_pano_attention(shard, pano_tokens) else []
                seen = [0]

                def help_owner(t_pack, h, lat_h=lat_h, pano_tokens=pano_tokens):
                    k = seen[0]
                    seen[0] += 1
                    sc = lat_h // h.shape[1]
                    tokens = pano_tokens // (sc * sc)
                    if _sh.splits_pano_attention(shard, tokens):
                        _sh.help_pano_attention(shard, pano_ts[k], tokens, h)
                pers.attn_help = help_owner
            if pers_layout_cond is not None:        # reference :66-74 (ControlNet on the view branch)
                controlnet("pers_cn", pers, latents.flatten(0, 1), timestep.reshape(-1), pers_layout_cond.flatten(0, 1))
        else:
            pano_t = timestep
        # view-sharded rank without the panorama branch (sharding layout "pano_rank"): the view branch only;
        # the panorama tokens arrive by broadcast inside every EPA block
        view_only = two and shard is not None and not shard.has_pano
        main = torch.cuda.current_stream(dev) if pano_latent.is_cuda else None
        side = None
        train_streams = tape is None or train_engine.TWO_STREAMS       # (a training step overlaps the two branches too)
        if two and self.two_streams and main is not None and not view_only and not pano_only and train_streams:
            if self._side is None:
                # PF_PANO_PRIORITY=1: the panorama branch's stream gets HIGH priority.  Its ~1400 small kernels are the
                # critical path at the deep levels (the view stream idles 3-4 ms per step at the joins there,
                # profiles/archive/r3f_streams.txt): with priority their workgroups take the first compute units that come free
                # instead of queueing behind a whole round of the view branch's persistent GEMM blocks.
                self._side = torch.cuda.Stream(dev, priority=-1 if os.environ.get("PF_PANO_PRIORITY", "0") == "1" else 0)
            side = self._side
        keep = []                                   # tensors produced on one stream and read on the other stay
                                                    # referenced until the next join (allocator reuse is per stream)

        def on_pano():
            return torch.cuda.stream(side) if side is not None else contextlib.nullcontext()

        def fork():
            if side is not None:
                side.wait_stream(main)

        def join():
            if side is not None:
                main.wait_stream(side)
                keep.clear()

        fork()
        keeps = tape is not None and train_engine.KEEP  # (such a forward projects the padded text inside every block)
        if side is not None and not keeps:          # the view branch's text K / V^T: 32 tiny GEMMs, off the critical path
            with on_pano():                         # (computed once per prompt tensor, then served from the cache)
                pers.precompute_text_kv()
                pers.text_ready = torch.cuda.Event()
                pers.text_ready.record(side)
        elif pers is not None and not keeps:
            pers.precompute_text_kv()
        pano = None
        if not view_only:
            with on_pano():
                pano = make_branch(self.packed("pano_unet", dev), pano_latent.flatten(0, 1), pano_t,
                                   self._prompt16(pano_prompt_embd), pano=True, pad=self.pano_pad)
                pano.on_side = side is not None     # (train_engine.backward walks its entries on the same stream)
                if shard is not None and shard.pano_g is not None and tape is None and pano_latent.shape[0] == 1:
                    pano.attn_split = shard
                    if side is not None:
                        pano.split_streams, pano.keep = (main, side), keep
                if not keeps:
                    pano.precompute_text_kv()
                if pano_layout_cond is not None:    # reference :75-83: plain convolutions on the un-padded latent
                    controlnet("pano_cn", pano, pano_latent.flatten(0, 1), pano_t, pano_layout_cond.flatten(0, 1))
            branches.append(pano)

        def each_branch(fn):
            for br in branches:
                if br is pano:
                    with on_pano():
                        fn(br)
                else:
                    fn(br)

        # A panorama-only owner contributes nothing to an EPA block's all-gather of view tokens: it POSTS it before it runs the
        # level's panorama resnets and collects the tokens when it reaches the block.  Not when a
        # self-attention of the level is query-split: those collectives would then be issued in a different order than on the view ranks.
        posted = {}
        can_post = False
        if pano_only and b == 1:
            from ... import sharding as _sh
            can_post = _sh.ASYNC and not _sh.splits_pano_attention(shard, pano_latent.shape[-2] * pano_latent.shape[-1])

        def prepost(block, after_down=False):
            if not can_post:
                return
            sc = pano_latent.shape[-2] // pano.h.shape[1] * (2 if after_down else 1)
            hw = (latents.shape[-2] // sc, latents.shape[-1] // sc)
            posted[id(block)] = block.post_view_gather(shard, hw, dev)

        def fuse(block):
            if pano_only:                           # view feature map size at this level, from the latents' ratio
                sc = pano_latent.shape[-2] // (pano.h.shape[1])
                hw = (latents.shape[-2] // sc, latents.shape[-1] // sc)
                _, pano.h = block.forward_nhwc(None, pano.h, groups, m_total, shard=shard, pers_hw=hw, posted=posted.pop(id(block), None))
                return
            if view_only:                           # panorama feature map size at this level, from the latents' ratio
                sc = latents.shape[-2] // pers.h.shape[1]
                hw = (pano_latent.shape[-2] // sc, pano_latent.shape[-1] // sc)
                pers.h, _ = block.forward_nhwc(pers.h, None, groups, m_total, shard=shard, equi_hw=hw)
                return
            join()
            keep.append(pano.h)
            if tape is not None and train_engine.KEEP:      # training forward that keeps the block's activations
                xp_in, xe_in = pers.h, pano.h
                pers.h, pano.h, rec = block.forward_nhwc_keep(xp_in, xe_in, groups, m_total)
                tape.append(("fuse", pers, pano, block, xp_in, xe_in, groups, m_total, rec))
                fork()
                return
            if tape is not None:
                tape.append(("fuse", pers, pano, block, pers.h, pano.h, groups, m_total, None))
            epa_side = side if os.environ.get("PF_EPA_STREAMS", "2") != "1" else None
            pers.h, pano.h = block.forward_nhwc(pers.h, pano.h, groups, m_total, shard=shard, side=epa_side)
            keep.append(pano.h)
            fork()

        pu = branches[-1].u
        # encoder (reference :98-152): EPA after each downsample
        for i in range(len(pu.down)):
            def level(br, i=i):
                blk = br.u.down[i]
                for j, r in enumerate(blk.resnets):
                    br.resnet(r)
                    if blk.attns is not None:
                        br.attention(blk.attns[j])
                    br.push()
                if blk.down is not None:
                    br.downsample(blk.down)
                    br.push()
            if pu.down[i].down is not None and two:
                prepost(self.cp_blocks_encoder[i], after_down=True)
            each_branch(level)
            if pu.down[i].down is not None and two:
                fuse(self.cp_blocks_encoder[i])

        # ControlNet residuals onto the skip stack (reference :154-170) and after the mid block (:200-203)
        def add_skips(br):
            if id(br) in cn_res:
                br.skips = [engine.ops.add(sk, r) for sk, r in zip(br.skips, cn_res[id(br)][0])]
                if id(br) in cn_recs:
                    tape.append(("cn_skips", br, cn_recs[id(br)]))
        each_branch(add_skips)

        # mid (reference :172-207)
        def middle(br):
            mid = br.u.mid
            br.resnet(mid.resnets[0])
            for a, r in zip(mid.attns, mid.resnets[1:]):
                br.attention(a)
                br.resnet(r)
            if id(br) in cn_res:
                br.h = engine.ops.add(br.h, cn_res[id(br)][1])
                if id(br) in cn_recs:
                    tape.append(("cn_mid", br, cn_recs[id(br)]))
        if two:
            prepost(self.cp_blocks_mid)
        each_branch(middle)
        if two:
            fuse(self.cp_blocks_mid)
        # decoder (reference :210-277): EPA before each upsample
        for i in range(len(pu.up)):
            def level_up(br, i=i):
                blk = br.u.up[i]
                for j, r in enumerate(blk.resnets):
                    br.resnet(r, skip=True)
                    if blk.attns is not None:
                        br.attention(blk.attns[j])
            if pu.up[i].up is not None and two:
                prepost(self.cp_blocks_decoder[i])
            each_branch(level_up)
            if pu.up[i].up is not None:
                if two:
                    fuse(self.cp_blocks_decoder[i])
                each_branch(lambda br, i=i: br.upsample(br.u.up[i].up))

        pano_head = None
        if pano is not None:
            with on_pano():
                pano_head = pano.head()
        join()
        out_dtype = pano_latent.dtype
        pano_sample = pano_head.to(out_dtype).unflatten(0, (-1, 1)) if pano_head is not None else None
        if pano_only:
            sample = latents.new_zeros(b, 0, *latents.shape[2:]).to(out_dtype)
        else:
            sample = pers.head().to(out_dtype).unflatten(0, (b, m)) if two else None
        if tape is not None:
            return sample, pano_sample, pers, pano, side
        return sample, pano_sample
