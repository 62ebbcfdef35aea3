"""EPA fusion block with the reference's name and call signature
(reference ``models/pano/modules.py:8-59``): ``WarpAttn(dim).forward(pers_x,
equi_x, cameras) -> (pers_x_out, equi_x_out)`` on NCHW tensors.
"""
import torch
import torch.nn as nn

from ... import engine, ops
from ..modules.transformer import BasicTransformerBlock, SphericalPE


_HOST_CAMS = {}        # storage identity + version of a device tensor -> (the tensor, its host values)


def _host_values(v):
    """Angles are read as host scalars (the reference does `.item()` per camera).  A device tensor costs a
    synchronising copy: done once per (storage, version), not once per forward (the reference's cameras are
    GPU tensors that do not change inside a sampling loop)."""
    if not isinstance(v, torch.Tensor):
        return list(v)
    if not v.is_cuda:
        return v.detach().tolist()
    key = (v.data_ptr(), v._version, tuple(v.shape), tuple(v.stride()), str(v.device), v.dtype)
    hit = _HOST_CAMS.get(key)
    if hit is None:
        if len(_HOST_CAMS) > 256:
            _HOST_CAMS.clear()
        # the entry holds the tensor: its storage cannot be freed and handed to another tensor (same address, version 0,
        # other angles) while the key is alive
        hit = _HOST_CAMS[key] = (v, v.detach().cpu().tolist())
    return hit[1]


def camera_groups(cameras, b):
    """cameras: dict of flattened (b*m,) values -> per-batch-element host tuples."""
    fov, theta, phi = _host_values(cameras["FoV"]), _host_values(cameras["theta"]), _host_values(cameras["phi"])
    m = len(fov) // b
    return m, [(tuple(fov[i * m:(i + 1) * m]), tuple(theta[i * m:(i + 1) * m]), tuple(phi[i * m:(i + 1) * m]))
               for i in range(b)]


def _repack_hook(module, incompatible):
    """load_state_dict post hook; module-level so that the block stays picklable."""
    module.repack()


class WarpAttn(nn.Module):
    def __init__(self, dim, compute_dtype=torch.float16, precision=None):
        super().__init__()
        self.precision = precision or engine.default_precision(compute_dtype)
        self.transformer = BasicTransformerBlock(dim, dim // 32, 32, context_dim=dim)
        self.pe = SphericalPE(dim // 4)
        self.compute_dtype = compute_dtype
        self._packed = None
        self._tables = engine.EPATables()
        # checkpoints loaded through a PARENT module never reach a load_state_dict override of this class
        # (nn.Module recurses via _load_from_state_dict): the post hook fires either way
        self.register_load_state_dict_post_hook(_repack_hook)

    def packed(self, device):
        # (the parameters' version counters: an optimizer step between two training forwards re-packs)
        key = (device, self.compute_dtype, self.precision, tuple(t._version for t in self.transformer.parameters()))
        if self._packed is None or self._packed.key != key:
            self._packed = engine.pack_epa(self, device, self.compute_dtype, self.precision == "mixed")
            self._packed.key = key
        return self._packed

    def repack(self):
        self._packed = None

    def tables_for(self, groups, ph, pw, eh, ew, device):
        e = self.packed(device)
        uniq = [groups[0]] if all(g == groups[0] for g in groups) else groups
        return [self._tables.get(f, t, p, ph, pw, eh, ew, e.freq, device) for f, t, p in uniq]

    @torch.no_grad()
    def post_view_gather(self, shard, pers_hw, device):
        """A panorama owner WITHOUT views posts this block's all-gather of the view tokens ahead of time (its own block of it is
        empty) and hands the handle to forward_nhwc(posted=...): the tokens of the view ranks travel while it runs the panorama
        resnets of the level (sharding.gather_view_tokens_async)."""
        from ... import sharding
        e = self.packed(device)
        return sharding.gather_view_tokens_async(None, shard, P=pers_hw[0] * pers_hw[1], C=e.wqk.shape[-1], dtype=e.cdtype, device=device)

    @torch.no_grad()
    def forward_nhwc(self, xp, xe, groups, m, shard=None, equi_hw=None, side=None, pers_hw=None, posted=None):
        """xp [b*m, ph, pw, C], xe [b, eh, ew, C] NHWC in the stream dtype (the denoiser's internal layout:
        16-bit, or fp32 in the mixed scheme).
        With ``shard`` (sharding.ShardInfo) xp holds only this rank's views of the m; on a rank without the
        panorama branch xe is None and equi_hw = (eh, ew); on a panorama owner without views xp is None and
        pers_hw = (ph, pw)."""
        dev = xp.device if xp is not None else xe.device
        e = self.packed(dev)
        eh, ew = (xe.shape[1], xe.shape[2]) if xe is not None else equi_hw
        ph, pw = (xp.shape[1], xp.shape[2]) if xp is not None else pers_hw
        tabs = self.tables_for(groups, ph, pw, eh, ew, dev)
        return engine.run_epa(e, tabs, xp, xe, m, shard=shard, equi_hw=(eh, ew), side=side, pers_hw=(ph, pw), n_samples=len(groups), posted=posted)

    @torch.no_grad()
    def forward_inference(self, pers_x, equi_x, cameras):
        b = equi_x.shape[0]
        m, groups = camera_groups(cameras, b)
        dt = engine.stream_dtype(self.compute_dtype, self.precision)
        xp = ops.nchw_to_nhwc(pers_x.float(), dt)
        xe = ops.nchw_to_nhwc(equi_x.float(), dt)
        op, oe = self.forward_nhwc(xp, xe, groups, m)
        return (ops.nhwc_to_nchw(op, torch.float32).to(pers_x.dtype),
                ops.nhwc_to_nchw(oe, torch.float32).to(equi_x.dtype))

    def forward(self, pers_x, equi_x, cameras):
        """Same call as the reference (modules.py:15).  Under autograd (training, PanFusion.py:64-98) the block is
        differentiable in its two inputs and its own parameters: forward on the inference kernels, backward on the
        HIP backward kernels with the block recomputed (the reference's CheckpointFunction, transformer.py:77-127)."""
        from ... import training
        params = training.train_params(self)
        if torch.is_grad_enabled() and any(t.requires_grad for t in (pers_x, equi_x, *params)):
            return training.WarpAttnFunction.apply(self, cameras, pers_x, equi_x, *params)
        return self.forward_inference(pers_x, equi_x, cameras)

    @torch.no_grad()
    def backward_block(self, pers_x, equi_x, cameras, d_pers, d_equi):
        """Gradients of forward() for output gradients d_pers / d_equi (NCHW, None = zero):
        (d pers_x, d equi_x, [parameter gradients in training.train_params order])."""
        from ... import training
        b = equi_x.shape[0]
        m, groups = camera_groups(cameras, b)
        dev = pers_x.device
        bm, Cc, ph, pw = pers_x.shape
        _, _, eh, ew = equi_x.shape
        tabs = self.tables_for(groups, ph, pw, eh, ew, dev)
        sdt = engine.stream_dtype(self.compute_dtype, self.precision)
        tok = lambda x, d: ops.nchw_to_nhwc(x.float(), d).view(-1, Cc)
        zeros = lambda like: torch.zeros(like.shape, device=dev, dtype=torch.float32)
        d_pers = zeros(pers_x) if d_pers is None else d_pers
        d_equi = zeros(equi_x) if d_equi is None else d_equi
        e = self.packed_train(dev)
        rec = training.epa_recompute(e, tabs, tok(equi_x, sdt), tok(pers_x, sdt), b, m)
        dx_e, dx_p, grads = training.epa_backward(e, tabs, rec, tok(d_equi, torch.float32), tok(d_pers, torch.float32), b, m)
        dx_p = ops.nhwc_to_nchw(dx_p.view(bm, ph, pw, Cc), torch.float32).to(pers_x.dtype)
        dx_e = ops.nhwc_to_nchw(dx_e.view(b, eh, ew, Cc), torch.float32).to(equi_x.dtype)
        params = training.train_params(self)
        return dx_p, dx_e, [g.view(p_.shape).to(p_.dtype) for g, p_ in zip(grads, params)]

    @torch.no_grad()
    def forward_nhwc_keep(self, xp, xe, groups, m):
        """forward_nhwc for a training step that keeps its activations (train_engine.KEEP): the block through
        training.epa_recompute (the record the backward reads) and the FF2 + residual tail.  -> (out_p, out_e, record)."""
        from ... import training
        dev = xp.device
        bm, ph, pw, Cc = xp.shape
        b, eh, ew, _ = xe.shape
        tabs = self.tables_for(groups, ph, pw, eh, ew, dev)
        e = self.packed_train(dev)
        rec = training.epa_recompute(e, tabs, xe.reshape(-1, Cc), xp.reshape(-1, Cc), b, m)
        out = ops.linear(rec.g, e.w2, bias=e.b2, residual=rec.y)
        return out[rec.Te:].view(bm, ph, pw, Cc), out[:rec.Te].view(b, eh, ew, Cc), rec

    @torch.no_grad()
    def backward_nhwc(self, xp, xe, groups, m, d_p, d_e, rec=None):
        """The same on the denoiser's internal layout: xp [b*m, ph, pw, C], xe [b, eh, ew, C] (the inputs of forward_nhwc),
        d_p / d_e fp32 NHWC gradients of its two outputs -> (dxp, dxe fp32 NHWC, parameter gradients).
        rec: the record forward_nhwc_keep left; None: the block is recomputed here."""
        from ... import training
        dev = xp.device
        bm, ph, pw, Cc = xp.shape
        b, eh, ew, _ = xe.shape
        tabs = self.tables_for(groups, ph, pw, eh, ew, dev)
        e = self.packed_train(dev)
        if rec is None:
            rec = training.epa_recompute(e, tabs, xe.reshape(-1, Cc), xp.reshape(-1, Cc), b, m)
        dx_e, dx_p, grads = training.epa_backward(e, tabs, rec, d_e.reshape(-1, Cc).float(), d_p.reshape(-1, Cc).float(), b, m)
        params = training.train_params(self)
        return dx_p.view(bm, ph, pw, Cc), dx_e.view(b, eh, ew, Cc), [g.view(p_.shape) for g, p_ in zip(grads, params)]

    def packed_train(self, device):
        """16-bit weight copies for the backward GEMMs, rebuilt when a parameter changed (optimizer steps bump
        the tensors' version counters)."""
        from ... import training
        key = (device, self.compute_dtype, tuple((t.data_ptr(), t._version) for t in training.train_params(self)))
        hit = getattr(self, "_packed_train", None)
        if hit is None or hit.key != key:
            hit = training.pack_epa_train(self, device, self.compute_dtype)
            hit.key = key
            self._packed_train = hit
        return hit
