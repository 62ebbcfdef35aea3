"""The DDIM sampling loop around the denoiser (reference
``models/pano/PanFusion.py:30-43,100-164``, ``PanoGenerator.py:240-269``).

``DenoiseLoop`` owns the static device buffers of one sampling run and runs
one loop iteration -- 90-degree latent roll, CFG-paired dual-branch denoiser,
CFG merge, two DDIM updates -- per ``step()``.  The denoiser call can be
captured into one hipGraph per rotation offset (the geometry is 4-periodic,
SURVEY.md §4) and replayed; the CFG+DDIM update is a separate fused kernel
whose scalar coefficients change every step.
"""
import os

import torch

from . import ops
from .external.Perspective_and_Equirectangular import e2p


class DDIMSchedule:
    """diffusers DDIMScheduler with the SD-2-base config (scaled-linear betas 0.00085..0.012,
    1000 train steps, steps_offset 1, leading spacing, eta 0, epsilon prediction)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.num_train_timesteps, self.steps_offset = num_train_timesteps, steps_offset

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        self.timesteps = [int(i * ratio) + self.steps_offset for i in range(n)][::-1]
        return self.timesteps

    def coefficients(self, t):
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return (float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_prev ** 0.5), float((1 - a_prev) ** 0.5))


def init_noise(pano_noise, cameras, pers_h, pers_w):
    """View noise = nearest-neighbour e2p of the SAME panorama noise (PanFusion.py:30-43).
    pano_noise (bs, 1, 4, H, W) on the GPU; cameras: dict of (bs, m)."""
    bs = pano_noise.shape[0]
    m = cameras["FoV"].shape[1]
    flat = {k: v.reshape(-1) for k, v in cameras.items()}
    rep = pano_noise.expand(-1, m, -1, -1, -1).flatten(0, 1)
    noise = e2p(rep, flat["FoV"], flat["theta"], flat["phi"], (pers_h, pers_w), mode="nearest")
    return pano_noise, noise.unflatten(0, (bs, m))


def rotate_cameras(cameras, degree):
    cams = dict(cameras)
    cams["theta"] = (cams["theta"] + degree) % 360
    return cams


class DenoiseLoop:
    """One text-to-panorama sampling run (batch 1 prompt, CFG pair inside)."""

    def __init__(self, model, latents, pano_latent, prompt_embd, pano_prompt_embd, cameras,
                 steps=50, rot_diff=90.0, guidance_scale=9.0, use_graphs=False, pano_layout_cond=None):
        """latents (1, m, 4, h, w), pano_latent (1, 1, 4, H, W) fp32 on the GPU;
        prompt_embd (2, m, L, D) / pano_prompt_embd (2, 1, L, D) = [null ; prompt];
        cameras: dict of (1, m) CPU tensors (FoV, theta, phi in degrees)."""
        self.model, self.guidance, self.rot_diff = model, guidance_scale, rot_diff
        # The loop's state lives as the CFG PAIR the denoiser reads ([x ; x], gen_cls_free_guide_pair, PanoGenerator.py:240-251):
        # the DDIM kernel writes both halves, so no torch.cat runs between two calls.  self.lat / self.pano are the first halves.
        self.lat2 = torch.stack([latents.float()[0]] * 2).contiguous()
        self.pano2 = torch.stack([pano_latent.float()[0]] * 2).contiguous()
        self.lat, self.pano = self.lat2[:1], self.pano2[:1]
        self.prompt, self.pano_prompt = prompt_embd, pano_prompt_embd
        self.cameras = {k: v.detach().cpu() for k, v in cameras.items()}
        self.m = latents.shape[1]
        self.W = pano_latent.shape[-1]
        self.shift = int(rot_diff / 360 * self.W)          # PanoGenerator.py:269
        self.sched = DDIMSchedule()
        self.timesteps = self.sched.set_timesteps(steps)
        self.tstep = torch.empty(2, self.m, dtype=torch.long, device=latents.device)
        self._tstep_value = None                          # what self.tstep holds (the DDIM kernel writes the next step's value)
        self.i = 0
        self.total_rot = 0.0
        self.use_graphs = use_graphs
        self.graphs = {}
        # layout condition image (1, 1, 3, Hi, Wi) for the panorama ControlNet: rolled with the panorama
        # every step (PanFusion.py:150-153); one static copy per rotation offset (the loop is 4-periodic)
        self.layout = None if pano_layout_cond is None else pano_layout_cond.float().contiguous()
        self._layout_rot = {}
        self.eps = self.pano_eps = None
        # the loop rolls the panorama BEFORE each denoiser call (PanFusion.py:149); afterwards the
        # DDIM kernel writes the next latent already rolled for the following step.
        self._rot_of = {}                                  # camera-theta key -> accumulated rotation (degrees)
        if rot_diff % 360:
            self.pano2.copy_(ops.roll_width(self.pano2, self.shift))       # (one-off set-up, not part of a step)
        self.cameras = rotate_cameras(self.cameras, rot_diff)
        self.total_rot += rot_diff
        self._rot_of[tuple(float(v) for v in self.cameras["theta"].reshape(-1))] = self.total_rot

    def _layout_for(self, cams):
        """The condition image rolled by the rotation these cameras carry (PanoGenerator.py:264-269)."""
        if self.layout is None:
            return None
        rot = float(self._rot_of.get(tuple(float(v) for v in cams["theta"].reshape(-1)), self.total_rot)) % 360
        if rot not in self._layout_rot:
            shift = int(rot / 360 * self.layout.shape[-1])
            img = ops.roll_width(self.layout, shift) if shift else self.layout
            self._layout_rot[rot] = torch.cat([img, img])          # CFG pair (gen_cls_free_guide_pair)
        return self._layout_rot[rot]

    def _denoise(self, cams):
        cams2 = {k: torch.cat([v, v]) for k, v in cams.items()}                 # (host tensors)
        return self.model(self.lat2, self.pano2, self.tstep, self.prompt, self.pano_prompt, cams2,
                          None, self._layout_for(cams))

    def _set_tstep(self, t):
        if self._tstep_value != t:
            self.tstep.fill_(t)
            self._tstep_value = t

    MAX_GRAPHS = 8        # distinct rotation offsets kept as graphs (4 at rot_diff = 90); beyond that: eager launches

    def _denoise_graphed(self, cams):
        key = tuple(float(v) for v in cams["theta"].reshape(-1))
        g = self.graphs.get(key)
        if g is None and len(self.graphs) >= self.MAX_GRAPHS:
            # a rotation step that does not divide 360 never revisits an offset: capturing (and keeping) a graph
            # plus its table set per step would re-capture every step and grow without bound
            return self._denoise(cams)
        if g is None:
            from .engine import EPATables
            with EPATables.pinned():                 # the graph reads its geometry tables by address
                self._denoise(cams)                  # warm-up: builds tables, sets kernel attributes
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                # PF_VIEW_PRIORITY: capture on the caller's (high-priority) stream, so that the view branch = the critical
                # path keeps that priority and the panorama side stream (default priority) only fills what it leaves idle
                cap = torch.cuda.current_stream() if os.environ.get("PF_VIEW_PRIORITY") else None
                with torch.cuda.graph(graph, stream=cap):
                    out = self._denoise(cams)
            g = (graph, out, self._graph_keepalive())
            self.graphs[key] = g
        g[0].replay()
        return g[1]

    def _graph_keepalive(self):
        """Tensors a captured graph reads by address but that live in bounded caches of the model (16-bit prompt
        casts, text K / V^T per UNet): referenced from the graph entry so that an eviction cannot free them under it."""
        import weakref

        class _Pin:                               # lives exactly as long as the graph entry that holds the returned list
            pass
        owner = _Pin()
        keep = [owner, list(getattr(self.model, "_prompt_cache", {}).values())]
        for packed in getattr(self.model, "_packed", {}).values():
            hits = list(getattr(packed, "text_kv_cache", {}).values())
            for hit in hits:
                # a captured graph reads these buffers by address: refold_lora refreshes them IN PLACE for as long as some graph entry
                # is alive (a WeakSet of the entries' tokens: dropping the loop / its graphs un-pins the entry -- ADVICE r5), and
                # Branch.precompute_text_kv does not evict a pinned entry
                hit.setdefault("pins", weakref.WeakSet()).add(owner)
            keep.append(hits)
        return keep

    def prepare(self):
        """Untimed set-up: build the geometry tables (and capture one hipGraph) for every rotation
        offset the loop will visit (4 at rot_diff = 90)."""
        cams, seen = self.cameras, set()
        self._set_tstep(self.timesteps[0])
        rot = self.total_rot
        for _ in range(64):
            key = tuple(float(v) for v in cams["theta"].reshape(-1))
            if key in seen:
                break
            seen.add(key)
            self._rot_of.setdefault(key, rot)
            rot += self.rot_diff
            (self._denoise_graphed if self.use_graphs else self._denoise)(cams)
            cams = rotate_cameras(cams, self.rot_diff)
        torch.cuda.synchronize()

    def step_eager(self):
        keep, self.use_graphs = self.use_graphs, False
        try:
            self.step()
        finally:
            self.use_graphs = keep

    def step(self):
        """One iteration of the loop body (PanFusion.py:146-162)."""
        t = self.timesteps[self.i]
        self._set_tstep(t)                                # (a no-op after the first step: the DDIM kernel left it there)
        run = self._denoise_graphed if self.use_graphs else self._denoise
        eps, pano_eps = run(self.cameras)
        coef = self.sched.coefficients(t)
        last = self.i == len(self.timesteps) - 1
        # Two launches update the whole loop state IN PLACE (captured graphs read it by address): views -- plain update, both
        # halves of the CFG pair; panorama -- update + roll for the next iteration (a block owns whole rows, so in place for any
        # roll), both halves, and the next call's timestep words.
        ops.cfg_ddim_step_pair(self.lat, eps[0], eps[1], self.guidance, coef, 0, out=self.lat, out2=self.lat2[1:])
        t_next = t if last else self.timesteps[self.i + 1]
        ops.cfg_ddim_step_pair(self.pano, pano_eps[0], pano_eps[1], self.guidance, coef, 0 if last else self.shift,
                               out=self.pano, out2=self.pano2[1:], tstep=self.tstep, t_next=t_next)
        self._tstep_value = t_next
        self.i += 1
        if not last:
            self.cameras = rotate_cameras(self.cameras, self.rot_diff)
            self.total_rot += self.rot_diff
            self._rot_of.setdefault(tuple(float(v) for v in self.cameras["theta"].reshape(-1)), self.total_rot)

    def run(self):
        while self.i < len(self.timesteps):
            self.step()
        return self.result()

    def result(self):
        """Latents with the accumulated rotation undone (PanFusion.py:164)."""
        back = int(-self.total_rot / 360 * self.W)
        return self.lat, ops.roll_width(self.pano, back)


def add_noise(sched, x, noise, t):
    """``scheduler.add_noise`` (diffusers DDIMScheduler / DDPMScheduler): sqrt(abar_t) x + sqrt(1 - abar_t) noise, t (b,) per sample."""
    out = torch.empty_like(x, dtype=torch.float32)
    for i, ti in enumerate(t.tolist()):
        a = float(sched.alphas_cumprod[ti])
        ops.axpby(x[i], noise[i], a ** 0.5, (1 - a) ** 0.5, out=out[i])
    return out


def training_step(model, vae_encoder, images, pano, cameras, prompt_embd, pano_prompt_embd, latent_pad=8, sched=None,
                  draws=None, generator=None, pers_layout_cond=None, pano_layout_cond=None):
    """The body of ``PanFusion.training_step`` (PanFusion.py:64-98) on the HIP path: VAE-encode the views and the circularly
    padded panorama, draw the timestep and the panorama noise, project that noise into the views (``init_noise``), add
    noise, ONE denoiser call without CFG, MSE on both predictions.  Returns (loss, loss_pers, loss_pano); ``loss.backward()``
    then fills the gradients of ``model.trainable_tensors()`` (the model must have been built ``differentiable=True``).

    images (b, m, 3, H, W), pano (b, 1, 3, Hp, Wp) in [-1, 1] on the GPU; cameras: dict of (b, m); the prompt embeddings as
    ``embed_prompt`` leaves them (b, m, L, D) / (b, 1, L, D) (text_encoder.TextEncoder).  draws: optional dict with the
    random draws (``eps_views``, ``eps_pano`` for the VAE posterior, ``t`` (b,), ``pano_noise`` (b, 1, 4, h, w)) -- what the
    tests fix to compare against the oracle; anything missing is drawn here.
    pers_layout_cond / pano_layout_cond: ``batch.get('images_layout_cond')`` / ``batch.get('pano_layout_cond')`` as the
    reference passes them (PanFusion.py:85-89).  Every ControlNet parameter that requires a gradient gets one (the
    reference's trainable set under layout_cond=True, PanoGenerator.py:153-157: all of them; train_engine.controlnet_backward);
    a ControlNet frozen with requires_grad_(False) contributes its residuals as constants."""
    from .utils.pano import pad_pano, unpad_pano
    from .vae import encode_image
    draws = draws or {}
    sched = sched or DDIMSchedule()
    dev = images.device
    latents = encode_image(images, vae_encoder, eps=draws.get("eps_views"), generator=generator)
    b, m, _, h, w = latents.shape
    pano_latent_pad = encode_image(pad_pano(pano, 8 * latent_pad), vae_encoder, eps=draws.get("eps_pano"), generator=generator)
    pano_latent = unpad_pano(pano_latent_pad, latent_pad).contiguous()
    t = draws.get("t")
    if t is None:
        t = torch.randint(0, sched.num_train_timesteps, (b,), device=dev, generator=generator)
    pano_noise = draws.get("pano_noise")
    if pano_noise is None:
        pano_noise = torch.randn(b, 1, 4, *pano_latent.shape[-2:], device=dev, generator=generator)
    pano_noise, noise = init_noise(pano_noise.to(dev).float(), cameras, h, w)
    noise_z = add_noise(sched, latents, noise, t)
    pano_noise_z = add_noise(sched, pano_latent, pano_noise, t)
    tt = t.to(dev).long()[:, None].repeat(1, m)
    denoise, pano_denoise = model(noise_z, pano_noise_z, tt, prompt_embd, pano_prompt_embd, cameras,
                                  pers_layout_cond, pano_layout_cond)
    loss_pers = torch.nn.functional.mse_loss(denoise, noise)
    loss_pano = torch.nn.functional.mse_loss(pano_denoise, pano_noise)
    return loss_pers + loss_pano, loss_pers, loss_pano
