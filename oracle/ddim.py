"""CPU restatement of the per-step glue around the denoiser: noise init,
latent rotation, classifier-free-guidance pairing/merge and the DDIM update.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

Follows (paths relative to /root/reference):
  * models/pano/PanFusion.py:30-43    init_noise
  * models/pano/PanFusion.py:114-123, models/pano/PanoGenerator.py:264-269  rotate_latent
  * models/pano/PanoGenerator.py:240-262  CFG pair / merge
  * models/pano/PanFusion.py:126-164  the sampling loop
  * diffusers==0.24.0 DDIMScheduler (third-party, not vendored; PARITY UNPINNED):
    SD-2-base scheduler config -- scaled-linear betas 0.00085..0.012 over 1000
    train steps, epsilon prediction, clip_sample False, set_alpha_to_one False,
    steps_offset 1, leading spacing, eta 0.
"""
import torch

from . import geometry as G


class DDIM:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                               dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.timesteps = None
        self.num_inference_steps = None

    def set_timesteps(self, n):
        ratio = self.num_train_timesteps // n
        self.num_inference_steps = n
        self.timesteps = (torch.arange(0, n) * ratio).round().flip(0).long() + self.steps_offset
        return self.timesteps

    def coefficients(self, t):
        """(sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)) as fp32 scalars."""
        t = int(t)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return a_t ** 0.5, (1 - a_t) ** 0.5, a_prev ** 0.5, (1 - a_prev) ** 0.5

    def step(self, eps, t, sample):
        sa, sb, sap, sbp = self.coefficients(t)
        x0 = (sample - sb * eps) / sa
        return sap * x0 + sbp * eps


def init_noise(pano_noise, cameras, pers_h, pers_w):
    """View noise = nearest-neighbour e2p resampling of the SAME pano noise
    (PanFusion.py:30-43).  pano_noise (bs,1,4,H,W); cameras (bs,m) dict."""
    bs = pano_noise.shape[0]
    m = cameras["FoV"].shape[1]
    flat = {k: v.flatten(0, 1) for k, v in cameras.items()}
    rep = pano_noise.expand(-1, m, -1, -1, -1).flatten(0, 1)
    noise = G.e2p(rep, flat["FoV"], flat["theta"], flat["phi"], (pers_h, pers_w), mode="nearest")
    return pano_noise, noise.unflatten(0, (bs, m))


def rotate_latent(pano_latent, cameras, degree):
    if degree % 360 == 0:
        return pano_latent, cameras
    shift = int(degree / 360 * pano_latent.shape[-1])
    cameras = dict(cameras)
    cameras["theta"] = (cameras["theta"] + degree) % 360
    return torch.roll(pano_latent, shift, dims=-1), cameras


def cfg_pair(x):
    if x is None:
        return None
    if isinstance(x, dict):
        return {k: torch.cat([v] * 2) for k, v in x.items()}
    return torch.cat([x] * 2)


def cfg_merge(pred, guidance_scale):
    uncond, cond = pred.chunk(2)
    return uncond + guidance_scale * (cond - uncond)


@torch.no_grad()
def denoise_step(model, sched, t, latents, pano_latent, prompt_embd, pano_prompt_embd, cameras,
                 rot_diff=90.0, guidance_scale=9.0):
    """One iteration of PanFusion.inference's loop body (PanFusion.py:146-162).
    prompt_embd / pano_prompt_embd already hold [null ; prompt] along batch."""
    m = latents.shape[1]
    timestep = torch.full((latents.shape[0], m), int(t), dtype=torch.long)
    pano_latent, cameras = rotate_latent(pano_latent, cameras, rot_diff)
    eps, pano_eps = model(cfg_pair(latents), cfg_pair(pano_latent), cfg_pair(timestep),
                          prompt_embd, pano_prompt_embd, cfg_pair(cameras))
    eps, pano_eps = cfg_merge(eps, guidance_scale), cfg_merge(pano_eps, guidance_scale)
    return sched.step(eps, t, latents), sched.step(pano_eps, t, pano_latent), cameras


@torch.no_grad()
def denoise_loop(model, latents, pano_latent, prompt_embd, pano_prompt_embd, cameras,
                 steps=50, rot_diff=90.0, guidance_scale=9.0):
    sched = DDIM()
    total = 0.0
    for t in sched.set_timesteps(steps):
        latents, pano_latent, cameras = denoise_step(
            model, sched, t, latents, pano_latent, prompt_embd, pano_prompt_embd, cameras,
            rot_diff, guidance_scale)
        total += rot_diff
    pano_latent, cameras = rotate_latent(pano_latent, cameras, -total)
    return latents, pano_latent


def add_noise(sched, x, noise, t):
    """diffusers ``scheduler.add_noise``: sqrt(abar_t) x + sqrt(1 - abar_t) noise with t (b,) broadcast over the sample."""
    a = sched.alphas_cumprod[t].to(x.dtype)
    shape = (-1,) + (1,) * (x.dim() - 1)
    return a.sqrt().reshape(shape) * x + (1 - a).sqrt().reshape(shape) * noise


def training_step(model, vae, images, pano, cameras, prompt_embd, pano_prompt_embd, draws, latent_pad=8, sched=None):
    """Restatement of ``PanFusion.training_step`` (PanFusion.py:64-98) with the random draws given: ``eps_views`` /
    ``eps_pano`` (VAE posterior), ``t`` (b,), ``pano_noise`` (b, 1, 4, h, w).  Returns (loss, loss_pers, loss_pano)."""
    from . import vae as OV
    sched = sched or DDIM()
    latents = OV.encode_image(images, vae, eps=draws["eps_views"].flatten(0, 1))
    b, m, _, h, w = latents.shape
    pano_pad = G.pad_pano(pano, 8 * latent_pad)
    pano_latent = G.unpad_pano(OV.encode_image(pano_pad, vae, eps=draws["eps_pano"].flatten(0, 1)), latent_pad)
    t = draws["t"].long()
    pano_noise, noise = init_noise(draws["pano_noise"], cameras, h, w)
    noise_z, pano_noise_z = add_noise(sched, latents, noise, t), add_noise(sched, pano_latent, pano_noise, t)
    denoise, pano_denoise = model(noise_z, pano_noise_z, t[:, None].repeat(1, m), prompt_embd, pano_prompt_embd, cameras)
    loss_pers = torch.nn.functional.mse_loss(denoise, noise)
    loss_pano = torch.nn.functional.mse_loss(pano_denoise, pano_noise)
    return loss_pers + loss_pano, loss_pers, loss_pano
