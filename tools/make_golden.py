"""Generate tests/golden/*.npz from the REFERENCE'S OWN code.

Runs only in the build container: imports /root/reference through
oracle/ref_import.py (third-party shims: oracle/third_party.py), drives the
reference's e2p / p2e / get_masks / get_coords / SphericalPE / WarpAttn /
MultiViewBaseModel on seeded inputs and stores inputs + outputs as small
fixtures.  UNet weights are NOT stored: they are regenerated from the seed by
oracle.sd2_unet.init_synthetic (CPU generator, machine independent).

    python tools/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ddim as oddim  # noqa: E402
from oracle import geometry as G  # noqa: E402
from oracle import mvgen as MV  # noqa: E402
from oracle import ref_import  # noqa: E402
from oracle import sd2_unet as U  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TINY = dict(width=64, cross_attention_dim=128, heads=(1, 2, 4, 4), groups=32)


def ico_deg():
    th, ph = G.icosahedron_cameras()
    return np.degrees(th), np.degrees(ph)


def tiny_inputs(seed, b, m, lat, pano_hw, L, D):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(latents=r(b, m, 4, lat, lat), pano_latent=r(b, 1, 4, *pano_hw),
                prompt_embd=r(b, m, L, D), pano_prompt_embd=r(b, 1, L, D))


def build_tiny_models(ref_cls, seed=11):
    cfg = U.tiny_config(**TINY)
    unet = U.UNet2DConditionModel(**cfg)
    pano_unet = U.UNet2DConditionModel(**cfg)
    unet.add_lora(4)
    pano_unet.add_lora(4)
    U.init_synthetic(unet, seed)
    U.init_synthetic(pano_unet, seed + 1)
    model = ref_cls(unet, pano_unet, None, None, True)
    U.init_synthetic(model.cp_blocks_encoder, seed + 2)
    U.init_synthetic(model.cp_blocks_mid, seed + 3)
    U.init_synthetic(model.cp_blocks_decoder, seed + 4)
    MV.randomize_epa(model, seed + 5)
    return model


def main():
    ref = ref_import.load()
    os.makedirs(OUT, exist_ok=True)
    thd, phd = ico_deg()

    # ---- 1. grids + nearest indices (the bit-exact contract), benchmark sizes -----------------
    idx, mapx, mapy = {}, {}, {}
    for rot in (0, 90, 180, 270):
        for name, (eh, ew, h, w) in {"64": (64, 128, 64, 64), "8": (8, 16, 8, 8)}.items():
            ii = []
            for i in range(20):
                lon, lat = ref.map_pers_pix_to_equi(eh, ew, 90, (thd[i] + rot) % 360, phd[i], h, w)
                ii.append(G.nearest_indices(lon, lat, eh, ew))
            idx["idx_%s_rot%d" % (name, rot)] = np.stack(ii).astype(np.int16)
    # float maps for one rotation at a small size (e2p) and p2e maps + masks
    cams = [((thd[i] + 90) % 360, phd[i]) for i in range(20)]
    e2p_maps = np.stack([np.stack(ref.map_pers_pix_to_equi(16, 32, 90, t, p, 16, 16)) for t, p in cams])
    p2e = [ref.map_equi_pix_to_pers(16, 16, 90, t, p, 16, 32) for t, p in cams]
    np.savez_compressed(os.path.join(OUT, "grids.npz"), theta=thd, phi=phd, e2p_maps_16=e2p_maps,
                        p2e_u_16=np.stack([a[0] for a in p2e]), p2e_v_16=np.stack([a[1] for a in p2e]),
                        p2e_mask_16=np.stack([a[2] for a in p2e]), **idx)

    # ---- 2. init_noise gather (e2p nearest) on real noise ----------------------------------------
    g = torch.Generator().manual_seed(0)
    pano_noise = torch.randn(1, 1, 4, 64, 128, generator=g)
    camd = {"FoV": torch.full((20,), 90), "theta": torch.tensor(thd), "phi": torch.tensor(phd)}
    views = ref.e2p(pano_noise.expand(-1, 20, -1, -1, -1).flatten(0, 1), camd["FoV"], camd["theta"],
                    camd["phi"], (64, 64), mode="nearest")
    np.savez_compressed(os.path.join(OUT, "init_noise.npz"), view_noise_sum=views.sum((2, 3)).numpy(),
                        view_noise_c0=views[:, 0].numpy().astype(np.float32))

    # ---- 3. EPA bias tables, coords, PE ----------------------------------------------------------
    camd90 = {"FoV": torch.full((20,), 90), "theta": torch.tensor((thd + 90) % 360), "phi": torch.tensor(phd)}
    pm, em = ref.get_masks(8, 8, 8, 16, camd90, "cpu")
    pc, ec = ref.get_coords(8, 8, 8, 16, camd90, "cpu")
    pe80 = ref.SphericalPE(80)
    pe320 = ref.SphericalPE(320)
    cam4 = {"FoV": torch.full((4,), 90), "theta": torch.tensor([0., 90, 180, 270], dtype=torch.float64),
            "phi": torch.tensor([0., 10, -20, 45], dtype=torch.float64)}
    pm4, em4 = ref.get_masks(16, 16, 16, 32, cam4, "cpu")
    nz = lambda a: (a.reshape(-1).nonzero().flatten().numpy().astype(np.int32), a.reshape(-1)[a.reshape(-1) != 0].numpy())
    pm4i, pm4v = nz(pm4 + 1)
    em4i, em4v = nz(em4 + 1)
    np.savez_compressed(os.path.join(OUT, "epa_tables.npz"), pers_masks=pm.numpy(), equi_masks=em.numpy(),
                        pers_coords=pc.numpy(), equi_coords=ec.numpy(),
                        pe80_pers=pe80(pc).numpy(), pe80_equi=pe80(ec).numpy(),
                        pe320_equi=pe320(ec).numpy(), freq80=pe80.freq_bands.numpy(),
                        freq320=pe320.freq_bands.numpy(),
                        m4_pers_idx=pm4i, m4_pers_val=pm4v, m4_equi_idx=em4i, m4_equi_val=em4v)

    # ---- 4. WarpAttn in/out (C=64) ------------------------------------------------------------------
    torch.manual_seed(3)
    wa = ref.WarpAttn(64)
    U.init_synthetic(wa, 21)
    MV.randomize_epa(wa, 22)
    gx = torch.Generator().manual_seed(4)
    px, ex = torch.randn(8, 64, 8, 8, generator=gx), torch.randn(2, 64, 8, 16, generator=gx)
    cam8 = {k: torch.cat([v, v]) for k, v in cam4.items()}
    with torch.no_grad():
        po, eo = wa(px, ex, cam8)
    np.savez_compressed(os.path.join(OUT, "warpattn_c64.npz"), pers_x=px.numpy(), equi_x=ex.numpy(),
                        pers_out=po.numpy(), equi_out=eo.numpy())

    # ---- 5. MultiViewBaseModel forward, tiny config --------------------------------------------------
    model = build_tiny_models(ref.MultiViewBaseModel)
    inp = tiny_inputs(7, 2, 4, 16, (16, 32), 7, TINY["cross_attention_dim"])
    t = torch.full((2, 4), 981, dtype=torch.long)
    camb = {k: torch.stack([v, v]) for k, v in cam4.items()}
    with torch.no_grad():
        s, ps = model(inp["latents"], inp["pano_latent"], t, inp["prompt_embd"], inp["pano_prompt_embd"], camb)
    np.savez_compressed(os.path.join(OUT, "mvgen_tiny.npz"), sample=s.numpy(), pano_sample=ps.numpy(),
                        **{k: v.numpy() for k, v in inp.items()})

    # pano-only shape (PanoOnly.py:13,39-41)
    ponly = ref.MultiViewBaseModel(None, model.pano_unet, None, None, True)
    with torch.no_grad():
        _, ps1 = ponly(None, inp["pano_latent"], torch.tensor([981, 981]), None, inp["pano_prompt_embd"], None)
    np.savez_compressed(os.path.join(OUT, "panoonly_tiny.npz"), pano_sample=ps1.numpy())

    # ---- 6. three DDIM steps through the reference model (loop glue from oracle/ddim.py) ------------
    lat, pl = inp["latents"][:1], inp["pano_latent"][:1]
    cam1 = {k: v[None] for k, v in cam4.items()}
    l3, p3 = oddim.denoise_loop(model, lat, pl, inp["prompt_embd"], inp["pano_prompt_embd"], cam1, steps=3)
    np.savez_compressed(os.path.join(OUT, "ddim3_tiny.npz"), latents=l3.numpy(), pano_latent=p3.numpy())
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
