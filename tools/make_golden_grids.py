"""Round-2 grid fixtures from the REFERENCE'S OWN e2p / p2e code (same recipe as tools/make_golden.py, kept
separate so the round-1 fixtures stay byte-identical): tests/golden/grids_r2.npz

  * p2e (external/Perspective_and_Equirectangular/p2e.py:9-49) for the 20 benchmark cameras x 4 rotation
    offsets at the benchmark size (64x64 view -> 64x128 panorama) and at 8x8 -> 8x16: the visibility mask
    (bit-packed), the nearest-neighbour gather indices of (u, v) into the view, and a SHA-256 of the
    float32 u / v maps (the reference casts the float64 maps with `.type(p_img.dtype)`, p2e.py:66-67) --
    the full maps are stored only at the small size.
  * e2p (e2p.py:39-51) nearest indices at BASELINE.json configs[3]: 128x256 panorama latent -> 64x64 views.

    python tools/make_golden_grids.py        (build container only: imports /root/reference)
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import geometry as G  # noqa: E402
from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "grids_r2.npz")


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def main():
    ref = ref_import.load()
    th, ph = G.icosahedron_cameras()
    thd, phd = np.degrees(th), np.degrees(ph)
    out = dict(theta=thd, phi=phd)
    for rot in (0, 90, 180, 270):
        for name, (vh, vw, H, W) in {"64": (64, 64, 64, 128), "8": (8, 8, 8, 16)}.items():
            us, vs, ms, ii = [], [], [], []
            for i in range(20):
                u, v, mask = ref.map_equi_pix_to_pers(vh, vw, 90, (thd[i] + rot) % 360, phd[i], H, W)
                us.append(u.astype(np.float32))
                vs.append(v.astype(np.float32))
                ms.append(mask)
                ii.append(G.nearest_indices(u, v, vh, vw))
            key = "%s_rot%d" % (name, rot)
            out["p2e_idx_" + key] = np.stack(ii).astype(np.int16)
            out["p2e_mask_" + key] = np.packbits(np.stack(ms))
            out["p2e_u_sha_" + key], out["p2e_v_sha_" + key] = sha(np.stack(us)), sha(np.stack(vs))
            if name == "8":
                out["p2e_u_" + key], out["p2e_v_" + key] = np.stack(us), np.stack(vs)
        ii = []
        for i in range(20):                                  # cfg 4: 128x256 panorama latent, 64x64 views
            lon, lat = ref.map_pers_pix_to_equi(128, 256, 90, (thd[i] + rot) % 360, phd[i], 64, 64)
            ii.append(G.nearest_indices(lon, lat, 128, 256))
        out["e2p_idx_cfg4_rot%d" % rot] = np.stack(ii).astype(np.int32)
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
