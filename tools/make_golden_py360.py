"""tests/golden/py360_e2p.npz from the REFERENCE'S OWN external/py360convert (numpy + scipy, runs in the build
container): a seeded 32x64 panorama (uint8 RGB and float32 single channel), crops for cameras that hit the poles,
the +-180 degree seam and the icosahedron ring, bilinear and nearest.

    python tools/make_golden_py360.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

CAMS = [(0.0, 0.0), (36.0, 52.6226), (-144.0, -52.6226), (180.0, 10.8123), (-180.0, -10.8123), (90.0, 90.0), (0.0, -90.0),
        (72.0, 35.0)]


def main():
    e2p = importlib.import_module("external.py360convert.e2p").e2p
    rng = np.random.default_rng(3)
    rgb = (rng.random((32, 64, 3)) * 255).astype(np.uint8)
    gray = rng.standard_normal((32, 64)).astype(np.float32)
    out = dict(rgb=rgb, gray=gray, cams=np.array(CAMS))
    for mode in ("bilinear", "nearest"):
        out["rgb_" + mode] = np.stack([e2p(rgb, (90, 90), u, v, (24, 24), mode=mode) for u, v in CAMS])
        out["gray_" + mode] = np.stack([e2p(gray, (90, 90), u, v, (24, 24), mode=mode) for u, v in CAMS])
    out["rgb_fov60x45"] = np.stack([e2p(rgb, (60, 45), u, v, (18, 24)) for u, v in CAMS])
    p = os.path.join(ROOT, "tests", "golden", "py360_e2p.npz")
    np.savez_compressed(p, **out)
    print(p, os.path.getsize(p) // 1024, "KiB")


if __name__ == "__main__":
    main()
