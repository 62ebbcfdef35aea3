"""How far the spherical PE of the EPA block (SphericalPE, models/modules/transformer.py:166-186) amplifies coordinate
differences: C = 320 has 80 frequencies up to 2^79, C = 64 / 128 / 256 go to 2^15 / 2^31 / 2^63.  A view-pixel
coordinate that is 0 on one side and 5e-17 on the other (axis-aligned cameras: cos(90 deg) residues whose fate the
summation order of the host BLAS decides) flips sin / cos of the high bands completely.  The product's coordinates
(pf_e2p_grid, want_lonlat) against the oracle's (reference get_coords), and the PE on both.  Needs a GPU.

    python tools/pe_sensitivity.py
"""
import sys, torch, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import cam4
from oracle import geometry as G, mvgen as MV
from panfusion_amd import ops
c = cam4()
for dim, ph, eh in ((320, 16, 16), (128, 8, 16), (320, 8, 8), (640, 16, 16)):
    blk = MV.EPABlock(dim)
    cp, ce = G.get_coords(ph, ph, eh, 2 * eh, c, dtype=torch.float32)
    _, _, ll = ops.e2p_grid(c["FoV"], c["theta"], c["phi"], eh, 2 * eh, ph, ph, "cuda", want_lonlat=True)
    d = (ll.cpu().reshape(-1, 2) - cp.reshape(-1, 2))
    print(dim, ph, eh, "pers coords: mismatching", int((d != 0).sum()), "of", d.numel(), "max abs", float(d.abs().max()))
    ec = ops.equi_coords(eh, 2 * eh, "cuda").cpu().reshape(-1, 2) - ce.reshape(-1, 2)
    print("   equi coords mismatching", int((ec != 0).sum()), "max", float(ec.abs().max()))
    freq = blk.pe.freq_bands
    pe_ref = blk.pe(cp).reshape(-1, dim)
    pe_same = ops.spherical_pe(cp.reshape(-1, 2).cuda(), freq.cuda()).cpu()      # product PE on the ORACLE's coords
    pe_prod = ops.spherical_pe(ll.view(-1, 2), freq.cuda()).cpu()
    print("   PE on identical coords: max abs diff %.3e ; PE product coords: max abs %.3e, entries > 1e-3: %d of %d"
          % (float((pe_same - pe_ref).abs().max()), float((pe_prod - pe_ref).abs().max()), int(((pe_prod - pe_ref).abs() > 1e-3).sum()), pe_ref.numel()))
