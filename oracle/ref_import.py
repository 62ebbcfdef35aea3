"""Import the reference's OWN hot-path modules from /root/reference (read-only).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Works only where
``/root/reference`` exists (the build container); the GPU box never has it, so
nothing marked ``gpu`` and nothing in bench/smoke may call this.  It is used by
``tools/make_golden.py`` to generate ``tests/golden/*`` and by the CPU tests
that pin ``oracle/*.py`` against the reference's code.

Recipe:
  * never write bytecode into the read-only tree;
  * pre-register bare ``models``, ``models.pano``, ``models.modules`` packages
    so ``models/__init__.py`` (lightning / wandb / diffusers imports) never runs;
  * shim the third-party modules that are not installed here with the
    restatements in ``oracle/third_party.py``: cv2.Rodrigues, kornia
    {remap, create_meshgrid, gaussian_blur2d}, xformers memory_efficient_attention,
    an empty skimage (pulled in by utils/pano.py -> external.PanoAnnotator);
  * reproduce the pinned numpy==1.26.4 scalar promotion: under numpy>=2,
    ``float32_array * np.radians(python_float)`` silently becomes float64.  The
    reference's geometry modules get an ``np`` proxy whose ``radians`` returns a
    Python float for scalar input (a "weak" scalar under NEP 50), which gives
    the same float32 results as numpy 1.26 value-based casting
    (e2p.py:25-26, p2e.py:25-26).
"""
import importlib
import os
import sys
import types

import numpy as _np

from . import third_party as tp

REFERENCE_ROOT = os.environ.get("PANFUSION_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models", "pano"))


class _NumpyLegacyPromotion:
    """numpy proxy: scalar ``radians`` returns a Python float (weak scalar)."""

    def __getattr__(self, name):
        return getattr(_np, name)

    @staticmethod
    def radians(x, *a, **k):
        r = _np.radians(x, *a, **k)
        if _np.ndim(r) == 0:
            return float(r)
        return r


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _install_shims():
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = _module(
            "cv2", Rodrigues=tp.rodrigues, INTER_LINEAR=1, INTER_CUBIC=2,
            INTER_NEAREST=0, INTER_AREA=3, BORDER_WRAP=3,
            remap=None)
    if "kornia" not in sys.modules:
        k = _module("kornia")
        k.__path__ = []
        ku = _module("kornia.utils", create_meshgrid=tp.create_meshgrid)
        kf = _module("kornia.filters", gaussian_blur2d=tp.gaussian_blur2d)
        kg = _module("kornia.geometry")
        kg.__path__ = []
        kgt = _module("kornia.geometry.transform", remap=tp.remap)
        k.utils, k.filters, k.geometry = ku, kf, kg
        kg.transform = kgt
        sys.modules.update({"kornia": k, "kornia.utils": ku, "kornia.filters": kf,
                            "kornia.geometry": kg, "kornia.geometry.transform": kgt})
    if "xformers" not in sys.modules:
        x = _module("xformers")
        x.__path__ = []
        xo = _module("xformers.ops", memory_efficient_attention=tp.memory_efficient_attention)
        x.ops = xo
        sys.modules.update({"xformers": x, "xformers.ops": xo})
    if "skimage" not in sys.modules:
        for n in ("skimage", "skimage.draw", "skimage.transform", "skimage.morphology",
                  "skimage.measure", "skimage.feature", "skimage.filters", "skimage.io"):
            m = _module(n)
            m.__path__ = []
            sys.modules[n] = m


_loaded = None


def load():
    """Returns a namespace with the reference's hot-path callables."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    _install_shims()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for name, sub in (("models", "models"), ("models.pano", "models/pano"),
                      ("models.modules", "models/modules")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REFERENCE_ROOT, sub)]
            sys.modules[name] = m

    e2p_mod = importlib.import_module("external.Perspective_and_Equirectangular.e2p")
    p2e_mod = importlib.import_module("external.Perspective_and_Equirectangular.p2e")
    e2p_mod.np = _NumpyLegacyPromotion()
    p2e_mod.np = _NumpyLegacyPromotion()
    pe_pkg = importlib.import_module("external.Perspective_and_Equirectangular")
    pano_utils = importlib.import_module("utils.pano")
    pm_utils = importlib.import_module("models.pano.utils")
    pm_modules = importlib.import_module("models.pano.modules")
    mvgen = importlib.import_module("models.pano.MVGenModel")
    tfm = importlib.import_module("models.modules.transformer")

    ns = types.SimpleNamespace(
        e2p=pe_pkg.e2p, p2e=pe_pkg.p2e,
        map_pers_coords_to_equi=e2p_mod.map_pers_coords_to_equi,
        map_pers_pix_to_equi=e2p_mod.map_pers_pix_to_equi,
        map_equi_pix_to_pers=p2e_mod.map_equi_pix_to_pers,
        pad_pano=pano_utils.pad_pano, unpad_pano=pano_utils.unpad_pano,
        icosahedron_sample_camera=pano_utils.icosahedron_sample_camera,
        horizon_sample_camera=pano_utils.horizon_sample_camera,
        random_sample_camera=pano_utils.random_sample_camera,
        py360_e2p=importlib.import_module("external.py360convert.e2p").e2p,
        get_masks=pm_utils.get_masks, get_coords=pm_utils.get_coords,
        WarpAttn=pm_modules.WarpAttn, MultiViewBaseModel=mvgen.MultiViewBaseModel,
        BasicTransformerBlock=tfm.BasicTransformerBlock, SphericalPE=tfm.SphericalPE,
        CrossAttention=tfm.CrossAttention,
    )
    _loaded = ns
    return ns
