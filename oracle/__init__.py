"""CPU oracle for the PanFusion denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
The product path (``panfusion_amd``) never imports this package and fails
loudly when its HIP extension is missing.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md §4, §8c).  The restatements here are pinned instead against the
reference's OWN code, imported from ``/root/reference`` in the build container
by ``oracle/ref_import.py`` (third-party symbols that are not installed --
kornia, cv2, xformers, diffusers -- are restated from their pinned versions'
published behaviour, so that part is "parity unpinned" and says so in
DESIGN.md), and against the committed fixtures in ``tests/golden/`` that
``tools/make_golden.py`` generated from that import.
"""
