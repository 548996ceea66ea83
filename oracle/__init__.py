"""CPU oracle for the FocalFormer3D Hard-Instance-Probing decoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / reported CPU baseline.  The product
package ``focalformer3d_amd`` never imports this package and has no CPU
fallback - it raises when the HIP library is missing.
"""
