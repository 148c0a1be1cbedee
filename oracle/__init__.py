"""CPU oracle for the PixelwiseRegression hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pixelwiseregression_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and there only as the checker / the timed CPU baseline.

Parity status: **pinned** against outputs of the reference itself, generated in the build
container by importing ``/root/reference/model.py`` (``oracle/gen_golden.py``) and committed
under ``tests/golden/``.  The reference ships no tests or golden vectors of its own
(SURVEY.md section 4), so those generated vectors are the only pins there are.
"""
