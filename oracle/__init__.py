"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the tractography env-step path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.  The product path
(``tracktolearn_amd``) never imports this package and fails loudly when the HIP
library is missing.
"""
