"""CPU oracle for the inexact-Lanczos shift-and-invert hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker / reported CPU
baseline.  ``eigensolvers_amd`` never imports this package.

What it restates (file:line into the read-only upstream reference):

* ``oracle.numpy_vector.RefVector``   <- numpyVector.py:23-238 (NumpyVector)
* ``oracle.lanczos_ref``               <- inexact_Lanczos.py:23-443 and the
  helpers it calls in util_funcs.py:208-385
* ``oracle.minres_ref.minres``         <- the inner solve.  The reference calls
  ``scipy.sparse.linalg.minres`` (numpyVector.py:163); SciPy is a third-party
  dependency that is NOT part of the reference tree (README.md:22 recommends
  SciPy 1.10.1, the container has 1.15.3).  The restatement follows the
  published Paige-Saunders MINRES recurrences exactly as SciPy 1.15.3
  ``_isolve/minres.py`` evaluates them, and is pinned against SciPy itself in
  ``tests/test_oracle_golden.py``.

Pinning: ``tests/golden/*.json|npz`` were produced by importing the real
reference in the build container (``tests/golden/make_golden.py``, which is the
only file that ever touches ``/root/reference``); ``tests/test_oracle_golden.py``
checks every function here against those vectors.
"""
