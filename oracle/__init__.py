"""
CPU oracle for the PET hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``metatrain_amd/`` (the product) may import this package. Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker / reported baseline.

Contents
--------
``nl.py``        numpy restatement of the neighbour-list *contract* visible at
                 the reference call site ``utils/neighbor_lists.py:125-201``
                 (the arithmetic lives in the un-vendored ``vesin`` wheel, pin
                 ``>=0.6.1,<0.7``): **parity unpinned** for values, see header.
``nef.py``       numpy restatement of the integer index kernels
                 ``pet/modules/nef.py:34-251`` and the integer part of
                 ``pet/modules/structures.py:285-363`` (bit-exact contract).
``pet.py``       torch-CPU (fp32/fp64) functional restatement of the PET
                 numerical core on an unpadded CSR edge layout, with autograd
                 providing dE/dR (``utils/output_gradient.py:34-40``).
                 Pinned against the reference's own regression energies
                 (``pet/tests/test_regression.py:66-74``) and against golden
                 vectors generated in the build container by importing the
                 reference (``tests/golden/make_golden.py``).
"""
