"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the SegDistill KD hot path (reference: wzpscott/SegDistill,
``mmseg/models/distillation/losses.py``), used solely as the *checker* for the
HIP product path in ``segdistill_amd``.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import from this package.  Nothing under ``segdistill_amd/``
imports it, and the product path raises if its HIP library is missing rather
than falling back to anything here.

Pinning status: the reference's own test-suite holds NO test, golden vector or
fixture for the KD path.  The restatement is therefore
pinned against outputs of the reference itself, produced in the build
container by ``oracle/gen_golden.py`` (which imports the reference's
``losses.py`` from /root/reference) and committed as data under
``tests/golden/``.
"""
