"""CPU oracle for the action-conditioned GAN hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (yidingjiang/action_conditioned_GANs) ships no tests,
fixtures or golden vectors, and its arithmetic lives in tensorflow==1.0.0
(requirements.txt:1), which is not installable here (no network, not in the wheelhouse);
the committed reference code additionally does not parse/run.
This package is therefore a *restatement* of the TF-1.0 / tf.contrib.slim semantics that
the reference composes, written from the published definitions of
those ops, and checked for internal consistency by two independent implementations:

* ``oracle.tf_ops`` / ``oracle.models`` / ``oracle.trainer`` - torch-CPU (fp64 or fp32)
  functional composition; gradients come from torch.autograd on that composition.
* ``oracle/c/acg_oracle.c`` - brute-force C loops written directly from the index
  formulas (double accumulation), exporting the same C ABI as the HIP library so it can
  stand in for the device library in CPU-only host-logic tests.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import anything from here.  The product package (``action_conditioned_gans_amd``) never
does, and fails loudly when its HIP library is missing.
"""
