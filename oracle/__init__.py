"""CPU oracle for the MVAE train step -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a plain torch-CPU fp32 restatement of the reference's per-batch
train step (mhw32/multimodal-vae-public: mnist/, fashionmnist/, celeba/,
celeba19/ -- model.py + the loss half of train.py).  It exists only so that
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
can check / time the HIP path against it.  Nothing under
``multimodal-vae-public_amd/`` imports it, and the product path raises when the
HIP library is missing instead of falling back to this code.

Parity pinning: the reference has no tests and no golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned by fixtures under ``tests/golden/``
that were generated in the build container by importing the *unmodified*
reference modules from /root/reference (``tests/golden/make_golden.py``) and
recording inputs, noise, per-term ELBOs, latents and gradient digests.
``tests/test_oracle_golden.py`` asserts that this restatement reproduces those
fixtures; the HIP parity tests then compare against both.

Differences from the reference, all deliberate and behaviour-preserving:
  * the reparameterisation noise and dropout masks are explicit inputs
    (the reference draws them from the global torch generator inside the model);
    when omitted they are drawn in the reference's order;
  * python-2 idioms (xrange, Variable, .data[0]) are gone.
"""
from . import functional, models, steps  # noqa: F401
