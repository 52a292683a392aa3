"""CPU oracle for the BigGAN train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``biggan-tensorflow_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and there only as the checker / the reported CPU baseline.

PARITY UNPINNED.  The reference (david-jk/BigGAN-Tensorflow) is TensorFlow-1.x graph
code; TensorFlow is not installed in the build container (``import tensorflow`` raises
ModuleNotFoundError), the repository has no tests, golden vectors or fixtures, and its
arithmetic lives in an un-pinned third-party dependency (TensorFlow 1.14/1.15 inferred
from ``tf.contrib`` + ``tf.gather_nd(batch_dims=1)``).  This oracle is therefore a
restatement of the reference's algorithm written from its source text plus TensorFlow's
documented op semantics, and is pinned only by
  * hand-derived known-answer tests of the TF op semantics (``oracle/kat.py``),
  * float64 finite-difference checks of its own gradients,
  * dual formulations (direct loops vs torch.nn.functional),
  * algebraic invariants.
Every function cites the reference file:line it follows.
"""
