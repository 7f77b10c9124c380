"""MI355X: the factorisation's schedule variants -- trailing updates after every second panel
(DFH_CHOL_PAIR), panel strips (DFH_CHOL_STRIPS), one launch per panel (DFH_CHOL_FUSED) -- are chosen by
problem size and batch size; here each is forced on
(and off) for small sizes too, in a subprocess (the switches are read once per process), so that
every path sees ragged sizes, odd and even panel counts and lock-step batches."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

VARIANTS = {
  'defaults': {},
  'paired-everywhere': {'DFH_CHOL_PAIR': '1', 'DFH_CHOL_PAIR_MIN_REM': '0'},
  'unpaired': {'DFH_CHOL_PAIR': '0'},
  'strips-everywhere': {'DFH_CHOL_STRIPS': '1', 'DFH_CHOL_STRIPS_MIN_WG': '1'},
  'no-strips': {'DFH_CHOL_STRIPS': '0'},
  'paired+strips-everywhere': {'DFH_CHOL_PAIR_MIN_REM': '0', 'DFH_CHOL_STRIPS_MIN_WG': '1'},
  'fused-panels-in-batches': {'DFH_CHOL_FUSED': '1', 'DFH_CHOL_FUSED_MAX_BATCH': '64'},
  'fused-panels-single-only': {'DFH_CHOL_FUSED': '1', 'DFH_CHOL_FUSED_MAX_BATCH': '1'},
  'fused-panels+paired': {'DFH_CHOL_FUSED': '1', 'DFH_CHOL_PAIR_MIN_REM': '0'},
  'pivot-steps-only': {'DFH_CHOL_FUSED': '0', 'DFH_CHOL_STRIPS': '0', 'DFH_CHOL_PAIR': '0'},
  # round 3: the resident look-ahead schedule (diagonal block resident before the trailing update starts,
  # rows below by GEMM with the block inverse) from the second panel row on instead of above 7680 rows
  'resident-lookahead-everywhere': {'DFH_CHOL_LR_MIN_REM': '640'},
  'resident-lookahead+every-refinement-step': {'DFH_CHOL_LR_MIN_REM': '640', 'DFH_REFINE_TOL': '0'},
  'no-resident-lookahead': {'DFH_CHOL_LR': '0'},
  # more refinement steps due than the resident panels run on the device: second attempt by substitution
  'resident-lookahead-abandoned': {'DFH_CHOL_LR_MIN_REM': '640', 'DFH_REFINE_FORCE_STEPS': '5'},
  # every inter-workgroup wait expires at once: the status word sends the host to the schedule without hand-offs
  'forced-handoff-timeout': {'DFH_TEST_SPIN_LIMIT': '0'},
  'forced-handoff-timeout+resident': {'DFH_TEST_SPIN_LIMIT': '0', 'DFH_CHOL_LR_MIN_REM': '640'},
  'no-handoffs': {'DFH_CHOL_SAFE': '1'},
  # round 4: the one-launch panel in its round-3 form (strip rows in the accumulators' rows; fences or
  # write-through hand-offs), and the transposed form in lock-step batches / with lazy polling
  'fused-panels-round3-form': {'DFH_CHOL_FUSED_TR': '0', 'DFH_CHOL_FUSED_SC1': '0'},
  'fused-panels-round3-form+write-through': {'DFH_CHOL_FUSED_TR': '0', 'DFH_CHOL_FUSED_SC1': '1'},
  'fused-panels-round3-form-in-batches': {'DFH_CHOL_FUSED_TR': '0', 'DFH_CHOL_FUSED_MAX_BATCH': '64'},
  'split-panels': {'DFH_CHOL_FUSED_SPLIT': '1'},
  'split-panels+paired': {'DFH_CHOL_FUSED_SPLIT': '1', 'DFH_CHOL_PAIR_MIN_REM': '0'},
  'transposed-panels-lazy-polling': {'DFH_CHOL_PROG_SLEEP': '64', 'DFH_CHOL_FUSED_MAX_BATCH': '64'},
}


@pytest.mark.parametrize('name', sorted(VARIANTS))
def test_schedule_variant(engine, name):
  env = dict(os.environ)
  env.update(VARIANTS[name])
  res = subprocess.run([sys.executable, os.path.join(HERE, 'chol_paths_check.py')], env=env, capture_output=True,
                       text=True, timeout=600)
  assert res.returncode == 0 and res.stdout.strip().endswith('OK'), (res.stdout[-2000:], res.stderr[-4000:])
