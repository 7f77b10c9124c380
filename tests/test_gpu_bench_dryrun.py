"""GPU: bench.py's N > 1 route, end to end, on whatever devices there are.  With fewer than two GPUs the
contexts share the device (DFH_MGPU_ALLOW_DUPLICATE_DEVICES: the test mode of dfh_mgpu_*, pairs reduced
on the host because RCCL refuses duplicate devices); everything else -- shard arithmetic, global indices,
replicated fit, the reduce, the JSON of a multi-GPU run -- is the code an 8-GPU node runs.  The two-way
arg-max over config 4's 2 097 152 candidates must be the one a single GPU finds over all of them
(reference: utils/oper_utils.py:73, one argmax over one array)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

# config 4 on one GPU, all 2 097 152 candidates (BENCH_r03.json configs.C4_full_1gpu, equal to the reduce over
# eight shards there; re-measured in profiles/r04_bench.json)
C4_ARGMAX = 801909


def test_two_way_strong_scaling_run_finds_the_single_gpu_argmax():
  env = dict(os.environ, DFH_MGPU_ALLOW_DUPLICATE_DEVICES='1')
  res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--no-cpu-baseline', '--no-extras'], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
  assert res.returncode == 0, res.stderr[-2000:]
  line = json.loads(res.stdout.strip().splitlines()[-1])
  assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['higher_is_better'] is False
  assert line['config']['candidates_total'] == 2097152 and line['config']['candidates_per_gpu'] == 1048576
  assert line['result']['ts_argmax'] == C4_ARGMAX
  assert line['roofline']['frac'] > 0 and line['value'] == line['ms_per_step']
