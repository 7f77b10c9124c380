#!/bin/bash
# fused posterior mean in the strip kernel: parity + timing against the separate product
set -u
timeout 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_configs.py tests/test_gpu_oracle_parity.py tests/test_gpu_properties.py tests/test_gpu_headline.py tests/test_gpu_conditioning.py tests/test_gpu_trajectory.py -x -q -m gpu > gpurun_out/r3_tests13.log 2>&1
tail -4 gpurun_out/r3_tests13.log
for f in 1 0; do
  DFH_KM_FUSED_MEAN=$f python bench.py --no-cpu-baseline --no-c4-full --steps 2 --warmup 1 > gpurun_out/bench_mu$f.json 2> gpurun_out/bench_mu$f.err
  python - gpurun_out/bench_mu$f.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric')][0])
print(sys.argv[1], d['value'], d['sections_ms_extra_untimed_step_rank0'], 'C2', d['configs']['C2']['ms'], d['configs']['C2']['sections_ms'], 'C5', d['configs']['C5']['ms'], d['configs']['C5']['sections_ms'])
PY
done
