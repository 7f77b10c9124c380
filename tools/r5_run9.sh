#!/bin/bash
# round 5: the reference optimiser's engine calls replayed on the device (25 traces), the hallucination-ladder parity
# case with computed bounds, the tuning-objective variants
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5i; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_engine_traces.py tests/test_gpu_oracle_parity.py tests/test_gpu_lml_wg.py -q -m gpu -s > $O/pytest.txt 2>&1
grep -v "^$" $O/pytest.txt | tail -60
