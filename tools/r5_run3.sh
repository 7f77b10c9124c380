#!/bin/bash
# round 5, call 3: tests of the one-workgroup-per-candidate objective + the existing tuning tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_lml_wg.py tests/test_gpu_hp_tuning.py tests/test_gpu_post_sampling.py tests/test_gpu_mf_fitter.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -30 $O/pytest.txt
