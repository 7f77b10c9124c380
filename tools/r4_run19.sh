#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_hp_tuning.py tests/test_gpu_chol_paths.py tests/test_gpu_post_sampling.py tests/test_gpu_mf_fitter.py tests/test_gpu_golden.py tests/test_gpu_trajectory.py -m gpu -q ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
timeout 300 python tools/time_lml_batch.py > $O/lml_batch_default.txt 2>&1
timeout 300 python tests/gpu_check.py hptune > $O/hptune.txt 2>&1
tail -5 $O/tests.log; grep "nb= 32\|nb= 64" $O/lml_batch_default.txt; tail -12 $O/hptune.txt
