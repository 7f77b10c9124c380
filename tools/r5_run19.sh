#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_post_sampling.py tests/test_gpu_hp_tuning.py tests/test_gpu_mgpu.py tests/test_gpu_trajectory.py -q -m gpu 2>&1 | tail -4
