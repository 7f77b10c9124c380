#!/bin/bash
# round 6: poisoned upper triangle (advisor), team cool-down, the whole tuning-objective test files
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_upper_triangle_unread.py tests/test_gpu_lml_wg.py tests/test_gpu_lml_fused.py -q -x 2>&1 | tail -8
