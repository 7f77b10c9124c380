#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_mgpu.py -q -m gpu 2>&1 | tail -4
