#!/bin/bash
# round 6: the two substitutions on dedicated kernels (k_trsv_*): parity files that reach them, timings on / off;
# lml_wgf_kernel after the range cut
cd "$GRAFT_REPO_ROOT" || exit 1
for f in 0 1; do echo "== DFH_TRSV_FAST=$f"; DFH_TRSV_FAST=$f timeout 300 python tools/time_fit_sections.py 1000 4096 8192 16384; done
timeout 1500 python -m pytest tests/test_gpu_lml_fused.py tests/test_gpu_oracle_parity.py tests/test_gpu_headline.py tests/test_gpu_conditioning.py tests/test_gpu_properties.py tests/test_gpu_configs.py tests/test_gpu_incremental.py tests/test_gpu_golden.py tests/test_gpu_hp_tuning.py -q -x 2>&1 | tail -8
