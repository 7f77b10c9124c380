#!/bin/bash
# kernel-trace timelines of one n = 4096 and one n = 8192 factorisation (round-4 kernels): where the time between panels goes
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/r3_trace.sh r04_n4096 4096
bash tools/r3_trace.sh r04_n8192 8192
