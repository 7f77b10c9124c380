#!/bin/bash
# The three diagnostics builds behind docs/NOTES_r05.md section 2 (the scalar-wave-index anomaly of panel_fused_kernel<true>):
#   good   = dragonfly_amd/libdfhip_dbg.so as the tree builds it (python -m dragonfly_amd.build --debug-hooks)
#   bad    = the same source with `const int w = readfirstlane(tid >> 6)` at the kernel's scope
#   badnop = bad + `s_nop 4` in front of the pivot v_readlane of every owner step (f64_owner_step)
# Run in the build container from the repo root; the libraries go to tools/_exp/ (git-ignored, shipped by gpurun);
# then `gpurun -- 'bash tools/r5_run8.sh'`.
set -e
R=$(pwd); W=/tmp/nanv; mkdir -p $W $R/tools/_exp
python -m dragonfly_amd.build --debug-hooks > /dev/null
cp $R/dragonfly_amd/csrc/common.h $W/ && sed -i "s#\"../../include/dfhip.h\"#\"$R/include/dfhip.h\"#" $W/common.h
python3 - "$R" "$W" <<'PY'
import sys
R, W = sys.argv[1], sys.argv[2]
s = open(R + '/dragonfly_amd/csrc/chol.hip').read()
old = "  const int w = tid >> 6;\n  const int kq = lane >> 4, l15 = lane & 15;\n  const int g = blockIdx.x + a.g0;"
assert s.count(old) == 1
s = s.replace(old, "#ifdef DFH_EXP_SCALAR_W\n  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);\n#else\n  const int w = tid >> 6;\n#endif\n"
                   "  const int kq = lane >> 4, l15 = lane & 15;\n  const int g = blockIdx.x + a.g0;")
old = "  const int lo = __builtin_amdgcn_readlane(__double2loint(a[KL]), k);"
assert s.count(old) == 1
s = s.replace(old, "#ifdef DFH_EXP_NOP\n  __builtin_amdgcn_sched_barrier(0);\n  asm volatile(\"s_nop 4\");\n  __builtin_amdgcn_sched_barrier(0);\n#endif\n" + old)
open(W + '/chol_v.hip', 'w').write(s)
PY
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DDFH_DEBUG_HOOKS"
( cd $W && /opt/rocm/bin/hipcc $F -DDFH_EXP_SCALAR_W -c chol_v.hip -o chol_bad.o ) &
( cd $W && /opt/rocm/bin/hipcc $F -DDFH_EXP_SCALAR_W -DDFH_EXP_NOP -c chol_v.hip -o chol_badnop.o ) &
wait
objs=$(ls $R/dragonfly_amd/csrc/_obj_dbg/*.o | grep -v "/chol.o")
for v in bad badnop; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_exp/libdfhip_dbg_$v.so $objs $W/chol_$v.o; done
ls -la $R/tools/_exp/
