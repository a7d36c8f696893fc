#!/bin/bash
# gather-role depth of the matrix-core seed-level launch: U = 2 / 3 / 4 work items (x 10 rows) in flight per lane
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT
PKG=pytorch-graphsage_amd
for U in ${R5_US:-2 3 4}; do
  /opt/rocm/bin/hipcc -DGSAGE_TM_ROLE_U=$U -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -c $PKG/csrc/gsage_tail_mfma.hip -o $PKG/csrc/gsage_tail_mfma.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libgsage_hip.so $PKG/csrc/*.o || exit 1
  for fr in ${R5_FRS:-50 99}; do
    R5_TESTS=0 R5_CFGS="n$fr" bash tools/r5_ab.sh 2>&1 | sed "s/^/U=$U /" | grep mfma
  done
done
python tools/kbench.py tailm 2>&1 | grep tailm
