#!/bin/bash
# K5's A ring depth: 4 / 6 / 8 k-tiles in flight per workgroup
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PKG=pytorch-graphsage_amd
for R in ${R5_RS:-4 6 8}; do
  /opt/rocm/bin/hipcc -DGSAGE_WP_R=$R -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -c $PKG/csrc/gsage_packed.hip -o $PKG/csrc/gsage_packed.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libgsage_hip.so $PKG/csrc/*.o || exit 1
  R5_TESTS=0 R5_CFGS="n60" bash tools/r5_ab.sh 2>&1 | sed "s/^/WP_R=$R /" | grep mfma
done
