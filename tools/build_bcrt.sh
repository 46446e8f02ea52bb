#!/bin/bash
# timing build of the block-cyclic-reduction kernels (in-kernel s_memtime stamps, printed by workgroup 0):
# visual-slam_amd/exp/libvslam_hip_bcrt.so; on the GPU box copy it over visual-slam_amd/libvslam_hip.so of the scratch copy
set -e
cd "$(dirname "$0")/../visual-slam_amd/csrc"
make -j8 > /dev/null
mkdir -p /tmp/bachk ../exp
(cd /tmp/bachk && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DBCR_TIMING \
  -I"$OLDPWD" -c "$OLDPWD/chol.hip" -o /tmp/bachk/chol_t.o --save-temps)
grep -E "^\s+\.(name|vgpr_count|private_segment_fixed_size|sgpr_count):" /tmp/bachk/chol-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - | grep "bcr_chol\|bcr_last_k" | sed 's/ \+/ /g'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/libvslam_hip_bcrt.so ba.o ba_fused.o bow.o /tmp/bachk/chol_t.o ctx.o describe.o detect.o keypoints_api.o match.o orb.o pgo.o vo.o
