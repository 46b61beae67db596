#!/bin/bash
# k_icp_lat alone on the GPU (one 256-frame batch, strictly serial) by workgroup size -> stdout
cd $GRAFT_REPO_ROOT
export ICP_MS_NOCHECK=1
for t in 64 128 256 512 1024; do
  echo "threads $t: $(CUBOID_LAT_THREADS=$t timeout -k 10 100 python3 tools/icp_ms.py 256 6 2>/dev/null | tail -1)"
done
