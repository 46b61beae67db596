#!/bin/bash
# k_icp_lat alone on the GPU (one 256-frame batch, strictly serial) by launch shape -> stdout
cd $GRAFT_REPO_ROOT
export ICP_MS_NOCHECK=1
for sh in "$@"; do
  echo "shape $sh: $(CUBOID_LAT_SHAPE=$sh timeout -k 10 100 python3 tools/icp_ms.py 256 6 2>/dev/null | tail -1)"
done
