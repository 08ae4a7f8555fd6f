#!/bin/bash
# phase times inside mask_match_kernel: a -DMP_MM_TIMING build of the library (written next to the unique ids), then the timing loop
cd "$GRAFT_REPO_ROOT/maskplanner_amd/csrc"
touch mask_match.hip
make -s EXTRA=-DMP_MM_TIMING -j8 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT" && python3 tools/mask_match_time.py --phases 2>&1 | grep -v amdgpu
