#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp kalign_amd/libkalign_amd.so /tmp/reg.so
for v in reg e1 e2 e3; do
  [ $v = reg ] || { [ -f kalign_amd/libkalign_amd_$v.so ] || continue; cp kalign_amd/libkalign_amd_$v.so kalign_amd/libkalign_amd.so; }
  echo "== $v"; timeout 300 python tools/race_probe.py tree_rna16x300 150 KA_HW=1 2>&1 | tail -1 | cut -c1-300
  cp /tmp/reg.so kalign_amd/libkalign_amd.so
done
