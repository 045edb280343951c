#!/bin/bash
# A/B of library builds inside ONE gpurun call: tools/ab.sh "<python command>" tag1 tag2 ...   ("" = product build)
CMD="$1"; shift
for rep in 1 2; do
  for tag in base "$@"; do
    if [ "$tag" = base ]; then unset XM_LIB_PATH; else export XM_LIB_PATH=$PWD/mcncrossmodalemotions_amd/libxmodal_hip_$tag.so; fi
    echo "== $tag"
    eval "$CMD" 2>&1 | grep -v amdgpu.ids
  done
done
