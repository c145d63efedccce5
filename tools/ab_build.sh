#!/bin/bash
# tools/ab_build.sh NAME [-DFLAG=VALUE ...]   -> krisp_amd/variants/NAME.so  (run where hipcc is)
set -e
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude "$@" -o krisp_amd/variants/$NAME.so krisp_amd/csrc/krisp_hip.hip -lrccl -lz -ldl
echo built krisp_amd/variants/$NAME.so "$@"
