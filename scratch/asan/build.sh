#!/bin/bash
# CPU container: the library and the two C++ drivers with the HOST code under AddressSanitizer (device code as always).
# Then on the GPU box: bash scratch/asan/run.sh   (through gpurun; the binaries travel with the tree, git ignores them)
set -eu
cd "$(dirname "$0")"
CL=/opt/rocm/lib/llvm/bin/clang++
/opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
  -o libvsf_hip.so ../../vi-slam_amd/csrc/vsf_api.hip
$CL -std=c++17 -O1 -g -fsanitize=address,undefined -I ../../include -I ../../vi-slam_amd/host ../../tests/cpp/frontend_mirror_main.cpp \
  -o frontend_mirror_asan -L . -lvsf_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib
$CL -std=c++17 -O1 -g -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I ../../include \
  ../../tests/cpp/abi_exercise_main.cpp -o abi_exercise_asan -L . -lvsf_hip -L /opt/rocm/lib -lamdhip64 \
  -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
$CL -std=c++17 -O1 -g -fPIC -shared -fsanitize=address,undefined -o libvsf_io.so ../../vi-slam_amd/csrc/vsf_io.cpp -lz
$CL -std=c++17 -O1 -g -fsanitize=address,undefined -I ../../include -I ../../vi-slam_amd/host ../../tests/cpp/euroc_example_main.cpp \
  -o euroc_example_asan -L . -lvsf_hip -lvsf_io -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib
