#!/bin/bash
# builds raytracer_amd/lib/ab_<name>.so from the current sources with extra -D flags for rt_kernels.hip: tools/mkvariant.sh <name> [-DRT_X=..]...
set -e
name=$1; shift
mkdir -p build/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC -Iinclude -Iraytracer_amd/csrc -Wno-unused-result "$@" -c raytracer_amd/csrc/rt_kernels.hip -o build/var_$name/hip_rt_kernels.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC -Iinclude -Iraytracer_amd/csrc -Wno-unused-result "$@" -c raytracer_amd/csrc/rt_capi_device.hip -o build/var_$name/hip_rt_capi_device.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o raytracer_amd/lib/ab_$name.so build/host_*.o build/var_$name/hip_rt_kernels.o build/var_$name/hip_rt_capi_device.o
echo built raytracer_amd/lib/ab_$name.so
