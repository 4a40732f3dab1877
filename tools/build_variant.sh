#!/bin/bash
# Experiment build that differs from the product in ONE translation unit: tools/build_variant.sh <name> <unit> "<flags>" [<unit> "<flags>" ...]
#   e.g. tools/build_variant.sh a1 msm_g1 "-DZK_G1_SHAPE=1,1,2"   ->  zksnark_rs_amd/libzkgpu_a1.so  (load with ZKGPU_LIB)
set -e
cd "$(dirname "$0")/../zksnark_rs_amd/csrc"
name=$1; shift
mkdir -p _build_var
objs=""
for o in _build/*.o; do objs="$objs $o"; done
while [ $# -gt 0 ]; do
  unit=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -ffp-contract=off $flags -c $unit.hip -o _build_var/${unit}_$name.o
  objs=$(echo $objs | sed "s#_build/$unit.o#_build_var/${unit}_$name.o#")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libzkgpu_$name.so $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built libzkgpu_$name.so
