# build an experiment variant of the library: bash profiles/r06/sessions/mkvariant.sh <name> "<extra hipcc flags>" [sources to recompile, default: k_dp.hip k_conv_split.hip]
# -> partsbaseddetector_amd/libpbd_hip_<name>.so (the other objects are the product build's)
set -e
NAME=$1; FLAGS=$2; shift 2
SRCS=${@:-k_dp.hip k_conv_split.hip}
cd partsbaseddetector_amd/csrc
CXX="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=8"
OBJS=""
for f in pbd_api.cpp pbd_group.cpp k_pyramid.hip k_hog.hip k_conv.hip k_conv_split.hip k_dp.hip; do
  o=${f%.*}.o
  if echo " $SRCS " | grep -q " $f "; then
    o=/tmp/${f%.*}.$NAME.o
    $CXX $FLAGS -x hip -c $f -o $o 2>/dev/null
  fi
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpbd_hip_$NAME.so $OBJS -ldl
ls -la ../libpbd_hip_$NAME.so
