#!/bin/bash
# Build an A/B variant of the library with extra device-compile flags: tools/build_variant.sh NAME "-DTRACE_TOP_NODES=0 ..."
# -> pathtrace-on-cuda_amd/build_NAME/libptamd_NAME.so  (use with PTAMD_LIB=...).  Only the .hip objects are rebuilt.
set -e
NAME=$1; EXTRA=$2
PKG=$(dirname $(dirname $(readlink -f $0)))/pathtrace-on-cuda_amd
OD=$PKG/build/var_$NAME
mkdir -p $OD
FP="-ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt ${NOSLP--fno-slp-vectorize}"      # NOSLP= (empty) in the environment builds the variant with the SLP vectoriser on
for f in pt_kernels pt_wavefront pt_api pt_probe pt_comm; do
  # only pt_wavefront depends on the variant flags; reuse the others
  if [ $f = pt_wavefront ] || [ ! -f $OD/$f.o ]; then
    if [ $f = pt_wavefront ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -fvisibility=hidden $FP $EXTRA -c $PKG/csrc/$f.hip -o $OD/$f.o;
    else cp $PKG/build/$f.o $OD/$f.o; fi
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/build/libptamd_$NAME.so $PKG/build/pt_host.o $PKG/build/bvh_build.o $PKG/build/scenes.o $PKG/build/obj_loader.o $PKG/build/accel_build.o $OD/pt_kernels.o $OD/pt_wavefront.o $OD/pt_api.o $OD/pt_probe.o $OD/pt_comm.o -ldl
echo built $PKG/build/libptamd_$NAME.so
