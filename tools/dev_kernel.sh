#!/bin/bash
# Compiles ONE instantiation of a kernel of geoac_kernels.hip for gfx950 (seconds instead of a minute) and leaves its ISA and resource
# usage in /tmp/geoac_dev: usage  tools/dev_kernel.sh 'k_rk4<Eq3DRngDep<true,1>,false,false>'  ['k_postpass<EqGlobal<true>>' ...]
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/geoac_dev; mkdir -p $OUT
{
  echo '#define GEOAC_NO_LAUNCHERS 1'
  echo "#include \"$HERE/geoac_amd/csrc/geoac_kernels.hip\""
  for k in "$@"; do
    case "$k" in
      k_postpass_tab*) echo "template __global__ void $k(GeoacDevParams, int, int);" ;;
      k_postpass*) echo "template __global__ void $k(GeoacDevParams, int);" ;;
      *)           echo "template __global__ void $k(GeoacDevParams);" ;;
    esac
  done
} > $OUT/dev.hip
cd $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$HERE/geoac_amd/csrc" -c -o dev.o -x hip dev.hip --save-temps=obj -Rpass-analysis=kernel-resource-usage $EXTRA 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs:|VGPRs Spill|ScratchSize|Occupancy" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - -
ls $OUT/*.s | head
