#!/bin/bash
# Diagnostic: build variants of the kernel object with extra -D flags into audio_codec_amd/_var/ (git-ignored; travels with gpurun).
# usage: tools/variants.sh name1 "-DFOO=1" name2 "-DBAR=2" ...      then on the GPU box: tools/variants_run.sh
set -e
cd "$(dirname "$0")/../audio_codec_amd/csrc"
mkdir -p ../_var
FLAGS="-Os -ffp-contract=off --offload-arch=gfx950 -fPIC -Wno-unused-value"
while [ $# -ge 2 ]; do
  n=$1; d=$2; shift 2
  ( hipcc $FLAGS $d -c lc3_kernels.hip -o ../_var/k_$n.o 2>../_var/build_$n.log && hipcc --offload-arch=gfx950 -shared -fPIC -o ../_var/lib_$n.so ../_var/k_$n.o lc3_kernels_big.o lc3_host.o -lm && rm ../_var/k_$n.o && echo built $n ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
