#!/bin/bash
# pack kernel counters for each variant library
for lib in audio_codec_amd/liblc3plus_hip.so audio_codec_amd/_var/lib_*.so; do
  n=$(basename $lib .so)
  LC3PLUS_HIP_LIB=$PWD/$lib PMC_ONLY=1 PMC_EXTRA="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" bash tools/pmc_kernels.sh pkv_$n --no-parity > gpurun_out/pkv_$n.log 2>&1
  echo "$n $(grep -E '^pack' gpurun_out/pkv_$n.log)"
done
