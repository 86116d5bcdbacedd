#!/bin/bash
# does the chip deliver random 64-byte records faster when consecutive DEPENDENT fetches fall into one small aligned
# window (a treelet) than when each is anywhere in 4 GiB? tools/micro/gather_bw.hip chain4_* against gather64, at the BVH kernel's occupancy
# (40 KiB of LDS per block = 4 waves per SIMD) and unbounded.   bash tools/gather_chain.sh <out_dir>   (needs a GPU)
out=${1:-gpurun_out/gather_chain}; mkdir -p "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/gather_bw.hip -o "$out/gather_bw" || exit 1
for lds in 40 0; do
  for pat in gather64 chain4_1k chain4_2k chain4_4k chain4_64k; do
    "$out/gather_bw" $pat 4096 64 8192 $lds
  done
done | tee "$out/gather_chain.jsonl"
