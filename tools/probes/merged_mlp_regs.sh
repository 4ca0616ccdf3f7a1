#!/bin/bash
# Register / scratch bill of the weight-gradient HALF of a merged fused-MLP backward kernel, read off the ISA without a GPU:
# mlp_wgrad2_kernel<64, 4> compiled with 8 / 4 / 2 hidden tiles per wave = 512 / 256 / 128 hidden units per workgroup (2 = the product) and one wave per SIMD allowed.
# A merged kernel needs the accumulators of ALL the hidden units whose input-gradient contribution it finishes on chip, plus the dx consumer's own 16-64 accumulator
# registers and W1^T fragments on top.  Works on a patched COPY of csrc/mlp.hip (removed again: the product sources and their hash stay untouched).
#   bash tools/probes/merged_mlp_regs.sh > profiles/r06_merged_mlp_isa.txt
cd "$(dirname "$0")/../.."
tmp=mvlt_amd/csrc/_probe_mlp.hip
trap 'rm -f $tmp' EXIT
for jt in 8 4 2; do
  echo "== hidden tiles per wave: $jt (hidden units per four-wave workgroup: $((jt * 64)), one wave per SIMD allowed: 512 registers)"
  sed -e "s|constexpr int JT = 8 / NW; |constexpr int JT = $jt; |" -e "s|__launch_bounds__(NW \* 64, NW == 8 ? 2 : 2) void mlp_wgrad2_kernel|__launch_bounds__(NW * 64, 1) void mlp_wgrad2_kernel|" mvlt_amd/csrc/mlp.hip > $tmp
  grep -q "constexpr int JT = $jt;" $tmp || { echo "patch failed"; exit 1; }
  python3 tools/isa_mix.py $tmp "mlp_wgrad2_kernel<64, 4>" 2>&1 | cut -c1-220
done
