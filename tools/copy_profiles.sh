#!/bin/bash
# Copies what tools/collect_profiles.sh / collect_sq.sh wrote under gpurun_out/ into profiles/ (tracked), dropping the libdrm warning line:
#   bash tools/copy_profiles.sh r06
TAG=${1:-r06}
cd "$(dirname "$0")/.."
for f in ${TAG}_bench_n1.json ${TAG}_bench_n1_kernel_stats.csv ${TAG}_bench_n1_profiled_run.json ${TAG}_pmc_fetch_size.csv ${TAG}_pmc_write_size.csv \
         ${TAG}_conv_per_launch.txt ${TAG}_primary_pmc_sq_summary.json ${TAG}_ss2d_scan_bwd_pmc_all.json ${TAG}_xs_rl_bwd_kernel_hbm_traffic.json \
         ${TAG}_m2net_graph_kernels.txt ${TAG}_swt2net_graph_kernels.txt ${TAG}_ssnd2net_graph_kernels.txt ${TAG}_swt2net_pmc_sq_summary.json \
         ${TAG}_scan_bench.txt ${TAG}_conv_layers.txt ${TAG}_window_attention_bench.txt ${TAG}_swin_ops.txt ${TAG}_zoo_bench.txt \
         ${TAG}_m2net_small_op_sources.txt ${TAG}_swt2net_small_op_sources.txt ${TAG}_dice_m2netp_64_vs_oracle.json ${TAG}_full_shape_parity.txt \
         conv_box_kernel_hbm_traffic.json win_attn_hbm_traffic.json; do
  if [ -f gpurun_out/$f ]; then grep -v "amdgpu.ids" gpurun_out/$f > profiles/$f; else echo "MISSING $f"; fi
done
cp gpurun_out/${TAG}_ss2d_scan_bwd_pmc_all.json profiles/ss2d_scan_bwd_hbm_traffic.json
