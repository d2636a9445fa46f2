cd $GRAFT_REPO_ROOT
NNZ_SEPCONV32=0 bash tools/zoo_prof1.sh r06d_old SwT2Net > /dev/null
bash tools/zoo_prof1.sh r06d_new SwT2Net > /dev/null
grep -h model gpurun_out/r06d_old_swt2net_bench.txt gpurun_out/r06d_new_swt2net_bench.txt | cut -c1-110
head -1 gpurun_out/r06d_old_swt2net_graph_kernels.txt gpurun_out/r06d_new_swt2net_graph_kernels.txt
