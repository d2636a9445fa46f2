cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04m
for T in 1 4; do python3 tools/bench_conv_layers.py --tuning 9=$T --only 0. 2>&1 | grep -E "tuning|enc0.1|dec0.0"; done > gpurun_out/r04m/persist.txt
cat gpurun_out/r04m/persist.txt | cut -c1-160
