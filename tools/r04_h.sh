cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
python3 -m pytest tests/test_fused_adamw_gpu.py tests/test_ssnd2net.py tests/test_zoo_trajectory_gpu.py tests/test_backends.py tests/test_graph_replay_gpu.py -q -m gpu 2>&1 | grep -v GridwiseOp | tail -30 > gpurun_out/r04h/t.log
tail -12 gpurun_out/r04h/t.log
python3 tools/bench_conv_layers.py --only enc1.0 > gpurun_out/r04h/s2_a.txt 2>&1; python3 tools/bench_conv_layers.py --only enc1.0 --tuning 7=1 >> gpurun_out/r04h/s2_a.txt 2>&1
python3 tools/bench_conv_layers.py --only enc2.0 >> gpurun_out/r04h/s2_a.txt 2>&1; python3 tools/bench_conv_layers.py --only enc2.0 --tuning 7=1 >> gpurun_out/r04h/s2_a.txt 2>&1
grep -v amdgpu gpurun_out/r04h/s2_a.txt | grep "enc\|tuning"
for M in M2Net SwT2Net SSND2Net; do
NNZ_HIP_ADAMW=0 python3 tools/bench_zoo.py --models $M --steps 8 --warmup 14 2>/dev/null | grep '"model"' | cut -c1-140
NNZ_HIP_ADAMW=1 python3 tools/bench_zoo.py --models $M --steps 8 --warmup 14 2>/dev/null | grep '"model"' | cut -c1-140
done > gpurun_out/r04h/zoo_adamw.txt
cat gpurun_out/r04h/zoo_adamw.txt
