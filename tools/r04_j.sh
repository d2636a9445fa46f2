cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
python3 -m pytest tests/test_fused_adamw_gpu.py tests/test_param_shadow_gpu.py -q -m gpu 2>&1 | grep -v GridwiseOp | tail -40 > gpurun_out/r04j/t.log
tail -25 gpurun_out/r04j/t.log | cut -c1-250
python3 tools/probes/zoo_determinism_probe.py > gpurun_out/r04j/determinism.txt 2>&1; grep -v amdgpu gpurun_out/r04j/determinism.txt | head -40
for S in 0 1; do
NNZ_PARAM_SHADOW=$S python3 tools/bench_zoo.py --models SSND2Net,M2Net,LM2Net --steps 8 --warmup 14 2>/dev/null | grep '"model"' | cut -c1-120
done
python3 bench.py --no-swt2net --no-secondary --no-cpu-baseline 2>/dev/null | cut -c1-400
