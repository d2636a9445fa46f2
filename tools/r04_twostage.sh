#!/bin/bash
mkdir -p gpurun_out/r04y
timeout 900 python -m pytest -m gpu tests/test_two_stage_wgrads_gpu.py tests/test_token_linear_gpu.py tests/test_ss2d_cross_scan_gpu.py -x -q > gpurun_out/r04y/t.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r04y/t.log | tail -8
timeout 300 python tools/probes/zoo_module_determinism.py --models M2NetP > gpurun_out/r04y/moddet_m2.txt 2>&1; grep -v "MIOpen\|amdgpu\|_benchmark" gpurun_out/r04y/moddet_m2.txt | cut -c1-160 | head -12
for d in 0 1; do
  NNZ_TWO_STAGE_WGRADS=$d timeout 900 python tools/bench_zoo.py --models M2Net,SSND2Net,M2NetP --steps 5 --warmup 10 > gpurun_out/r04y/zoo_ts$d.txt 2>&1
  grep -v "MIOpen\|amdgpu" gpurun_out/r04y/zoo_ts$d.txt | cut -c1-110
done
timeout 1200 python -m pytest -m gpu tests/test_zoo_gpu.py tests/test_swin_umamba.py tests/test_segmamba.py -x -q > gpurun_out/r04y/t_zoo.log 2>&1; grep -E "passed|failed" gpurun_out/r04y/t_zoo.log | tail -3
