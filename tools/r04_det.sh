#!/bin/bash
mkdir -p gpurun_out/r04x
for m in deterministic off; do
  timeout 300 python tools/probes/zoo_module_determinism.py --models SwT2Net --library $m > gpurun_out/r04x/moddet_swt_$m.txt 2>&1
  grep -v "MIOpen\|amdgpu\|_benchmark" gpurun_out/r04x/moddet_swt_$m.txt | cut -c1-200 | head -6
done
for d in 0 1; do
  NNZ_LIBRARY_DETERMINISTIC=$d timeout 600 python tools/bench_zoo.py --models SwT2Net,M2Net --steps 5 --warmup 8 > gpurun_out/r04x/zoo_det$d.txt 2>&1
  grep -v "MIOpen\|amdgpu" gpurun_out/r04x/zoo_det$d.txt | cut -c1-120
done
