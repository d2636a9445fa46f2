#!/bin/bash
mkdir -p gpurun_out/r04u
timeout 800 python tools/probes/zoo_first_nondeterminism.py --models SwT2Net,M2NetP > gpurun_out/r04u/nondet.txt 2>&1; grep -v "MIOpen" gpurun_out/r04u/nondet.txt | tail -40
timeout 900 python -m pytest -m gpu tests/test_determinism_gpu.py tests/test_plain_unet_gpu.py tests/test_graph_replay_gpu.py tests/test_trainer_gpu.py -x -q > gpurun_out/r04u/t.log 2>&1; grep -E "passed|failed" gpurun_out/r04u/t.log | tail -2
for k in 0 1; do timeout 300 python tools/probes/conv_phase_probe.py --only enc0.1 --tuning 11=$k > gpurun_out/r04u/phases_m$k.txt 2>&1; done
for k in 0 1 0 1; do
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-secondary --no-swt2net --no-h2d-leg --tune conv11=$k > gpurun_out/r04u/bench_m$k.json 2> gpurun_out/r04u/bench_m$k.err
  python -c "import json;d=json.loads(open('gpurun_out/r04u/bench_m$k.json').read().strip().splitlines()[-1]);print('knob11=$k',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['ms_per_step'])"
done
