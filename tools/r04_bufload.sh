#!/bin/bash
# buffer-load staging (conv_box + conv_wgrad): correctness, phases, bench, layers
mkdir -p gpurun_out/r04r
timeout 1200 python -m pytest -m gpu tests/test_plain_unet_gpu.py tests/test_trainer_gpu.py tests/test_conv_kernels_gpu.py tests/test_determinism_gpu.py tests/test_graph_replay_gpu.py tests/test_rebnconv_gpu.py tests/test_swin_umamba.py -x -q > gpurun_out/r04r/t.log 2>&1; tail -3 gpurun_out/r04r/t.log
timeout 300 python tools/probes/conv_phase_probe.py --only enc0.1 > gpurun_out/r04r/phases.txt 2>&1
timeout 300 python tools/probes/conv_phase_probe.py --only enc1.1 >> gpurun_out/r04r/phases.txt 2>&1
timeout 300 python tools/probes/conv_phase_probe.py --only enc1.0 >> gpurun_out/r04r/phases.txt 2>&1
timeout 300 python tools/bench_conv_layers.py > gpurun_out/r04r/layers.txt 2>&1
timeout 300 python tools/bench_conv_layers.py --innorm 1 > gpurun_out/r04r/layers_in1.txt 2>&1
for k in 1 2; do
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-secondary --no-swt2net --no-h2d-leg > gpurun_out/r04r/bench_$k.json 2> gpurun_out/r04r/bench_$k.err
  python -c "import json;d=json.loads(open('gpurun_out/r04r/bench_$k.json').read().strip().splitlines()[-1]);print('bench',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['ms_per_step'],d['roofline']['wgrad_ms_per_step'])"
done
