#!/bin/bash
# knob 10 (separate statistics / reduction finishing kernels) - correctness, phases, bench A/B on one box
mkdir -p gpurun_out/r04p
timeout 900 python -m pytest -m gpu tests/test_plain_unet_gpu.py tests/test_trainer_gpu.py -x -q > gpurun_out/r04p/t.log 2>&1; tail -3 gpurun_out/r04p/t.log
timeout 900 python -m pytest -m gpu tests/test_conv_kernels_gpu.py tests/test_determinism_gpu.py tests/test_graph_replay_gpu.py tests/test_swin_umamba.py -x -q > gpurun_out/r04p/t2.log 2>&1; tail -3 gpurun_out/r04p/t2.log
for k in 0 1; do
  timeout 300 python tools/probes/conv_phase_probe.py --only enc0.1 --tuning 10=$k > gpurun_out/r04p/phases_k$k.txt 2>&1
  timeout 300 python tools/probes/conv_phase_probe.py --only enc1.1 --tuning 10=$k >> gpurun_out/r04p/phases_k$k.txt 2>&1
done
for k in 0 1 0 1; do
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-secondary --no-swt2net --no-h2d-leg --tune conv10=$k > gpurun_out/r04p/bench_k$k.json 2> gpurun_out/r04p/bench_k$k.err
  python -c "import json;d=json.loads(open('gpurun_out/r04p/bench_k$k.json').read().strip().splitlines()[-1]);print('knob10=$k',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['ms_per_step'])"
done
