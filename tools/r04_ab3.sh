#!/bin/bash
# same-box A/B: library of the last commit (head) / current / current with persistent depth-reuse workgroups
mkdir -p gpurun_out/r04s
B="--steps 60 --warmup 15 --no-cpu-baseline --no-secondary --no-swt2net --no-h2d-leg"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B > gpurun_out/r04s/bench_$name.json 2> gpurun_out/r04s/bench_$name.err
  python -c "import json;d=json.loads(open('gpurun_out/r04s/bench_$name.json').read().strip().splitlines()[-1]);print('$name',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['ms_per_step'],d['roofline']['wgrad_ms_per_step'])"
}
run head NNZ_HIP_LIBRARY=$PWD/tools/probes/_ts/libnnuzoo_hip_head.so
run cur A=1
run persist NNZ_HIP_LIBRARY=$PWD/tools/probes/_ts/libnnuzoo_hip_persist.so
run head2 NNZ_HIP_LIBRARY=$PWD/tools/probes/_ts/libnnuzoo_hip_head.so
run cur2 A=1
run persist2 NNZ_HIP_LIBRARY=$PWD/tools/probes/_ts/libnnuzoo_hip_persist.so
timeout 600 python -m pytest -m gpu tests/test_plain_unet_gpu.py tests/test_determinism_gpu.py tests/test_conv_kernels_gpu.py -x -q > gpurun_out/r04s/t.log 2>&1; tail -2 gpurun_out/r04s/t.log
NNZ_HIP_LIBRARY=$PWD/tools/probes/_ts/libnnuzoo_hip_persist.so timeout 600 python -m pytest -m gpu tests/test_plain_unet_gpu.py -x -q > gpurun_out/r04s/t_persist.log 2>&1; tail -2 gpurun_out/r04s/t_persist.log
timeout 300 python tools/probes/conv_phase_probe.py --only enc0.1 > gpurun_out/r04s/phases.txt 2>&1
NNZ_HIP_LIBRARY=$PWD/tools/probes/_ts/libnnuzoo_hip_persist.so timeout 300 python tools/bench_conv_layers.py --innorm 1 > gpurun_out/r04s/layers_persist.txt 2>&1
timeout 300 python tools/bench_conv_layers.py --innorm 1 > gpurun_out/r04s/layers_cur.txt 2>&1
