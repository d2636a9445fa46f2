cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k
python3 -m pytest tests/test_ddp_rccl_gpu.py tests/test_bench_launch.py tests/test_trainer_gpu.py tests/test_fused_adamw_gpu.py tests/test_param_shadow_gpu.py -q -m gpu 2>&1 | grep -v GridwiseOp | tail -40 > gpurun_out/r04k/t.log
tail -25 gpurun_out/r04k/t.log | cut -c1-300
python3 bench.py --no-swt2net --no-secondary --no-cpu-baseline > gpurun_out/r04k/bench_plain.json 2>/dev/null
NNZ_BENCH_FORCE_DDP=1 python3 bench.py --no-swt2net --no-secondary --no-cpu-baseline > gpurun_out/r04k/bench_ddp_graph.json 2>gpurun_out/r04k/bench_ddp_graph.err
NNZ_BENCH_FORCE_DDP=1 NNZ_DDP_GRAPH=0 python3 bench.py --no-swt2net --no-secondary --no-cpu-baseline > gpurun_out/r04k/bench_ddp_eager.json 2>/dev/null
python3 -c "
import json
for f in ['bench_plain','bench_ddp_graph','bench_ddp_eager']:
    try:
        d=json.load(open('gpurun_out/r04k/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['hip_graph'], d.get('hip_graph_segments'), d['rccl_ranks'], d['allreduce_buckets_per_step'], d['roofline']['frac'], d.get('h2d_inclusive',{}).get('value'))
    except Exception as e: print(f, 'ERR', e)
"
tail -3 gpurun_out/r04k/bench_ddp_graph.err | cut -c1-300
