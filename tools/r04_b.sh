cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
python3 -m pytest tests/test_plain_unet_gpu.py -x -q -m gpu -k "consumer_side or forward_backward_parity" > gpurun_out/r04b/t_consumer.log 2>&1
tail -15 gpurun_out/r04b/t_consumer.log
python3 -m pytest tests/test_trainer_gpu.py tests/test_determinism_gpu.py -x -q -m gpu > gpurun_out/r04b/t_trainer.log 2>&1
tail -5 gpurun_out/r04b/t_trainer.log
NNZ_CONSUMER_NORM=0 python3 bench.py --no-swt2net --no-secondary --no-cpu-baseline > gpurun_out/r04b/bench_cn0.json 2> gpurun_out/r04b/bench_cn0.err
NNZ_CONSUMER_NORM=1 python3 bench.py --no-swt2net --no-secondary --no-cpu-baseline > gpurun_out/r04b/bench_cn1.json 2> gpurun_out/r04b/bench_cn1.err
python3 -c "
import json
for f in ['bench_cn0','bench_cn1']:
    try:
        d=json.load(open('gpurun_out/r04b/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'], d['roofline']['wgrad_ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
"
tail -3 gpurun_out/r04b/bench_cn1.err
