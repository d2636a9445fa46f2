cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
python3 bench.py --no-swt2net > gpurun_out/r04a/bench_base.json 2> gpurun_out/r04a/bench_base.err
python3 tools/bench_conv_layers.py > gpurun_out/r04a/conv_layers_base.txt 2>&1
python3 tools/probes/ssnd2net_loss_probe.py --size 512 --steps 10 > gpurun_out/r04a/ssnd_probe_512.txt 2>&1
python3 tools/probes/ssnd2net_loss_probe.py --size 512 --steps 6 --fp32 1 > gpurun_out/r04a/ssnd_probe_512_fp32.txt 2>&1
tail -3 gpurun_out/r04a/bench_base.json | cut -c1-600
tail -12 gpurun_out/r04a/ssnd_probe_512.txt | cut -c1-400
