cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
python3 -m pytest tests/test_ssnd2net.py tests/test_segmamba.py -q -m gpu 2>&1 | grep -v GridwiseOp | tail -40 > gpurun_out/r04g/t_new.log
tail -30 gpurun_out/r04g/t_new.log
python3 -m pytest tests -q -m gpu -x --deselect tests/test_ssnd2net.py --deselect tests/test_segmamba.py 2>&1 | grep -v GridwiseOp | tail -15 > gpurun_out/r04g/t_all.log
tail -8 gpurun_out/r04g/t_all.log
