set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputest1.log
tail -5 gpurun_out/r02_gputest1.log
bash tools/size_sweep.sh > gpurun_out/r02_size_sweep1.txt 2>&1
bash tools/config_sweep.sh > gpurun_out/r02_config_sweep1.txt 2>&1
cat gpurun_out/r02_size_sweep1.txt gpurun_out/r02_config_sweep1.txt
