O=gpurun_out/r2o; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/gpu_tests.log
python tools/profile_config.py cfg3 400 2>&1 | grep -v amdgpu
BPM_DEMC_LPC1=1 python tools/profile_config.py cfg3 400 2>&1 | grep -v amdgpu | sed "s/^/LPC1 /"
BPM_FORCE_MODE1=1 python tools/profile_config.py cfg3 400 2>&1 | grep -v amdgpu | sed "s/^/MODE1 /"
