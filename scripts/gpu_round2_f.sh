#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2f_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r2f_tests.log
python scripts/c5_parity.py 32768 > gpurun_out/r2f_c5.log 2>&1; echo "c5 rc=$?"; tail -2 gpurun_out/r2f_c5.log
MONSOON_LANES=4 python scripts/c5_parity.py 32768 > gpurun_out/r2f_c5_u4.log 2>&1; echo "c5 u4 rc=$?"; tail -2 gpurun_out/r2f_c5_u4.log
python scripts/rollout_timing.py 16384 N12M > gpurun_out/r2f_rt_n12m.log 2>&1; tail -4 gpurun_out/r2f_rt_n12m.log
python scripts/rollout_timing.py 65536 N12M > gpurun_out/r2f_rt_n12m_64k.log 2>&1; tail -2 gpurun_out/r2f_rt_n12m_64k.log
python scripts/rollout_timing.py 16384 S12 > gpurun_out/r2f_rt_s12.log 2>&1; tail -2 gpurun_out/r2f_rt_s12.log
python scripts/rollout_timing.py 16384 random > gpurun_out/r2f_rt_random.log 2>&1; tail -2 gpurun_out/r2f_rt_random.log
