python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "latency_mode or bad_pivots" 2>&1 | tail -3
python tools/lm_waves_ab.py mit_humanoid mini_cheetah 2>&1 | cut -c1-150
