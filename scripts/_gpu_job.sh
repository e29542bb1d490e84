STAMP_LIB=libcdpr_hip_stamps_ctl.so python3 scripts/stamp_probe_general.py 2>&1 | head -10
python3 scripts/stamp_probe_general.py 2>&1 | tail -30
python3 -m pytest tests/test_gpu_general_matrix.py tests/test_gpu_force_mode.py -x -q -m gpu 2>&1 | tail -4
