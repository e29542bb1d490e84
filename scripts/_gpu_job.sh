set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_general_matrix.py -x -q 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_force_mode.py -x -q -k "general or hold or cascade or lumped or per_robot or force or square or facade" 2>&1 | tail -25
CDPR_CHUNK=0 CDPR_LOWREG=1 CDPR_STAGGER=40 python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
import cdpr_simulation_amd as pkg, bench
B=131072
model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
eng.profile_begin(); eng.update(100); ms, nl = eng.profile_end(); print("stagger 40 check: us/step", ms/100*1e3)
PY
