cd $GRAFT_REPO_ROOT
cat > /tmp/t.py <<'PY'
import sys, os, numpy as np
sys.path.insert(0, '.')
import cdpr_simulation_amd as pkg, bench
for B in (16384, 65536):
    model, pose, command, _ = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=0.001), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    ts = []
    for _ in range(7):
        eng.profile_begin(); eng.update(100); ms, nl = eng.profile_end(); ts.append(ms / 100 * 1e3)
    print(os.environ.get("LABEL"), B, "us/step %.2f (min %.2f)" % (np.median(ts), min(ts)), flush=True)
    eng.close()
PY
for rep in 1 2; do
LABEL="group 1" python /tmp/t.py
LABEL="group 2" CDPR_LIB=libcdpr_hip_g2.so python /tmp/t.py
LABEL="group 4" CDPR_LIB=libcdpr_hip_g4.so python /tmp/t.py
done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
