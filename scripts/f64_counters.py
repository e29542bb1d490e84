"""Driver for a counter pass over the role-split fp64 kernel: B robots x 8 cables, plain or with the hold branch live, 200 one-step
launches (run under rocprofv3 --pmc ...; scripts/f64_counters.sh).  argv: B eps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CDPR_NO_GRAPH"] = "1"
import cdpr_simulation_amd as pkg, bench
B, eps = int(sys.argv[1]), float(sys.argv[2])
model, pose, command, _ = bench.make_workload(pkg, B, 8, 1235, 10)
eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, precision=64, velocityEpsilon=eps), 0)
eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(200); eng.synchronize()
print(eng.kernel_name)
eng.close()
