cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_force_mode.py -x -q -k "scheduled" 2>&1 | tail -8
python bench.py --config 2 --no-cpu-baseline > gpurun_out/r04_bench_config2.json 2> gpurun_out/r04_bench_config2.err; tail -c 600 gpurun_out/r04_bench_config2.err; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_config2.json')); print({k: d[k] for k in ('value','ms_per_step')}, d['config']['launch_form'], d['roofline']['frac'], d['parity_check']['ok'], d.get('fused',{}).get('value'))"
python bench.py --config 2 --no-cpu-baseline --launch-per-step --no-secondary | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('launch per step', {k: d[k] for k in ('value','ms_per_step')}, d['parity_check']['ok'])"
python bench.py --no-cpu-baseline --steps 2000 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config3', {k: d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['parity_check']['ok'], 'fused', d['fused']['value'], 'rollout', d['rollout']['f32_vector_frac'], d['rollout']['parity_check']['ok'])"
