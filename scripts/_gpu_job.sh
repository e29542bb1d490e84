set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 900 python scripts/cliff_scan.py > gpurun_out/r04_cliff_scan.txt 2>&1
tail -60 gpurun_out/r04_cliff_scan.txt
