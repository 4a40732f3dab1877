set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5c/pytest.txt
bash tools/ab.sh r5c_ab 3 60 -- "default" "pf1" "pf2" "pf12" "park" "measure --opt msm_run_fill=0"
cat gpurun_out/r5c/pytest.txt gpurun_out/r5c_ab/ab.txt
