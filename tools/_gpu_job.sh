set -u
cd $GRAFT_REPO_ROOT
bash tools/ab.sh r5l_ab 3 60 -- "measure" "measure --opt msm_fold=2" "measure --opt msm_fold=8" "measure --opt msm_fold=16" "measure --opt msm_quad_buckets=0" "measure --opt msm_unchain_lanes=10000000"
cat gpurun_out/r5l_ab/ab.txt
