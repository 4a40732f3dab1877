set -u
cd $GRAFT_REPO_ROOT
bash tools/ab.sh r5u_ab60 4 60 -- "r4" "measure" "measure --opt tail_stream=1" "measure --depth 3" "measure --opt tail_stream=1 --depth 3"
bash tools/ab.sh r5u_ab20 4 20 -- "r4" "measure" "measure --opt tail_stream=1" "measure --depth 3" "measure --opt tail_stream=1 --depth 3"
echo 60 steps; cat gpurun_out/r5u_ab60/ab.txt; echo 20 steps; cat gpurun_out/r5u_ab20/ab.txt
