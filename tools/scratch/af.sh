set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6af; mkdir -p $O
cd $R
for shape in 0 4 1 2; do
  echo "#### MSPACK_HIP_CHUNK_SHAPE=$shape" >> $O/sweep.txt
  MSPACK_HIP_CHUNK_SHAPE=$shape timeout 300 python tools/exp_hostpath.py 4096 5 3,4,6,8 2>&1 | grep -v to_host >> $O/sweep.txt
done
cat $O/sweep.txt
