# round 6, session AH (GPU box): the headline batch to the DEVICE -- the chunks' shares spelled out (MSPACK_HIP_CHUNK_WEIGHTS)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6ah; mkdir -p $O
cd $R
for wts in "2,2,4,8" "1,2,4,8" "1,1,2,4,8" "1,2,3,6" "1,1,2" "1,2,4" "2,3,5" "1,1,1,3" "1,1,3,5" "1,2,5,8" "2,2,4,8"; do
  n=$(echo $wts | tr ',' '\n' | wc -l)
  echo "#### MSPACK_HIP_CHUNK_WEIGHTS=$wts" >> $O/sweep.txt
  MSPACK_HIP_CHUNK_WEIGHTS=$wts timeout 300 python tools/exp_hostpath.py 4096 5 $n 2>&1 | grep -v "to_host\|==" >> $O/sweep.txt
done
echo "#### 1024 units (config 3's size), weights as shipped / 2,2,4,8" >> $O/sweep.txt
timeout 300 python tools/exp_hostpath.py 1024 5 4 2>&1 | grep -v "to_host\|==" >> $O/sweep.txt
MSPACK_HIP_CHUNK_WEIGHTS=2,2,4,8 timeout 300 python tools/exp_hostpath.py 1024 5 4 2>&1 | grep -v "to_host\|==" >> $O/sweep.txt
echo "#### 16384 units, weights as shipped / 2,2,4,8" >> $O/sweep.txt
timeout 300 python tools/exp_hostpath.py 16384 4 4 2>&1 | grep -v "to_host\|==" >> $O/sweep.txt
MSPACK_HIP_CHUNK_WEIGHTS=2,2,4,8 timeout 300 python tools/exp_hostpath.py 16384 4 4 2>&1 | grep -v "to_host\|==" >> $O/sweep.txt
cat $O/sweep.txt
