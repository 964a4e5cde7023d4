# round 6, session AL (GPU box): a chunk's launch that runs beside other chunks' launches asks for FEWER waves than it has tickets
# (MSPACK_HIP_CHUNK_WAVE_DIV: a launch with a wave for every ticket runs unit-major, its resolve waves waiting on their slots for the parse
# waves -- slots the next chunk's launch could use): headline batch to the device and to the host, 1024 intervals (config 3's size)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6al; mkdir -p $O
cd $R
for d in 1 2 3 4; do
  echo "#### MSPACK_HIP_CHUNK_WAVE_DIV=$d" >> $O/sweep.txt
  MSPACK_HIP_CHUNK_WAVE_DIV=$d timeout 300 python tools/exp_hostpath.py 4096 5 4 2>&1 | grep -v "==" >> $O/sweep.txt
  MSPACK_HIP_CHUNK_WAVE_DIV=$d timeout 300 python tools/exp_hostpath.py 1024 5 4 2>&1 | grep -v "==" | sed 's/^/   1024 units: /' >> $O/sweep.txt
done
echo "#### trace, MSPACK_HIP_CHUNK_WAVE_DIV=2, to the host" >> $O/sweep.txt
MSPACK_HIP_CHUNK_WAVE_DIV=2 MSPACK_HIP_TRACE=1 timeout 300 python tools/exp_hostpath.py 4096 2 4 2>&1 | grep "chunk" | tail -n 4 >> $O/sweep.txt
cat $O/sweep.txt
