# round 6, session AG (GPU box): randomized differential sweeps of the final build against the oracle -- through the job entry points
# (MSPACK_PY_VIA_JOBS=1: _begin / _wait_unit / _end, a third of the units waited for one by one), with the chunk thresholds lowered so
# that batches are cut into several chunks, and with every ticket order forced
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6ag; mkdir -p $O
cd $R
run() { echo "$*" >> $O/sweeps.txt; ( env "$@" 2>&1 | tail -n 1 ) >> $O/sweeps.txt; }
for seed in 71 72 73; do
  run MSPACK_PY_VIA_JOBS=1 MSPACK_HIP_CHUNK_BYTES=65536 MSPACK_HIP_CHUNK_UNITS=16 timeout 600 python tools/sweep_lzx_frames.py $seed 60
  run MSPACK_PY_VIA_JOBS=1 MSPACK_HIP_CHUNK_BYTES=65536 MSPACK_HIP_CHUNK_UNITS=16 timeout 600 python tools/sweep_mszip_blocks.py $seed 40
  run MSPACK_PY_VIA_JOBS=1 timeout 600 python tools/sweep_lzx.py $seed
  run MSPACK_PY_VIA_JOBS=1 MSPACK_HIP_CHUNK_BYTES=65536 MSPACK_HIP_CHUNK_UNITS=16 timeout 600 python tools/sweep_mszip.py $seed
done
for ord in 0 1 2; do
  run MSPACK_HIP_TICKET_ORDER=$ord timeout 600 python tools/sweep_lzx_frames.py 74 60
done
cat $O/sweeps.txt
