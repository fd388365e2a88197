cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--smoother zebra --aniso-y 100 --no-cpu-baseline --steps 2 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/zp_trace -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/zp_f -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/zp_w -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/zp_s -- python3 $R/bench.py $ARGS > /dev/null 2>&1
echo done
