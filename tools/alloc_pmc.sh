cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04/alloc_pmc; mkdir -p $O
run() { tag=$1; shift
  NOFILL=1 SEED=$tag rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$tag -o p -- python3 $R/profiles/alloc_probe.py 14 > $O/$tag.log 2>&1
  python3 $R/profiles/alloc_pmc_join.py $O/$tag > $O/${tag}_join.txt 2>&1; echo $tag done; }
run 11 TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_LATENCY_FIFO_FULL_sum
run 12 TCC_IB_STALL_sum TCC_BUSY_sum TCC_REQ_sum TCC_WRITE_sum
run 13 TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
