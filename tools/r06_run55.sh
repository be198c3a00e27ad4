#!/bin/bash
# round 6, GPU run 55: HBM traffic of the two cubic particle kernels at C5 on the present code (FETCH_SIZE / WRITE_SIZE, separate
# passes; the figures in the C5 bench line date from round 3), and their SQ busy / wait shares
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_run55; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  echo "== $C" | tee -a $O/c5_pmc.txt
  bash tools/pmc_probe.sh "$C" bench.py --config C5 --steps 4 --warmup 2 --no-cpu-baseline --no-side-legs --no-kernel-timing 2>&1 | grep -E "k_gather_cubic|k_perm_deposit|k_cycle|k_push" | tee -a $O/c5_pmc.txt
done
