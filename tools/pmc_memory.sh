#!/bin/bash
# which unit of the memory pipeline the large-scene ray kernels saturate: TLB (UTCL1), address / data path (TA, TCP), L2 (TCC), fabric credits
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out; mkdir -p $OUT
CMD="python bench.py --config ${1:-3} --steps 4 --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe --no-extra-configs --sustained-seconds 0"
# NEVER RUN AS IT WAS FIRST WRITTEN: eight counters of one block in a pass made rocprofv3 abort (signal 6) and then sit in its
# finalizer until the timeout - three passes ate the last 15 GPU-minutes of round 3 without a number.  At most FOUR counters of a
# block per pass, a short timeout per pass, and the whole script stops at the first pass that yields no database.
P1="TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_STALL_INFLIGHT_MAX"
P2="TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_THRASHING_STALL TCP_UTCL1_SERIALIZATION_STALL TCP_GATE_EN1"
P3="TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES TA_DATA_STALLED_BY_TC_CYCLES"
P4="TCP_TCP_TA_ADDR_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_RFIFO_STALL_CYCLES"
P5="TCC_BUSY TCC_CYCLE TCC_TAG_STALL TCC_IB_STALL"
P6="TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_GMI_CREDIT_STALL TCC_LATENCY_FIFO_FULL TCC_SRC_FIFO_FULL"
P7="TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_LATENCY TCP_TOTAL_ACCESSES"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6" "$P7"; do
  i=$((i+1))
  rm -rf $OUT/prof_m$i
  timeout -s KILL 60 rocprofv3 --kernel-trace --pmc $P -d $OUT/prof_m$i -- $CMD > $OUT/pmc_mem_$i.log 2>&1
  DB=$(find $OUT/prof_m$i -name "*_results.db" | head -1)
  if [ -n "$DB" ]; then python tools/pmc_summary.py $DB > $OUT/pmc_mem_$i.txt; grep -A9 "k_wf_trace\|k_prepass<false" $OUT/pmc_mem_$i.txt | head -24; else echo "pass $i failed - stopping"; tail -3 $OUT/pmc_mem_$i.log; exit 1; fi
  rm -rf $OUT/prof_m$i
done
