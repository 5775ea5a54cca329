#!/bin/bash
# Counter evidence for one BASELINE configuration at its per-GPU share: rocprofv3 kernel stats + separate --pmc passes (one counter set
# per pass, never combined with another trace domain) of ONE bench.py command, condensed into gpurun_out/TAG/traffic_config<C>.json
# (copied by hand to profiles/traffic_config<C>.json, which bench.py reads when the hash of the kernel sources and the ray count agree).
# usage: tools/gpu_traffic_config.sh TAG CONFIG [bench.py arguments, e.g. --shard 3/8]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; C=$2; shift; shift
OUT=gpurun_out/$TAG/config$C$TRAFFIC_SUFFIX; mkdir -p $OUT        # (TRAFFIC_SUFFIX=_aimed: a batch of another ray kind keeps its own directory and file)
export TMPDIR=/tmp
ROOT=$PWD
python -c "import __graft_entry__ as g; g.build()" || exit 1
BENCH="python $ROOT/bench.py --gpus 1 --steps 5 --warmup 2 --build-iter 1 --no-cpu-baseline --inflight 0 --no-order-compare --config $C $*"
echo "$BENCH" > $OUT/command.txt
# the un-profiled line first (its kernel_ms is what the counters are divided by when the stats pass is missing)
timeout -k 5 ${PASS_LIMIT:-900} $BENCH > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json
(cd /tmp && timeout -k 5 ${PASS_LIMIT:-900} rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/stats -o trace -- $BENCH > /dev/null 2> $ROOT/$OUT/stats.err)
# ESSENTIAL=1 leaves out the three passes the bench line does not read (SALU / LDS counts, TA busy, TCP stalls): configuration 5 takes 2.6 minutes per pass
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum"; do
  i=$((i+1))
  if [ -n "$ESSENTIAL" ] && { [ $i = 5 ] || [ $i = 7 ] || [ $i = 8 ]; }; then continue; fi
  # (pass 7, the TA set, is refused by the counter hardware of these boxes -- "exceeds the capabilities of the hardware to collect" -- and the aborted process then sat out its
  # whole time limit, a quarter of an hour per configuration, unnoticed for most of round 5: skipped; the bench line does not read it)
  if [ $i = 7 ]; then continue; fi
  (cd /tmp && timeout -k 5 ${PASS_LIMIT:-900} rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/$OUT/p$i -o pmc -- $BENCH > /dev/null 2> $ROOT/$OUT/p$i.err)
done
python tools/summarize_counters.py $OUT $C > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
