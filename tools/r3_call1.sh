#!/bin/bash
# round 3, GPU call 1: true-cycle microbenchmark, 3-waves A/B, instruction-cache counters
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r3c1; mkdir -p $out
cd $R
(cd /tmp && rocprofv3 -L > $out/avail.txt 2>&1 || true)
./build/valu_issue > $out/valu_issue.txt 2>&1 || echo "microbench failed"
tools/ab.sh "" build/lib_base.so build/lib_w3.so build/lib_c3.so build/lib_s3.so build/lib_base.so > $out/ab_sponza.txt 2>&1
tools/ab.sh "--scene helmet_syn --spp 64" build/lib_base.so build/lib_w3.so > $out/ab_helmet.txt 2>&1
tools/pmc_adhoc.sh icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH > $out/icache.log 2>&1 || true
tools/pmc_adhoc.sh ifl SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS > $out/ifl.log 2>&1 || true
cat $out/ab_sponza.txt $out/ab_helmet.txt; tail -5 $out/icache.log
