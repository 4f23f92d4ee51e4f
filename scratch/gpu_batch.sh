#!/bin/bash
# round 6, call 78: HBM traffic counters again (separate --pmc passes) with the kernel names of the final HEAD, keyed by chain family; B=32 too
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-legs --no-parity --repeats 1"
for tag in "" "_b32"; do
  extra=""; [ "$tag" = "_b32" ] && extra="--batch 32"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf$tag -o p -- $B $extra > $O/pmcf$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw$tag -o p -- $B $extra > $O/pmcw$tag.log 2>&1
  (cd $R && python scratch/pmc_traffic.py $O/pmcf$tag/p_counter_collection.csv $O/pmcw$tag/p_counter_collection.csv $O/pmc_traffic_r06$tag.json > $O/pmc_traffic_r06$tag.txt 2>&1)
  rm -rf $O/pmcf$tag $O/pmcw$tag
done
head -14 $O/pmc_traffic_r06.txt | cut -c1-150; python -c "
import json; j=json.load(open('$O/pmc_traffic_r06.json')); print(j['chain_by_family'])"
head -8 $O/pmc_traffic_r06_b32.txt | cut -c1-150
