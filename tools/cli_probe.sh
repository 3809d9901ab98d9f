#!/bin/bash
# end-to-end probe of the yaha command line on the bench genome (run on the GPU box after bench.py built its cache): tools/cli_probe.sh ["opts" ...]
R=$GRAFT_REPO_ROOT; C=/tmp/yaha_bench_cache; X=$C/${YAHA_PARITY_GENOME:-g100m_s42}.X15_01_65525S
READS=$C/e2e_${YAHA_PARITY_GENOME:-g100m_s42}_n262144.fa
[ -f $READS ] || $R/tools/yaha_sim reads --genome $C/${YAHA_PARITY_GENOME:-g100m_s42}.fa --out $READS --seed 3000 --n 262144 --len 1000 --div 0.017
head -32 $READS > $C/tiny.fa
if [ $# -eq 0 ]; then set -- "-ctx 2 -batch 8192" "-ctx 3 -batch 8192" "-ctx 2 -batch 4096" "-ctx 4 -batch 4096" "-ctx 3 -batch 16384"; fi
for opts in "$@"; do
  $R/yaha_amd/csrc/yaha -x $X -q $C/tiny.fa -osh /dev/shm/tiny.sam 2> /dev/null
  s=$(date +%s%N)
  YAHA_TIMING=1 $R/yaha_amd/csrc/yaha -x $X -q $READS -osh /dev/shm/o.sam $opts 2> /dev/shm/timing.txt
  e=$(date +%s%N)
  echo "== $opts : wall $(( (e - s) / 1000000 )) ms"; grep -v ticket /dev/shm/timing.txt | tail -2; grep ticket /dev/shm/timing.txt | awk '{p+=$7; d+=$13; f+=$15; n++} END {print "batches", n, "avg parse", p/n, "device", d/n, "format", f/n, "ms"}'; grep ticket /dev/shm/timing.txt | sort -k3n | head -4 | cut -c1-140
done
