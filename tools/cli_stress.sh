#!/bin/bash
# The command line many times over small batches (many hand-overs between context and filter threads): exit codes and one output checksum.
# usage: cli_stress.sh RUNS ["opts" ...]      (environment switches pass through; YGPU_CHECK_STATE=1 makes every run verify its state words on the device)
# Without option strings the runs rotate through a mix of -ctx 1..4, -batch 20..4096, contexts that are presized from the device's first one and contexts that
# grow their arenas themselves (YAHA_NO_PRESIZE), more contexts than batches (parked / idle ones) and both filters.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
W=$(mktemp -d)
python -c "
import sys, os, gzip, shutil
w = sys.argv[1]
for f in ('genome_small.fa', 'rchim.fa', 'r1k.fa'):
    with gzip.open(os.path.join('tests/golden', f + '.gz'), 'rb') as i, open(os.path.join(w, f), 'wb') as o: shutil.copyfileobj(i, o)
" $W
yaha_amd/csrc/yaha -g $W/genome_small.fa -L 11 > /dev/null 2>&1
X=$(ls $W/genome_small.X11*)
N=${1:-40}; shift
if [ $# -eq 0 ]; then set -- "-ctx 3 -batch 50" "-ctx 1 -batch 64" "-ctx 2 -batch 20 -t 4" "-ctx 4 -batch 50" "-ctx 4 -batch 4096" "-ctx 2 -batch 333" "NOPRESIZE -ctx 3 -batch 50" "NOPRESIZE -ctx 4 -batch 100" "-ctx 3 -batch 1000 -dpf N" "-ctx 1 -batch 4096"; fi
bad=0; tot=0
for i in $(seq 1 $N); do
  for opt in "$@"; do
    e=""; o="$opt"; case "$opt" in NOPRESIZE*) e="YAHA_NO_PRESIZE=1"; o="${opt#NOPRESIZE}";; esac
    env $e yaha_amd/csrc/yaha -x $X -q $W/rchim.fa -oss $W/o.sam -FBS Y $o 2> $W/err.txt; rc=$?; tot=$((tot+1))
    s=$(grep -v "^@PG" $W/o.sam | md5sum | cut -c1-12)
    if [ $rc -ne 0 ]; then echo "run $i [$opt]: exit code $rc"; grep -v "coredump\|core dump\|segment data" $W/err.txt | tail -3 | cut -c1-250; bad=$((bad+1)); fi
    echo "$s" >> $W/sums.txt
  done
done
echo "output checksums over $tot runs (one value = every run wrote the same SAM):"
sort $W/sums.txt | uniq -c
echo "failures: $bad of $tot"
rm -rf $W
