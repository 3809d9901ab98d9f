# A/B of the working tree against an older commit on ONE box (boxes differ by 1-2 %): first, here: git worktree add -f ab_old <commit> && (cd ab_old && python -c "import __graft_entry__ as g; g.build()");
# ab_old/ travels with the snapshot (keep it out of git); then: gpurun -- bash tools/ab_old_tree.sh
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for t in new old; do
    if [ $t = new ]; then d=.; else d=ab_old; fi
    (cd $d; python bench.py --steps 20 --warmup 2 --blocks 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$t', round(d['value']), round(d['ms_per_step'],2), d['ms_per_step_blocks'])")
  done
done
