#!/bin/bash
# the agent path at small and mid-size batches (bench.py --protocol agent --deepmind), rolled stack and plane ring: one line per case
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"
for n in 256 4096 16384; do for g in breakout space_invaders amidar gridworld; do for o in stack ring; do
  python bench.py --protocol agent --deepmind --obs $o --game $g --envs $n --steps 200 --warmup 20 2>/dev/null | G=$g N=$n O=$o python -c '
import sys, json, os
j = json.loads([l for l in sys.stdin if l.startswith("{")][-1])
print("%-15s %6s envs %-5s %8.2f M agent-steps/s  %.4f ms/step" % (os.environ["G"], os.environ["N"], os.environ["O"], j["value"] / 1e6, j["ms_per_step"]))'
done; done; done
