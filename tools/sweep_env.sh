#!/bin/bash
# GPU box: bench.py under a list of environment settings (one per argument, "A=1 B=2" form; "" = defaults); prints ms per frame
for E in "$@"; do
  R=$(env $E python bench.py --no-cpu --steps 60 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_median'],4), round(d['ms_per_step_min'],4))")
  echo "[$E] mean/median/min ms: $R"
done
