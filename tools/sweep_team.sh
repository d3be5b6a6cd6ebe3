#!/bin/bash
# experiment helper: team-kernel launch variants at a given batch (GPU box)
# (nmpc_create accepts NMPC_TEAM_OCC = 1 | 2 and NMPC_TEAM_TPW = 1 | 2 | 4 only)
B=${1:-4096}; EXTRA=${2:-}
for occ in 1 2; do
  for tpw in 4 2 1; do
    NMPC_TEAM_OCC=$occ NMPC_TEAM_TPW=$tpw timeout -k 10 120 python bench.py --steps 10 --warmup 2 --batch $B --no-cpu-baseline --no-secondary $EXTRA 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('occ $occ tpw $tpw: %.3fM solves/s, kernel %.3f ms'%(d['value']/1e6, d['roofline']['kernel_ms']))"
  done
done
