#!/bin/bash
# same-box A/B of bench.py under environment settings: scripts/ab_env.sh "NAME=VAL ..." "NAME=VAL ..." ... [-- bench args]   ("-" = the default environment)
cfgs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do cfgs+=("$1"); shift; done; [ "$1" == "--" ] && shift
for c in "${cfgs[@]}"; do
  echo "== $c"
  if [ "$c" == "-" ]; then envs=""; else envs="$c"; fi
  env $envs timeout 300 python bench.py --no-cpu "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), d['unit'], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['kernels_ms'].items()}, 'replay', round(d['dispatch']['ms_per_step_replay_one_batch'],4), 'index', round(d['dispatch']['ms_per_step_rotating_index_order'],4))"
done
