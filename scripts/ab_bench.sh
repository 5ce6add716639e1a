#!/bin/bash
# same-box A/B of two engine builds: scripts/ab_bench.sh <other.so> [bench args]   (the default library second)
other=$1; shift
for so in "$other" ""; do
  echo "== ${so:-default}"
  CE_ENGINE_SO=$so timeout 300 python bench.py --no-cpu "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms/step', round(d['value']), d['unit'], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['kernels_ms'].items()})"
done
