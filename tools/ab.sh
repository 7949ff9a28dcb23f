#!/bin/bash
# A/B on the GPU box: tools/ab.sh <repeats> <bench args...>; variants = the env settings listed in AB_VARIANTS (';'-separated, e.g.
# "TRAJADMM_LIB=/path/prev.so;;TJ_NO_SEQ_FOLD=1"), interleaved so that clock drift of the box hits all of them alike.
R=${1:-3}; shift
IFS=';' read -ra V <<< "${AB_VARIANTS:-;}"
for i in $(seq $R); do
  for v in "${V[@]}"; do
    ms=$(env $v python3 bench.py --no-cpu "$@" 2>/dev/null | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "[$v] $ms"
  done
done
