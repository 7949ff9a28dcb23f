#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the bench lines DESIGN.md / README.md quote, one JSON file each, under gpurun_out/<tag>/ -- copy them
# into profiles/<tag>_bench_<name>.json afterwards.     tools/bench_set.sh <tag>
# default = the driver's command (headline + extra.configs + cpu_baseline); the others are single lines (--no-cpu --no-extra).
set -u
TAG=${1:-round}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
run() { name=$1; shift; python3 bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || echo "bench $name failed (see $OUT/bench_$name.err)"; tail -n 1 "$OUT/bench_$name.json" | cut -c1-160; }
run default
Q="--no-cpu --no-extra"
run C $Q --scene C
run C100 $Q --scene C --steps 100
run A $Q --scene A
run B $Q --scene B
run D $Q --scene D
run Dtri $Q --scene Dtri
run E $Q --scene E
run Bc $Q --scene B --coupled
run Cc $Q --scene C --coupled
run Coptplane $Q --scene C --optimal-plane
run Cdist1 $Q --scene C --force-dist
run Cgroup2same $Q --scene C --group-devices 0,0
run H8 $Q --scene H8
# the asynchronous front's A/B (round 6): the same lines with k_front behind k_linesearch on the chain's queue
export TJ_FRONT_ASYNC=0
run C_fa0 $Q --scene C
run C100_fa0 $Q --scene C --steps 100
run A_fa0 $Q --scene A
run B_fa0 $Q --scene B
run Dtri_fa0 $Q --scene Dtri
run E_fa0 $Q --scene E
run Bc_fa0 $Q --scene B --coupled
run Cc_fa0 $Q --scene C --coupled
run H8_fa0 $Q --scene H8
unset TJ_FRONT_ASYNC
