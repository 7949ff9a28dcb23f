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
