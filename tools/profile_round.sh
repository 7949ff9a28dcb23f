#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): profiles the default bench.py command of the CURRENT build and leaves the summaries
# under gpurun_out/<tag>/ -- copy them into profiles/ afterwards.
#   tools/profile_round.sh <tag> [bench args...]        e.g.  tools/profile_round.sh round2_v1 --scene C
# 1. rocprofv3 --kernel-trace --stats : per-kernel durations (rocpd database -> CSV + one iteration's timeline)
# 2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots")
# The PMC json carries the source id of the build it was measured on; bench.py only quotes it for that build.
set -u
TAG=${1:-round}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 20 --warmup 3 --no-cpu --no-extra $*"
python3 $REPO/bench.py --steps 20 --warmup 3 --no-extra $* > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python3 $REPO/bench.py $ARGS > $OUT/kt.log 2>&1
DB=$(find $OUT/kt -name '*_results.db' | head -1)
python3 $REPO/tools/rocpd_timeline.py "$DB" --csv $OUT/kernel_stats.csv --regime 303:20 > $OUT/timeline.txt 2>&1      # statistics over the TIMED window only (300 ramp + 3 warm-up iterations come first)
python3 $REPO/tools/rocpd_timeline.py "$DB" --csv $OUT/kernel_stats_all_launches.csv > /dev/null 2>&1
python3 $REPO/tools/rocpd_periter.py "$DB" --regime 303:20 > $OUT/per_iteration.txt 2>&1                               # one row per timed iteration, one column per kernel
# Counter collection SERIALISES the dispatches of all queues -- in an order of its own: a gate kernel on the second queue can be held back behind the very kernel it waits for,
# and every wait then runs into its 2 s limit.  The counter passes therefore keep everything on ONE queue -- but with the TIMED schedule's kernels (round 6):
# TJ_XS_ASYNC=1 TJ_XS_ONE_QUEUE=1 runs the asynchronous solve's protocol (tickets, flags, k_ccd's units building the swept-hull records) and TJ_FRONT_ASYNC_ONE_QUEUE=1 the
# asynchronous front's data flow (k_front's units forming and publishing the hull records, k_linesearch writing none) -- the traffic of the timed two-queue run, serially.
# (pmc_one_queue_chain.json: the sibling schedule TJ_XS_ASYNC=0 TJ_FRONT_ASYNC=0 -- swept-hull tail in k_xsolve, hull cache from k_linesearch -- that rounds 4 and 5 reported.)
export TJ_XS_ASYNC=1 TJ_XS_ONE_QUEUE=1 TJ_FRONT_ASYNC_ONE_QUEUE=1 TJ_KEEP_ASYNC=0
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o x -- python3 $REPO/bench.py $ARGS > $OUT/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o x -- python3 $REPO/bench.py $ARGS > $OUT/pw.log 2>&1
F=$(find $OUT/pf -name '*_results.db' | head -1); W=$(find $OUT/pw -name '*_results.db' | head -1)
SCENE=$(python3 -c "import json;print(json.load(open('$OUT/bench.json'))['config']['workload'].split(':')[0])")
python3 $REPO/tools/pmc_traffic.py "$F" "$W" --scene "$SCENE" --command "TJ_XS_ASYNC=1 TJ_XS_ONE_QUEUE=1 TJ_FRONT_ASYNC_ONE_QUEUE=1 python3 bench.py $ARGS" --regime 303:20 --out $OUT/pmc.json > $OUT/pmc.txt 2>&1      # the TIMED window only, like the kernel statistics
# 3. SQ counters in their own pass: waves, busy cycles, VALU instructions, cycles some wave waited -- "latency bound" in counters
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY -d $OUT/ps -o x -- python3 $REPO/bench.py $ARGS > $OUT/ps.log 2>&1
S=$(find $OUT/ps -name '*_results.db' | head -1)
python3 $REPO/tools/pmc_sq.py "$S" --out $OUT/sq.json > $OUT/sq.txt 2>&1
rm -rf $OUT/pf $OUT/pw
export TJ_XS_ASYNC=0 TJ_FRONT_ASYNC=0
unset TJ_XS_ONE_QUEUE TJ_FRONT_ASYNC_ONE_QUEUE
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o x -- python3 $REPO/bench.py $ARGS > $OUT/pf0.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o x -- python3 $REPO/bench.py $ARGS > $OUT/pw0.log 2>&1
F=$(find $OUT/pf -name '*_results.db' | head -1); W=$(find $OUT/pw -name '*_results.db' | head -1)
python3 $REPO/tools/pmc_traffic.py "$F" "$W" --scene "$SCENE" --command "TJ_XS_ASYNC=0 TJ_FRONT_ASYNC=0 python3 bench.py $ARGS" --regime 303:20 --out $OUT/pmc_one_queue_chain.json > $OUT/pmc_one_queue_chain.txt 2>&1
unset TJ_XS_ASYNC TJ_KEEP_ASYNC TJ_FRONT_ASYNC
python3 - "$OUT/pmc.json" <<'PYEOF'
import json, sys
try:
    d = json.load(open(sys.argv[1])); d["note"] = (d.get("note") or "") + " | counter passes ran the TIMED schedule's kernels on one queue (TJ_XS_ASYNC=1 TJ_XS_ONE_QUEUE=1 TJ_FRONT_ASYNC_ONE_QUEUE=1: counter collection serialises dispatches across queues; tickets, flags, k_ccd-built swept-hull records and -- asynchronous front -- k_front-built hull records as in the timed run); the sibling one-queue chain of rounds 4 - 5 is in pmc_one_queue_chain.json"
    json.dump(d, open(sys.argv[1], "w"), indent=1)
except Exception as e:
    print("pmc.json not annotated:", e)
PYEOF
# 4. what one wave pays per instruction (the unit of bench.py's critical_path)
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -o $OUT/issue_probe $REPO/tools/micro/issue_probe.hip 2>/dev/null && $OUT/issue_probe > $OUT/issue_probe.txt 2>&1; rm -f $OUT/issue_probe
hipcc -O3 --offload-arch=gfx950 -Wno-unused-result -o $OUT/mem_probe $REPO/tools/micro/mem_probe.hip 2>/dev/null && $OUT/mem_probe > $OUT/mem_probe.txt 2>&1; rm -f $OUT/mem_probe
rm -rf $OUT/kt $OUT/pf $OUT/pw $OUT/ps      # the databases are large; the summaries are what is kept
tail -n 3 $OUT/bench.json | cut -c1-600
head -20 $OUT/timeline.txt
