#!/usr/bin/env python3
"""Kernel statistics and one-iteration timeline from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME).
Usage: python tools/rocpd_timeline.py DB [--iter K] [--csv OUT] [--regime START:COUNT]
  stats for all tj:: kernels; timeline of the K-th iteration.  --regime restricts the STATISTICS to the iterations
  [START, START + COUNT) of the run (an iteration begins with k_front): bench.py runs 300 untimed clock-ramp iterations with the
  stop test off, most of them at the fixed point, then W warm-up iterations, then the K timed ones -- `--regime 303:20` is the
  timed window of the default command (K = 20, W = 3), the regime the headline number describes."""
import argparse, csv, sqlite3, sys


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("db"); ap.add_argument("--iter", type=int, default=315); ap.add_argument("--csv"); ap.add_argument("--regime")
    a = ap.parse_args()
    db = sqlite3.connect(a.db); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.stream_id, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    import re
    def short(n):   # _ZN2tj5k_midILi1EEEv... -> k_mid (length-prefixed identifier after the namespace; templates included)
        m = re.search(r"_ZN2tj(\d+)", n)
        if not m:
            return n.replace(".kd", "")
        k = int(m.group(1)); st = m.end()
        return n[st:st + k]
    stat_rows = rows
    if a.regime:
        r0, rc = (int(x) for x in a.regime.split(":"))
        b_all = [i for i, r in enumerate(rows) if short(r[0]) in ("k_front", "k_obs_query")]
        if len(b_all) >= r0 + rc:
            lo = b_all[r0]; hi = b_all[r0 + rc] if len(b_all) > r0 + rc else len(rows)
            stat_rows = rows[lo:hi]
            ends = [r[2] for r in stat_rows if short(r[0]) in ("k_linesearch", "k_ls_coupled", "k_ls_commit")]   # an iteration ends with its line search (what follows the last one -- the batch's flush, copies, a gate already waiting for the next batch -- is not part of the window)
            t_end = max(ends) if ends else stat_rows[-1][2]
            print(f"# statistics over iterations [{r0}, {r0 + rc}) of the run: {len(stat_rows)} launches, span {1e-3 * (t_end - stat_rows[0][1]):.1f} us = {1e-3 * (t_end - stat_rows[0][1]) / rc:.2f} us per iteration under the profiler")
            if any(short(r[0]) == "k_xs_gate" for r in stat_rows):
                print("# asynchronous solve: k_xs_gate / k_xsolve run on a second queue; the gate's duration is one wave asleep until k_grad starts, k_xsolve's includes its blocks' sleep on the tickets,")
                print("# k_ccd's the units' sleep on the robots' flags -- durations overlap and do not add up to the iteration (see the timeline below; TJ_XS_ASYNC=0: the one-queue chain)")
            if any(short(r[0]) == "k_fa_gate" for r in stat_rows):
                print("# asynchronous front: k_fa_gate / k_front of iteration i + 1 run on the second queue NEXT TO k_linesearch of iteration i (an 'iteration' of this table still begins with its")
                print("# k_front: the k_linesearch listed in it is the one it overlaps, i.e. the previous iteration's); k_front's duration includes its units' sleep on the robots' commit flags,")
                print("# k_mid's the solve waves' sleep until k_front is through (TJ_FRONT_ASYNC=0: k_front behind k_linesearch on the chain's queue)")
        else:
            print(f"# --regime {a.regime}: the run has only {len(b_all)} iterations; statistics over all launches")
    st = {}
    for n, s, e, _, _ in stat_rows:
        nm = short(n)
        if nm in ("k_xs_gate", "k_keep_gate", "k_fa_gate"):
            nm += " [one wave asleep until the kernel it gates starts: not work; excluded from Percentage]"
        st.setdefault(nm, []).append(e - s)
    tot = sum(sum(v) for k_, v in st.items() if "gate [" not in k_)
    out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
    for n, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
        out.append((n, len(v), sum(v), round(sum(v) / len(v), 1), (0.0 if "gate [" in n else round(100.0 * sum(v) / tot, 2)), min(v), max(v)))
    for r in out: print(",".join(str(x) for x in r))
    if a.csv:
        with open(a.csv, "w", newline="") as f: csv.writer(f).writerows(out)
    # iterations start with k_front (k_begin runs once per batch: its work rides on the previous iteration's k_linesearch)
    begins = [i for i, r in enumerate(rows) if short(r[0]) in ("k_front", "k_obs_query")]
    if len(begins) > a.iter + 1:
        i0, i1 = begins[a.iter], begins[a.iter + 1]
        t0 = rows[i0][1]
        print(f"\ntimeline of iteration {a.iter} (us from its first kernel; span {1e-3 * (rows[i1][1] - t0):.1f} us):")
        for n, s, e, sid, q in rows[i0:i1]:
            print(f"  {short(n):22s} start {1e-3 * (s - t0):8.1f}  dur {1e-3 * (e - s):7.1f}  end {1e-3 * (e - t0):8.1f}  stream {sid} queue {q}")


if __name__ == "__main__":
    main()
