#!/usr/bin/env python3
"""Kernel statistics and one-iteration timeline from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME).
Usage: python tools/rocpd_timeline.py DB [--iter K] [--csv OUT]   -- stats for all tj:: kernels; timeline of the K-th k_begin..k_begin span"""
import argparse, csv, sqlite3, sys


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("db"); ap.add_argument("--iter", type=int, default=315); ap.add_argument("--csv")
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
    st = {}
    for n, s, e, _, _ in rows:
        st.setdefault(short(n), []).append(e - s)
    tot = sum(sum(v) for v in st.values())
    out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
    for n, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
        out.append((n, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / tot, 2), min(v), max(v)))
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
