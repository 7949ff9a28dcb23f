#!/usr/bin/env python3
"""Per-iteration kernel durations from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME): one row per ADMM iteration of
the window [START, START + COUNT) (an iteration begins with k_front), one column per chain kernel, plus the iteration's span on the
profiler's clock.  Usage: python tools/rocpd_periter.py DB [--regime 303:20]     (development aid)"""
import argparse, re, sqlite3


def short(n):
    m = re.search(r"_ZN2tj(\d+)", n)
    if not m:
        return n.replace(".kd", "")
    k = int(m.group(1)); st = m.end()
    return n[st:st + k]


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("db"); ap.add_argument("--regime", default="303:20")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = [(short(n), s, e) for n, s, e in cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start")]
    r0, rc = (int(x) for x in a.regime.split(":"))
    begins = [i for i, r in enumerate(rows) if r[0] in ("k_front", "k_obs_query")]
    cols = []
    for i in range(r0, min(r0 + rc, len(begins) - 1)):
        for n, _, _ in rows[begins[i]:begins[i + 1]]:
            if n not in cols:
                cols.append(n)
    print("iter " + " ".join(f"{c[:12]:>12s}" for c in cols) + "         span")
    tot = {c: 0.0 for c in cols}; nspan = 0.0; cnt = 0
    for i in range(r0, min(r0 + rc, len(begins) - 1)):
        seg = rows[begins[i]:begins[i + 1]]
        d = {}
        for n, s, e in seg:
            d[n] = d.get(n, 0.0) + (e - s) * 1e-3
        span = (rows[begins[i + 1]][1] - seg[0][1]) * 1e-3
        print(f"{i - r0:4d} " + " ".join(f"{d.get(c, 0.0):12.1f}" for c in cols) + f" {span:12.1f}")
        for c in cols:
            tot[c] += d.get(c, 0.0)
        nspan += span; cnt += 1
    if cnt:
        print("mean " + " ".join(f"{tot[c] / cnt:12.1f}" for c in cols) + f" {nspan / cnt:12.1f}")


if __name__ == "__main__":
    main()
