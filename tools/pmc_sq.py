#!/usr/bin/env python3
"""Per-kernel averages of SQ counters from a rocprofv3 PMC pass (rocpd database), e.g.
    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY -d D -o x -- python3 bench.py --no-cpu --no-extra
    python tools/pmc_sq.py D/.../x_results.db --out profiles/round3_scnC_sq.json
What they say about this path: a chain kernel keeps a few hundred waves on a machine with 2 048+ wave
slots busy for its whole duration with a few thousand VALU instructions each -- SQ_INSTS_VALU / SQ_WAVES is the instruction
count of an AVERAGE wave, SQ_WAIT_INST_ANY / SQ_BUSY_CYCLES the share of busy cycles in which waves waited."""
import argparse, json, re, sqlite3


def short_name(n):
    m = re.search(r"_ZN2tj(\d+)", n)
    if not m:
        return n.replace(".kd", "")
    k = int(m.group(1)); st = m.end()
    return n[st:st + k]


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("db"); ap.add_argument("--out")
    a = ap.parse_args()
    db = sqlite3.connect(a.db); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda k: [x for x in tabs if x.startswith(k)][0]
    pm, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    # a dispatch has one row per counter INSTANCE (XCD x shader engine ...): sum them, then average over the dispatches
    q = (f"select s.kernel_name, i.name, count(distinct e.event_id), sum(e.value) from {pm} e join {ip} i on e.pmc_id=i.id "
         f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, i.name")
    out = {}
    for name, counter, n, v in cur.execute(q):
        k = short_name(name)
        if not k.startswith("k_"):
            continue
        e = out.setdefault(k, {"launches": 0})
        e[counter] = e.get(counter, 0.0) + v
        e["_n_" + counter] = e.get("_n_" + counter, 0) + n
    res = {}
    for k, e in out.items():
        r = {}
        for c in [x for x in e if not x.startswith("_") and x != "launches"]:
            r[c + "_per_launch"] = e[c] / max(1, e["_n_" + c]); r["launches"] = e["_n_" + c]
        if "SQ_INSTS_VALU_per_launch" in r and r.get("SQ_WAVES_per_launch"):
            r["valu_insts_per_wave"] = r["SQ_INSTS_VALU_per_launch"] / r["SQ_WAVES_per_launch"]
        if "SQ_WAIT_INST_ANY_per_launch" in r and r.get("SQ_BUSY_CYCLES_per_launch"):
            r["wait_over_busy"] = r["SQ_WAIT_INST_ANY_per_launch"] / r["SQ_BUSY_CYCLES_per_launch"]
        res[k] = r
    for k, r in sorted(res.items()):
        print(k, {a_: (round(b, 1) if isinstance(b, float) else b) for a_, b in r.items()})
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
