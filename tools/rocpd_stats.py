"""Development aid: per-kernel totals from a rocprofv3 result database (the default output format of
`rocprofv3 --kernel-trace`): name, calls, total ms, average us, share.  An optional [t0_frac, t1_frac] window
selects dispatches by their position in the run's time span (e.g. 0.3 0.6 for the second of three phases).
usage: python tools/rocpd_stats.py results.db [from_frac to_frac]"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info('{ks}')")]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    rows = c.execute(f"select s.{name_col}, d.start, d.end from '{kd}' d join '{ks}' s on d.kernel_id = s.id").fetchall()
    t_lo, t_hi = min(r[1] for r in rows), max(r[2] for r in rows)
    f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
    w0, w1 = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
    agg = {}
    for n, s, e in rows:
        if s < w0 or s > w1:
            continue
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n).split("(")[0]
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    print(f"window {f0:.2f}-{f1:.2f} of {(t_hi - t_lo) / 1e6:.1f} ms span: kernel time {tot / 1e6:.1f} ms")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"| {n[:110]} | {k} | {t / 1e6:.2f} | {t / k / 1e3:.1f} | {100 * t / tot:.2f} |")


if __name__ == "__main__":
    main()
