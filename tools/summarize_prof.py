#!/usr/bin/env python
"""Condenses a rocprofv3 `*_kernel_stats.csv` into a short table (kernel names trimmed to
the function name) for committing under profiles/."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[^(]{0,40}>)?)", name)
    s = m.group(1) if m else name
    if s.startswith("at::native"):
        s = "torch:" + re.sub(r"^.*?([A-Za-z_]+kernel[A-Za-z_]*).*$", r"\1", name)[:48]
    return s[:60]


def main(path, out):
    rows = list(csv.DictReader(open(path)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
        a[2] = min(a[2], float(r["MinNs"]))
        a[3] = max(a[3], float(r["MaxNs"]))
    tot = sum(a[1] for a in agg.values())
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary of `{path}`\n\n")
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
            f.write(f"| {k} | {a[0]} | {a[1] / 1e6:.2f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | "
                    f"{a[3] / 1e3:.1f} | {100 * a[1] / tot:.2f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
