#!/usr/bin/env python
"""From a rocprofv3 `*_kernel_trace.csv`: how much of the exchange kernels of the sequence-parallel step
(scatter = rmsnorm_rope_kernel, gather_col_blocks_kernel, peer_barrier_kernel, peer_pull_kernel) ran while an
attention kernel of the same process was running.  Writes a short markdown table.
    python tools/overlap_from_trace.py <kernel_trace.csv> <out.md>"""
import csv
import re
import sys

EXCHANGE = ("rmsnorm_rope_kernel", "gather_col_blocks_kernel", "peer_barrier_kernel", "peer_pull_kernel")


def main(path, out):
    rows = list(csv.DictReader(open(path)))
    ev = []
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        m = re.match(r"(?:void )?([A-Za-z0-9_]+)", name)
        ev.append((m.group(1) if m else name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    attn = sorted((s, e) for n, s, e in ev if n.startswith("attn_"))
    starts = [a[0] for a in attn]
    import bisect

    def overlap(s, e):
        tot = 0
        i = max(0, bisect.bisect_right(starts, s) - 1)
        while i < len(attn) and attn[i][0] < e:
            tot += max(0, min(e, attn[i][1]) - max(s, attn[i][0]))
            i += 1
        return tot

    agg = {}
    for n, s, e in ev:
        if n in EXCHANGE:
            a = agg.setdefault(n, [0, 0, 0, 0])
            ov = overlap(s, e)
            a[0] += 1
            a[1] += e - s
            a[2] += ov
            a[3] += 1 if ov > 0 else 0
    with open(out, "w") as f:
        f.write(f"# exchange kernels running beside attention kernels (`{path}`)\n\n")
        f.write(f"attention launches: {len(attn)}, {sum(e - s for s, e in attn) / 1e6:.2f} ms in all\n\n")
        f.write("| kernel | launches | total ms | ms beside an attention kernel | launches that overlap one |\n|---|---:|---:|---:|---:|\n")
        for n, a in sorted(agg.items()):
            f.write(f"| {n} | {a[0]} | {a[1] / 1e6:.3f} | {a[2] / 1e6:.3f} | {a[3]} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
