#!/usr/bin/env python
"""Per-basic-block instruction histogram of one kernel in a hipcc -S listing."""
import collections, re, sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] + ":")
body = s[i:]
body = body[:body.index("s_endpgm")]
lines = body.split("\n")
labels = [(n, l) for n, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l)]
for k, (n, l) in enumerate(labels):
    end = labels[k + 1][0] if k + 1 < len(labels) else len(lines)
    seg = [x.split()[0] for x in lines[n:end] if x.strip() and not x.strip().startswith((".", ";")) and not x.strip().endswith(":")]
    c = collections.Counter(seg)
    print(l.split(":")[0], sum(c.values()), dict(c.most_common(12)))
