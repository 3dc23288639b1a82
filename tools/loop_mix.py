"""Instruction mix of the MFMA-heavy stretches of one kernel in a gfx950 assembly listing: the span between a label and the last backward branch to it.
Usage: python tools/loop_mix.py file.s kernel-substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
m = re.search(r"\n(_Z\w*" + sys.argv[2] + r"\w*):", s)
i = m.start(); j = s.index(".Lfunc_end", i)
body = s[i:j].splitlines()
labels = {l.split(":")[0].strip(): k for k, l in enumerate(body) if re.match(r"^\.LBB\w+:", l)}
seen = set()
for k, l in enumerate(body):
    mm = re.search(r"s_c?branch\w*\s+(\.LBB\w+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
        h = labels[mm.group(1)]
        if h in seen:
            continue
        seen.add(h)
        c = Counter(x.split()[0] for x in body[h:k + 1] if x.strip() and not x.strip().startswith((".", ";")) and not x.strip().endswith(":"))
        if sum(v for kk, v in c.items() if "mfma" in kk) == 0:
            continue
        print(mm.group(1), "lines", k - h, "instructions", sum(c.values()))
        print("   ", dict(c.most_common(70)))
