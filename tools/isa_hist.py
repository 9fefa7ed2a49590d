#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S listing: whole kernel and its largest basic block.

usage: isa_hist.py listing.s kernel_name_substring
"""
import collections
import re
import sys


def ops(text):
    out = collections.Counter()
    for line in text.split("\n"):
        line = line.strip()
        if not line or line[0] in ";." or line.endswith(":"):
            continue
        out[line.split()[0]] += 1
    return out


def main():
    s = open(sys.argv[1]).read()
    m = re.search(r"^(_Z\w*%s\w*):.*\n" % re.escape(sys.argv[2]), s, re.M)
    if not m:
        sys.exit("kernel not found")
    body = s[m.end():s.find("s_endpgm", m.end())]
    blocks = re.split(r"^\.LBB\d+_\d+:.*\n", body, flags=re.M)
    for title, text in (("kernel " + m.group(1), body), ("largest block", max(blocks, key=len))):
        c = ops(text)
        print(title, sum(c.values()))
        for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
            print("   %-28s %d" % (k, v))


if __name__ == "__main__":
    main()
