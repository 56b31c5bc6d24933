"""Pull one kernel's ISA out of a hipcc -save-temps .s file and summarise it.
usage: python tools/isa.py file.s <substring of mangled name> [--hist] [--mem] [--dump]"""
import re
import sys
from collections import Counter


def extract(path, key):
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S+:", l) and key in l:
            k = i
            while k + 1 < len(lines) and not lines[k + 1].startswith(".Lfunc_end"):
                k += 1
            return lines[i:k + 1]
    raise SystemExit("kernel not found")


if __name__ == "__main__":
    body = extract(sys.argv[1], sys.argv[2])
    print("lines", len(body))
    ops = [m.group(1) for l in body for m in [re.match(r"^\s+([a-z][a-z0-9_]+)", l)] if m]
    if "--hist" in sys.argv:
        for k, v in Counter(ops).most_common(40):
            print("%6d %s" % (v, k))
        print("VALU", sum(1 for o in ops if o.startswith("v_")))
    if "--mem" in sys.argv:
        for n, l in enumerate(body):
            if re.search(r"s_waitcnt vmcnt|global_load|global_store|s_cbranch|^\.LBB", l):
                print(n, l[:100])
    if "--dump" in sys.argv:
        open("/tmp/kernel.s", "w").write("\n".join(body))
