"""Wrap the over-long prose lines of a markdown file (not tables, not fenced code) at 140 columns, keeping list items' indentation:
    python tools/reflow_md.py DESIGN.md INTEGRATION.md README.md"""
import re, sys, textwrap
for path in sys.argv[1:]:
    out, fenced = [], False
    for line in open(path).read().split("\n"):
        if line.lstrip().startswith("```"):
            fenced = not fenced
        if fenced or line.lstrip().startswith("|") or len(line) <= 150:
            out.append(line); continue
        lead = re.match(r"\s*", line).group(0)
        m = re.match(r"\s*(\d+\.|[-*])\s+", line)
        cont = " " * len(m.group(0)) if m else lead
        out.extend(textwrap.wrap(line, width=140, subsequent_indent=cont, break_long_words=False, break_on_hyphens=False))
    open(path, "w").write("\n".join(out))
    print(path, "longest prose line now", max((len(l) for l in out if not l.lstrip().startswith("|")), default=0))
