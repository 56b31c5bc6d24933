"""Static instruction mix of one kernel in a device assembly file (tools/kernel_resources.py leaves it in /tmp):
    python tools/isa_mix.py /tmp/_kr_lorahip_fast.hip.s <mangled-name substring> [<second substring>]
Prints, for the whole kernel and for every backward-branch loop body, the count per instruction class."""
import re, sys, collections
txt = open(sys.argv[1]).read().split("\n")
keys = sys.argv[2:]
start = None
for i, l in enumerate(txt):
    if l.startswith("_Z") and l.split(";")[0].strip().endswith(":") and all(k in l.split(";")[0] for k in keys):
        start = i
        break
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(txt)) if txt[i].strip().startswith("s_endpgm"))
print(txt[start])
def cls(op):
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_") and "f64" in op: return "valu_f64"
    if op.startswith("v_cndmask") or op.startswith("v_mov") or op.startswith("v_perm") or "dpp" in op or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane") or op.startswith("v_accvgpr"): return "valu_move"
    if op.startswith("v_cmp"): return "valu_cmp"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "lds_read"
    if op.startswith("ds_"): return "lds_write/other"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "vmem_rd"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_"): return "vmem_other"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_"): return "salu"
    return "other"
labels, ins = {}, []
for i in range(start + 1, end + 1):
    l = txt[i].split(";")[0].strip()
    if not l or l.startswith("."):
        if l.startswith(".LBB") and l.endswith(":"): labels[l[:-1]] = len(ins)
        continue
    if l.endswith(":"):
        labels[l[:-1]] = len(ins); continue
    ins.append(l)
def mix(a, b):
    c = collections.Counter(cls(x.split()[0]) for x in ins[a:b])
    dpp = sum(1 for x in ins[a:b] if "dpp" in x or "row_" in x or "quad_perm" in x)
    return dict(sorted(c.items())), dpp
m, d = mix(0, len(ins))
print("whole kernel: %d instructions" % len(ins), m, "dpp-modified", d)
for j, x in enumerate(ins):
    p = x.split()
    if p[0].startswith("s_cbranch") or p[0] == "s_branch":
        t = p[-1]
        if t in labels and labels[t] <= j and j - labels[t] > 40:
            m, d = mix(labels[t], j + 1)
            print("loop %s: %d instructions" % (t, j + 1 - labels[t]), m, "dpp-modified", d)
