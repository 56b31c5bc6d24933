"""one screen of a bench.py line:  python tools/bench_digest.py gpurun_out/x_bench_default.json"""
import json, sys
lines = open(sys.argv[1]).read().strip().splitlines()
d = json.loads([ln for ln in lines if ln.startswith("{")][-1])
print("THE line: %d bytes" % len([ln for ln in lines if ln.startswith("{")][-1]))
for ln in lines:                                         # the full sections are the earlier `SECTION <name> {...}` lines (bench.py::emit)
    if ln.startswith("SECTION "):
        _, name, body = ln.split(" ", 2)
        d[name] = json.loads(body)
r = d["roofline"]
print("headline %.1f Msym/s frac %.4f launch %.1f us traffic %s src %s" % (d["value"], r["frac"], r["launch_us"], r.get("traffic"), (r.get("traffic_source") or {}).get("matches_this_tree")))
print("cpu", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores", "kind")}, "oracle", d.get("oracle"))
for e in d.get("per_sf", []):
    print("  steady SF%d %8.1f frac %.4f launch %.1f  oracle %s" % (e["sf"], e["Msym_s"], e["frac"], e["launch_us"], (e.get("oracle") or {}).get("index_mismatches")))
for e in d.get("moving", []):
    print("  moving SF%d %8.1f frac %.4f  oracle %s" % (e["sf"], e["Msym_s"], e["frac"], (e.get("oracle") or {}).get("index_mismatches")))
for e in d.get("level3", []):
    rn = e.get("running") or {}
    print("  level3 SF%d ch %d kernel %.4f unique %s e2e %.4f | running128 %s frac %s ms/work %s | chunk8 %s | mism ch %s calls %s near %s/%s" % (
        e["sf"], e["channels"], e["frac_kernel"], e.get("frac_unique"), e["frac_e2e"], rn.get("Msym_s"), rn.get("frac"), rn.get("ms_per_work"),
        (rn.get("chunk8") or {}).get("Msym_s"), e.get("oracle_channel_mismatches"), e.get("trace_call_mismatches"), e.get("near_squelch"), e.get("near_step")))
    if e.get("pothos_block"):
        pb = e["pothos_block"]
        print("      pothos_block off %s on %s cpu1 %s cpuN %s x%s" % ((pb.get("ports_off") or {}).get("Msym_s"), (pb.get("ports_on") or {}).get("Msym_s"),
              pb.get("cpu_reference_1_thread_Msym_s"), [v for k, v in pb.items() if k.startswith("cpu_reference_") and "threads" in k], pb.get("vs_cpu_same_threads")) if "error" not in pb else pb)
for k in ("level3_scaling", "mixed_level3", "config5", "mixed"):
    if k in d:
        print(" ", k, json.dumps(d[k])[:700])
