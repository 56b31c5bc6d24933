"""bench.py's level3_scaling section alone:  python tools/level3_scaling.py [--both] [--passes N] [sf counts,comma,separated] ...
--both: the persistent grid too (a build with -DLORAHIP_STREAM_PERSIST or -DLORAHIP_ALL_VARIANTS)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import lora_sdr_amd as L
class A: gpus = 1
env = bench.Env(A())
sweeps = []
args = sys.argv[1:]
both = "--both" in args
if both: args.remove("--both")
passes = 4
if "--passes" in args:
    i = args.index("--passes"); passes = int(args[i + 1]); del args[i:i + 2]
for i in range(0, len(args), 2):
    sweeps.append((int(args[i]), tuple(int(x) for x in args[i + 1].split(","))))
res = bench.section_level3_scaling(env, L, *((tuple(sweeps),) if sweeps else ()), both_grids=both, passes=passes)
for e in res:
    print(json.dumps(e))
