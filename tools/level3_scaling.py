"""bench.py's level3_scaling section alone:  python tools/level3_scaling.py [sf counts,comma,separated] ..."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import lora_sdr_amd as L
class A: gpus = 1
env = bench.Env(A())
sweeps = []
args = sys.argv[1:]
for i in range(0, len(args), 2):
    sweeps.append((int(args[i]), tuple(int(x) for x in args[i + 1].split(","))))
res = bench.section_level3_scaling(env, L, *( (tuple(sweeps),) if sweeps else ()))
for e in res:
    print(json.dumps(e))
