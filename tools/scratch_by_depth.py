"""Where a kernel's spills are: scratch_load / scratch_store instructions by LOOP DEPTH, from the assembly hipcc -S writes (LLVM annotates every
block with its loop and depth). The resident instances of the streaming kernels are at the edge of the register file: a build with 13-17 of them in the
window loop lost 20-25 % to one with 0-2 (profiles/r06/s31_scratch_in_the_window_loop.txt).
    cd lora_sdr_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I../../include -x hip --cuda-device-only -S lorahip_resident.hip -o /tmp/r.s
    python tools/scratch_by_depth.py /tmp/r.s"""
import sys,re,collections
# per kernel: count scratch_load/scratch_store per loop depth (from LLVM's "Loop Header: Depth=N" / "Parent Loop" comments)
txt=open(sys.argv[1]).read().split('\n')
kern=None; depth=0; res=collections.defaultdict(lambda: collections.Counter())
insts=collections.defaultdict(lambda: collections.Counter())
for l in txt:
    m=re.match(r'^(_Z\w+):',l)
    if m: kern=m.group(1); depth=0; continue
    if kern is None: continue
    if re.match(r'^\.LBB\d+_\d+:',l) or re.match(r'^; %bb',l.strip()):
        # new block: depth decided by following comments; reset to 0 until we see an annotation
        depth=0
    m=re.search(r'Loop Header: Depth=(\d+)',l)
    if m: depth=int(m.group(1))
    m=re.search(r'in Loop: Header=\S+ Depth=(\d+)',l)
    if m: depth=int(m.group(1))
    s=l.strip()
    if s.startswith('scratch_load') or s.startswith('scratch_store'): res[kern][depth]+=1
    if s and not s.startswith(';') and not s.startswith('.') and not s.endswith(':'): insts[kern][depth]+=1
for k in res:
    if 'demodStream' in k and k.endswith('Lb0ELb1EEEvNS_10StreamArgsE'):
        m=re.search(r'ILi(\d+)',k)
        print('SF'+m.group(1), 'scratch ops by loop depth', dict(sorted(res[k].items())), ' instructions by depth', dict(sorted(insts[k].items())))
