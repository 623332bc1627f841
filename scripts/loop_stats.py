#!/usr/bin/env python3
"""Static instruction mix of the tableau loop of a sweep kernel, basic block by basic block (VALU, broadcast-FMAs,
SALU, no-ops, LDS, scratch, VGPR-index reads): what DESIGN.md's per-trip table is counted from.

    cd pink_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=200000 \
        -DPINKHIP_TU_NV=30 -DPINKHIP_TU_MD=0 -DPINKHIP_TU_W=32 --cuda-device-only -S tu_sweep.hip -o /tmp/k.s
    python scripts/loop_stats.py /tmp/k.s
"""
import re,sys
f=sys.argv[1]
lines=[l.rstrip() for l in open(f)]
# kernel region
ks=[i for i,l in enumerate(lines) if re.match(r'^_ZN7pinkhip\d+ik_solve_sweepx?_kernel.*:',l)][0]
ke=[i for i in range(ks,len(lines)) if lines[i].strip().startswith('s_endpgm')][0]
# find main loop header: the loop containing the most v_fmac_f64_dpp "in Loop" lines with largest header id
hdr={}
for i in range(ks,ke):
    m=re.search(r'in Loop: Header=(BB\d+_\d+)',lines[i])
    if m: hdr.setdefault(m.group(1),[]).append(i)
best=None
for h,ls in hdr.items():
    lo,hi=min(ls),max(ls)
    n=sum(1 for k in range(lo,hi) if 'v_fmac_f64_dpp' in lines[k])
    # (the tableau loop is the one that reads a column with the VGPR index mode; the stacking loops of the hand-over code
    # can span more lines)
    idx=sum(1 for k in range(lo,hi) if 's_set_gpr_idx_on' in lines[k])
    score=(idx>0,hi-lo)
    if best is None or score>best[4]: best=(h,n,lo,hi,score)
h,n,lo,hi,_=best
# extend hi to next block label after last
while hi+1<ke and not re.match(r'^(\.LBB\d+_\d+):|^; %bb\.(\d+):',lines[hi+1]): hi+=1
print("loop header",h,"lines",lo+1,hi+1,"dpp fmacs",n)
blocks=[];cur=None
for i in range(lo,hi+1):
    l=lines[i]
    m=re.match(r'^(\.LBB\d+_\d+):|^; %bb\.(\d+):',l)
    if m:
        cur={'name':(m.group(1) or 'bb.'+m.group(2)).replace('.LBB0_',''),'line':i+1,'valu':0,'dppf':0,'salu':0,'nop':0,'ds':0,'scr':0,'idx':0,'inloop':h in l,'ops':{}}
        blocks.append(cur); continue
    s=l.strip()
    if not s or s.startswith(';') or s.startswith('.') or cur is None: continue
    op=s.split()[0]
    if op.startswith('v_'):
        cur['valu']+=1
        if op=='v_fmac_f64_dpp': cur['dppf']+=1
    elif op=='s_nop': cur['nop']+=1
    elif op.startswith('ds_'): cur['ds']+=1
    elif op.startswith('scratch_'): cur['scr']+=1
    elif op=='s_set_gpr_idx_on': cur['idx']+=1
    elif op.startswith('s_'): cur['salu']+=1
    cur['ops'][op]=cur['ops'].get(op,0)+1
tot=0
for b in blocks:
    if not b['inloop']: continue
    ops=sorted(((n,o) for o,n in b['ops'].items() if o.startswith('v_') and o!='v_fmac_f64_dpp'),reverse=True)[:6]
    if b['valu'] or b['scr']:
        print(f"{b['name']:8s} L{b['line']:5d} valu {b['valu']:3d} dppfma {b['dppf']:2d} salu {b['salu']:3d} nop {b['nop']:2d} ds {b['ds']:2d} scr {b['scr']} idx {b['idx']}  {ops}")
    tot+=b['valu']
print("total VALU in loop blocks (all paths):",tot)
