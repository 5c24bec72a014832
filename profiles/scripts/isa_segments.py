#!/usr/bin/env python3
"""Per-basic-block instruction / scratch / LDS / f64 / global counts of one kernel. usage: isa_segments.py file.s kernel_regex"""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
start=[i for i,l in enumerate(lines) if re.match(r'^_Z\w+:',l) and re.search(sys.argv[2],l)][0]
end=[i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end')][0]
seg=None; segs=[]
for l in lines[start:end]:
    t=l.strip()
    m=re.match(r'^(\.LBB\d+_\d+):(.*)',l)
    if m:
        seg={'name':m.group(1),'note':m.group(2).strip()[:50],'n':0,'scr':0,'ds':0,'f64':0,'gl':0}; segs.append(seg); continue
    if seg is None or not l.startswith('\t') or t.startswith('.') or t.startswith(';') or not t: continue
    op=t.split()[0]; seg['n']+=1
    if op.startswith('scratch'): seg['scr']+=1
    if op.startswith('ds_'): seg['ds']+=1
    if 'f64' in op: seg['f64']+=1
    if op.startswith('global'): seg['gl']+=1
for s in segs:
    if s['n']>=40 or s['scr']>0: print("%-12s n %5d scratch %4d ds %4d f64 %4d global %3d  %s"%(s['name'],s['n'],s['scr'],s['ds'],s['f64'],s['gl'],s['note']))
