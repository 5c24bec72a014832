#!/usr/bin/env python3
"""Show the vmem / waitcnt / branch skeleton of one kernel's loops. usage: isa_loops.py file.s kernel_regex"""
import re, sys
lines=open(sys.argv[1]).read().split('\n')
start=[i for i,l in enumerate(lines) if re.match(r'^_Z\w+:',l) and re.search(sys.argv[2],l)][0]
end=[i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end')][0]
n=0; out=[]
for l in lines[start:end]:
    t=l.strip()
    if re.match(r'^\.LBB\d+_\d+:',l): out.append(f"{n:5d} {t[:70]}")
    if not l.startswith('\t') or t.startswith('.') or t.startswith(';') or not t: continue
    n+=1
    op=t.split()[0]
    if op.startswith(('global_','s_waitcnt','s_cbranch','s_branch','scratch','s_load','s_buffer')):
        out.append(f"{n:5d} {t[:60]}")
res=[]; prev=None;cnt=0;last=''
for o in out:
    parts=o.split(); key=parts[1] if len(parts)>1 else ''
    k2='LOAD' if key.startswith('global_load') else ('STORE' if key.startswith('global_store') else ('WAITVM' if key.startswith('s_waitcnt') and 'vmcnt' in o else ('SLOAD' if key.startswith('s_load') else None)))
    if k2 and prev==k2: cnt+=1; last=o; continue
    if prev and cnt>0: res.append(f"        ... x{cnt+1} {prev} (last: {last.strip()[:44]})")
    prev=k2; cnt=0
    res.append(o)
if prev and cnt>0: res.append(f"        ... x{cnt+1} {prev}")
print('\n'.join(res))
