import re,sys
txt=open(sys.argv[1]).read() if len(sys.argv)>1 else sys.stdin.read()
cur=None
rows=[]
for line in txt.splitlines():
    m=re.search(r'remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s*(\S+)',line)
    if not m: continue
    k,v=m.groups()
    if k=='Function Name':
        cur={'name':v}; rows.append(cur)
    elif cur is not None: cur[k]=v
for r in rows:
    print(f"{r['name'][:48]:50s} sgpr={r.get('TotalSGPRs'):>4} vgpr={r.get('VGPRs'):>4} scratch={r.get('ScratchSize [bytes/lane]'):>4} occ={r.get('Occupancy [waves/SIMD]'):>2} lds={r.get('LDS Size [bytes/block]')}")
