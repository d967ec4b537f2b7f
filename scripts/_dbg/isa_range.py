import collections,re,sys
path,a,b=sys.argv[1],int(sys.argv[2]),int(sys.argv[3]); top=int(sys.argv[4]) if len(sys.argv)>4 else 40
lines=open(path).read().split("\n")
files={}
for l in lines:
    m=re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',l)
    if m: files[int(m.group(1))]=(m.group(3) or m.group(2)).split("/")[-1]
cur=None
# find cur loc before a
for l in lines[:a]:
    m=re.match(r"\s+\.loc\s+(\d+)\s+(\d+)",l)
    if m: cur=(files.get(int(m.group(1))),int(m.group(2)))
valu=collections.Counter();f64=collections.Counter();kinds=collections.Counter();other=collections.Counter()
for l in lines[a-1:b]:
    m=re.match(r"\s+\.loc\s+(\d+)\s+(\d+)",l)
    if m: cur=(files.get(int(m.group(1))),int(m.group(2)));continue
    m=re.match(r"\s+([a-z]\w+)",l)
    if not m: continue
    op=m.group(1)
    if op.startswith("v_mfma"): other["mfma"]+=1;continue
    if op.startswith("v_"):
        valu[cur]+=1;kinds[re.sub(r"_e32$|_e64$|_dpp$|_sdwa$","",op)+("_dpp" if "dpp" in l else "")]+=1
        if "f64" in op: f64[cur]+=1
    elif op.startswith("ds_"): other[op]+=1
    elif op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"): other[op]+=1
    elif op.startswith("s_"): other["salu"]+=1
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): other[op]+=1
print("VALU %d fp64 %d"%(sum(valu.values()),sum(f64.values())))
print("by opcode:",", ".join("%s %d"%kv for kv in kinds.most_common(30)))
print("other:",", ".join("%s %d"%kv for kv in other.most_common(30)))
for k,v in valu.most_common(top): print("%-22s:%-5d  vector %5d   fp64 %5d   other %5d"%(k[0],k[1],v,f64[k],v-f64[k]))
