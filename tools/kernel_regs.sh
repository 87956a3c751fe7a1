#!/bin/bash
# Register / LDS / scratch use of every kernel in an object or shared library built by csrc/Makefile:  tools/kernel_regs.sh raster.o [filter]
set -e
t=$(mktemp -d); trap "rm -rf $t" EXIT
objcopy -O binary --only-section=.hip_fatbin "$1" $t/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$t/fat.bin --output=$t/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $t/k.co | python3 -c '
import sys,re,subprocess
cur={}
rows=[]
for l in sys.stdin:
    m=re.match(r"\s+\.(\w+):\s+(.*)",l)
    if not m: continue
    k,v=m.groups()
    if k=="name":
        n=subprocess.run(["c++filt",v.strip()],capture_output=True,text=True).stdout.strip()
        cur["name"]=re.sub(r"\(anonymous namespace\)::","",n).split("(")[0]
    elif k in("vgpr_count","sgpr_count","vgpr_spill_count","sgpr_spill_count","group_segment_fixed_size","private_segment_fixed_size","agpr_count"): cur[k]=v.strip()
    if k=="vgpr_spill_count" or (k=="wavefront_size"):
        pass
    if "name" in cur and "vgpr_count" in cur and "vgpr_spill_count" in cur and "sgpr_count" in cur:
        rows.append(cur); cur={}
f=sys.argv[1] if len(sys.argv)>1 else ""
for r in rows:
    if f in r["name"]:
        print("%-60s vgpr %4s agpr %3s spill %3s sgpr %4s sspill %4s lds %6s scratch %5s"%(r["name"][:60],r["vgpr_count"],r.get("agpr_count","-"),r["vgpr_spill_count"],r["sgpr_count"],r.get("sgpr_spill_count","-"),r.get("group_segment_fixed_size","-"),r.get("private_segment_fixed_size","-")))
' "$2"
