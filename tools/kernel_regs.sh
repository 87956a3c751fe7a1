# usage: bash tools/kernel_regs.sh <file.hip> [filter]  -- VGPRs / scratch bytes / LDS / occupancy of every kernel in a translation unit
# (device-only compile to assembly; the per-kernel summary comments of the AMDGPU backend)
src=$1; filt=${2:-.}
out=/tmp/kregs_$(basename $src .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics --offload-device-only -S $src -o $out -I$(dirname $src) || exit 1
python3 - $out <<'PY' | c++filt | grep -E "$filt"
import re, sys
name = None
vals = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\S+):\s*;\s*@", line) or re.match(r"^(_Z\S+):\s*$", line)
    if m: name = m.group(1)
    m = re.match(r";\s*(NumVgprs|ScratchSize|LDSByteSize|Occupancy|NumSgprs):\s*(\d+)", line.strip())
    if m and name:
        vals.setdefault(name, {})[m.group(1)] = m.group(2)
for n, v in vals.items():
    print("%s  vgpr %s scratch %s lds %s occ %s" % (n, v.get("NumVgprs"), v.get("ScratchSize"), v.get("LDSByteSize"), v.get("Occupancy")))
PY
