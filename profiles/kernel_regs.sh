#!/bin/bash
# registers, scratch and LDS of every kernel of the built library (the code object's metadata); usage: profiles/kernel_regs.sh [lib] [filter]
LIB=${1:-moni_align_amd/csrc/libmoni_hip.so}; PAT=${2:-.}
T=$(mktemp -d)
python3 - "$LIB" "$T/dev.co" <<'PY'
import struct, sys
so = open(sys.argv[1], "rb").read()
i = so.find(b"__CLANG_OFFLOAD_BUNDLE__"); n = struct.unpack_from("<Q", so, i + 24)[0]; o = i + 32
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", so, o); o += 24
    t = so[o:o + tl].decode(); o += tl
    if "gfx950" in t: open(sys.argv[2], "wb").write(so[i + off:i + off + size])
PY
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | python3 -c '
import sys, re
cur = {}
rows = []
for ln in sys.stdin:
    m = re.match(r"\s+\.(\w+):\s+(.*)", ln)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == "name" and v.startswith("_Z") or k == "name" and not v.startswith("'"'"'") and "kernel" in v: cur["name"] = v
    if k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count"): cur[k] = v
    if k == "wavefront_size": rows.append(cur); cur = {}
for r in rows:
    if "name" in r: print("%-90s vgpr %4s agpr %3s spill %4s scratch %5s lds %6s" % (r["name"][:90], r.get("vgpr_count"), r.get("agpr_count"), r.get("vgpr_spill_count"), r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size")))
' | (c++filt 2>/dev/null || cat) | grep -E "$PAT"
[ -n "$KEEP_CO" ] && cp $T/dev.co $KEEP_CO
rm -rf $T
