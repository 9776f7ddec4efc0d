#!/bin/bash
# Developer tool: VGPR / SGPR / scratch / LDS of the kernels in the built library (from the code-object metadata).
#   tools/kernel_regs.sh [name-filter]
set -e
LIB=$(readlink -f "${ACG_LDPC_LIB:-$(dirname "$0")/../acg_alp_ldpc_amd/lib/libacg_ldpc_hip.so}")
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
cd "$TMP"
# the fat binary sits in .hip_fatbin; every gfx950 code object in it is an ELF file
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" fat.bin
python3 - "$1" <<'PY'
import re, subprocess, sys
data = open("fat.bin", "rb").read()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
pos, k, rows = 0, 0, []
while True:
    i = data.find(b"\x7fELF", pos)
    if i < 0:
        break
    j = data.find(b"\x7fELF", i + 4)
    open("co%d.elf" % k, "wb").write(data[i:j if j > 0 else len(data)])
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", "co%d.elf" % k], capture_output=True, text=True).stdout
    for blk in out.split("- .agpr_count:")[1:]:
        def g(key):
            mm = re.search(r"\." + key + r":\s+(\S+)", blk)
            return mm.group(1) if mm else "?"
        name = g("name")
        if flt and flt not in name:
            continue
        rows.append((name, g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), blk.split("\n")[0].strip()))
    pos = i + 4
    k += 1
for r in rows:
    dem = subprocess.run(["c++filt", r[0]], capture_output=True, text=True).stdout.strip()
    print("vgpr %3s agpr %3s sgpr %3s scratch %5s lds %6s  %s" % (r[1], r[5], r[2], r[3], r[4], dem[:160]))
PY
