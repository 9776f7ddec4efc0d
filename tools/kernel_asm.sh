#!/bin/bash
# Developer tool: disassembly of one kernel of the built library.
#   tools/kernel_asm.sh <substring of the mangled kernel name> [library]
set -e
LIB=$(readlink -f "${2:-${ACG_LDPC_LIB:-$(dirname "$0")/../acg_alp_ldpc_amd/lib/libacg_ldpc_hip.so}}")
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
cd "$TMP"
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" fat.bin
python3 - "$1" <<'PY'
import subprocess, sys
data = open("fat.bin", "rb").read()
pos, k = 0, 0
while True:
    i = data.find(b"\x7fELF", pos)
    if i < 0:
        break
    j = data.find(b"\x7fELF", i + 4)
    open("co%d.elf" % k, "wb").write(data[i:j if j > 0 else len(data)])
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "co%d.elf" % k], capture_output=True, text=True).stdout
    on = False
    for line in out.split("\n"):
        if line[:1].isalnum() and line.rstrip().endswith(">:"):
            on = sys.argv[1] in line
        if on:
            print(line.split("//")[0].rstrip())
    k += 1
    pos = i + 4
PY
