#!/usr/bin/env python3
"""Static VALU instruction mix of a kernel of the built library, priced by operation class.

The per-operation issue costs are measured by tools/microbench/valu_op_rates.hip (profiles/r02_valu_op_rates.txt): two
classes, "full rate" (2 SIMD cycles per wave-instruction: fp32 add/mul/fma, and/or/xor, add/sub, right shifts,
v_mov_b32, v_cndmask on vcc) and "half rate" (4: everything else, incl. every fp64 / compare / min / max / three-operand
integer / SDWA / DPP / packed-f16 form), and 8 for the fp32 transcendentals.  Counters cannot split SQ_INSTS_VALU by
these classes, so bench.py prices all non-transcendental instructions at 2 (a lower bound of the utilisation); this tool
gives the average cost of the kernel's instruction stream from its disassembly (static counts: every instruction of the
kernel once — the sweep loops dominate the text of these kernels, and their bodies are straight-line code under EXEC
masks), which bench.py reports next to the lower bound.

    python tools/valu_mix.py "bp_fused_kernel<float, 8, 32, 0, false, true, 12, false>" [library]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
FULL_RATE = {"v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32",
             "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co_u32",
             "v_addc_co_u32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32"}
TRANS = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32"}
CYC_FULL, CYC_HALF, CYC_TRANS = 2.0, 4.0, 8.0


def default_lib():
    return os.environ.get("ACG_LDPC_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                           "acg_alp_ldpc_amd", "lib", "libacg_ldpc_hip.so")


def disassemble(lib):
    """-> {demangled kernel name: [instruction mnemonics]} for every gfx950 code object in the library's fat binary"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        pos, k = 0, 0
        while True:
            i = data.find(b"\x7fELF", pos)
            if i < 0:
                break
            j = data.find(b"\x7fELF", i + 4)
            co = os.path.join(tmp, "co%d.elf" % k)
            open(co, "wb").write(data[i:j if j > 0 else len(data)])
            txt = subprocess.run([LLVM + "/llvm-objdump", "-d", "--demangle", co], capture_output=True, text=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if m:
                    cur = out.setdefault(m.group(1), [])
                    continue
                if cur is not None:
                    t = line.split()
                    if t and re.match(r"^[vsd]s?_|^global_|^scratch_|^buffer_|^flat_", t[0]):
                        cur.append(t[0])
            k += 1
            pos = i + 4
    return out


def mix_of(insts):
    n_full = n_half = n_trans = 0
    for op in insts:
        if not op.startswith("v_"):
            continue
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        if base in TRANS:
            n_trans += 1
        elif base in FULL_RATE and not op.endswith(("_dpp", "_sdwa")):
            n_full += 1
        elif op == "v_cndmask_b32_e32":
            n_full += 1
        else:
            n_half += 1
    n = n_full + n_half + n_trans
    if not n:
        return None
    non_trans = n_full + n_half
    return {"valu_static": n, "full_rate": n_full, "half_rate": n_half, "transcendental": n_trans,
            "cycles_per_non_transcendental": (CYC_FULL * n_full + CYC_HALF * n_half) / max(non_trans, 1),
            "cycles_per_valu": (CYC_FULL * n_full + CYC_HALF * n_half + CYC_TRANS * n_trans) / n}


def static_mix(patterns, lib=None):
    """patterns: {key: substring of the demangled kernel name} -> {key: mix dict of the first matching kernel}"""
    kernels = disassemble(lib or default_lib())
    res = {}
    for key, pat in patterns.items():
        for name, insts in kernels.items():
            if pat in name:
                m = mix_of(insts)
                if m:
                    m["kernel"] = name[:160]
                    res[key] = m
                break
    return res


if __name__ == "__main__":
    r = static_mix({"k": sys.argv[1]}, sys.argv[2] if len(sys.argv) > 2 else None)
    print(r.get("k", "no kernel matches"))
