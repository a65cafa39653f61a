"""Static check of the gfx950 code hipcc emits for kernels whose MFMAs are INLINE ASM (csrc/d3k_conv.hpp): the compiler pads no hazard
for an asm statement, so the distance between such an MFMA and the first non-MFMA instruction that touches its destination registers
is measured here, on the device assembly (`hipcc --cuda-device-only -S`), in wait states: 1 per instruction, n + 1 for `s_nop n`
(the pessimistic count: an intervening MFMA really occupies the issue port for several cycles).
Used by tests/test_isa_hazards.py (CPU, no GPU needed: hipcc cross-compiles)."""
import re
import subprocess

HIPCC = "/opt/rocm/bin/hipcc"
_REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def device_asm(source, out, extra=()):
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", source, "-o", out] + list(extra)
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    with open(out) as f:
        return f.read()


def regs_of(text):
    """{('v', 12), ('a', 3), ...} named in an operand string."""
    out = set()
    for kind, single, lo, hi in _REG.findall(text):
        if single:
            out.add((kind, int(single)))
        else:
            out.update((kind, i) for i in range(int(lo), int(hi) + 1))
    return out


def kernels(asm):
    """{kernel symbol: [(mnemonic, operand text, inside_inline_asm)]} for every .amdhsa kernel of a device assembly file."""
    out, cur, in_asm = {}, None, False
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            in_asm = False
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".section") and ".rodata" in s:
            cur = None
            continue
        if "#ASMSTART" in s:
            in_asm = True
            continue
        if "#ASMEND" in s:
            in_asm = False
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        s = s.split(";")[0].strip()
        parts = s.split(None, 1)
        cur.append((parts[0], parts[1] if len(parts) > 1 else "", in_asm))
    return {k: v for k, v in out.items() if v}


def is_mfma(op):
    return op.startswith("v_mfma") or op.startswith("v_smfmac")


def asm_mfma_distances(insts):
    """For every inline-asm MFMA: wait states up to the first non-MFMA instruction that names one of its destination registers.
    Returns [(index of the MFMA, index of that instruction, wait states, mnemonic of that instruction)]."""
    res = []
    pending = {}                                  # register -> (mfma index, wait states elapsed since it issued)
    for i, (op, args, in_asm) in enumerate(insts):
        ws = 1
        if op == "s_nop":
            ws = int(args.strip(), 0) + 1
        if not is_mfma(op):
            hit = {}
            for r in regs_of(args):
                if r in pending:
                    hit[pending[r][0]] = pending[r][1]
            for mi, w in hit.items():
                res.append((mi, i, w, op))
            for r in [r for r, (mi, _) in pending.items() if mi in hit]:
                del pending[r]
        else:
            # an MFMA that takes the accumulator whole as C continues the chain: no requirement; it re-arms the registers below
            pass
        for r in list(pending):
            pending[r] = (pending[r][0], pending[r][1] + ws)
        if is_mfma(op) and in_asm:
            dst = args.split(",")[0]
            for r in regs_of(dst):
                pending[r] = (i, 0)
    return res
