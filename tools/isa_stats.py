#!/usr/bin/env python
"""Static instruction statistics of one kernel in a device assembly file (hipcc --cuda-device-only -S).

    tools/isa_stats.py fft2.s 'k_fft_pair2.*5120.*4704.*384.*IfEfLi1|k_fft_pair2.*PairSpecILi5120ELi4704' [--dump out.s]

Counts by class (VALU / SALU / LDS / VMEM / SMEM / other), the VALU opcodes by frequency, vector instructions with an
SGPR source operand (half issue rate on gfx950: tools/ubench/valu_ops.hip), and the resource lines of the kernel.
Static counts: loops are counted once (the paired FFT kernels are fully unrolled, so static = dynamic per thread there)."""
import collections
import re
import sys


def main():
    path, pat = sys.argv[1], re.compile(sys.argv[2])
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    body, name, inside = [], None, False
    meta = []
    for line in open(path):
        if not inside:
            m = re.match(r"^(_Z\w+):", line)
            if m and pat.search(m.group(1)):
                name, inside = m.group(1), True
            continue
        if line.startswith(".Lfunc_end"):
            break
        body.append(line.rstrip("\n"))
    if not name:
        sys.exit("kernel not found")
    txt = open(path).read()
    i = txt.find(".amdhsa_kernel " + name)
    if i >= 0:
        for key in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "accum_offset"):
            m = re.search(r"\.amdhsa_%s (\S+)" % key, txt[i:i + 4000])
            if m:
                meta.append(f"{key}={m.group(1)}")
    cls = collections.Counter()
    ops = collections.Counter()
    sgpr_src = collections.Counter()
    lits = 0
    for l in body:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        op = l.split()[0]
        if op.startswith("v_"):
            c = "MFMA" if "mfma" in op else "VALU"
        elif op.startswith("s_"):
            c = "SMEM" if op.startswith(("s_load", "s_buffer_load")) else "SALU"
        elif op.startswith("ds_"):
            c = "LDS"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c = "VMEM"
        else:
            c = "other"
        cls[c] += 1
        ops[op] += 1
        if c == "VALU":
            args = l[len(op):].split(",")
            srcs = [a.strip() for a in args[1:]]
            if any(re.match(r"^-?\|?s\d+|^-?s\[", a) or a in ("vcc_lo", "vcc_hi") for a in srcs if not a.startswith(("v", "-v", "|v"))):
                sgpr_src[op] += 1
            if any(re.match(r"^0x[0-9a-f]+$", a) for a in srcs):
                lits += 1
    print(name)
    print("  ", " ".join(meta))
    print("   classes:", dict(cls))
    print("   VALU with an SGPR source:", sum(sgpr_src.values()), dict(sgpr_src.most_common(8)))
    print("   VALU with a 32-bit literal:", lits)
    print("   top opcodes:", ", ".join(f"{o} {n}" for o, n in ops.most_common(28)))
    if dump:
        open(dump, "w").write("\n".join(body))


if __name__ == "__main__":
    main()
