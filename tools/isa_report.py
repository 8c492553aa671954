#!/usr/bin/env python
"""Static resource report of every gfx950 kernel of the library: registers (arch + accumulator), scalar registers,
scratch (spill) bytes and LDS bytes, from the assembly hipcc emits (no GPU needed).
Usage: tools/isa_report.py [out.txt]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "monkey-net_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
         "-I" + SRC, "-S", "--cuda-device-only"]


def main():
    if len(sys.argv) > 1 and sys.argv[1].startswith("-"):     # (an option is not an output path: `--help` once became a file)
        print(__doc__)
        return
    lines = ["# kernel: vgpr+agpr (accumulator offset) sgpr scratch-bytes lds-bytes   [hipcc %s]" % " ".join(FLAGS[:3])]
    spills = 0
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
            out = os.path.join(tmp, os.path.basename(f) + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, f], check=True, capture_output=True)
            txt = open(out).read()
            rows = []
            for blk in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.S):
                def g(k, b=blk.group(2)):
                    m = re.search(r"\.amdhsa_%s (\d+)" % k, b)
                    return int(m.group(1)) if m else -1
                rows.append((blk.group(1), g("next_free_vgpr"), g("accum_offset"), g("next_free_sgpr"),
                             g("private_segment_fixed_size"), g("group_segment_fixed_size")))
            names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True,
                                   stdin=subprocess.DEVNULL).stdout.splitlines() if rows else []
            lines.append("== " + os.path.basename(f))
            for r, dn in zip(rows, names):
                dn = re.sub(r"\(anonymous namespace\)::", "", dn)
                dn = re.sub(r"\(.*", "", re.sub(r"^void ", "", dn))
                spills += r[4] > 0
                lines.append("%-58s regs %4d (acc %4d) sgpr %3d scratch %5d lds %6d%s"
                             % (dn[:58], r[1], r[2], r[3], r[4], r[5], "   <-- spills" if r[4] > 0 else ""))
    lines.append("# kernels with scratch (register spills): %d" % spills)
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
