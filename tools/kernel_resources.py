#!/usr/bin/env python3
"""Registers, scratch, LDS, occupancy and static instruction count of every kernel of a HIP source, from the compiler itself.

usage: python tools/kernel_resources.py [pam_amd/csrc/awfl_kernels.hip] > profiles/rNN_kernel_resources.txt
Compiles the file for gfx950 to assembly with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one row per kernel:
VGPRs / AGPRs / SGPRs (the compiler's request; the hardware allocates VGPRs in granules of 8), scratch bytes per lane, waves per
SIMD the registers allow, static LDS bytes per workgroup (the tile kernels request theirs dynamically), and the number of machine
instructions of the kernel's body -- code the tile kernels execute ONCE per wavefront, straight through."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "pam_amd", "csrc", "awfl_kernels.hip")
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                            "--cuda-device-only", "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage", src, "-o", asm],
                           capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
        s = open(asm).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    sizes = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n", s, flags=re.M):
        body = s[m.end():]
        e = body.find(".Lfunc_end")
        if e >= 0:
            sizes[m.group(1)] = sum(1 for l in body[:e].split("\n")
                                    if re.match(r"\s+(s_|v_|global_|ds_|buffer_|flat_|scratch_)", l))
    names = [b.split()[0] for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    print("# %s, hipcc --offload-arch=gfx950 -O3 (-Rpass-analysis=kernel-resource-usage); instr = machine instructions of the kernel (8 bytes each for most VALU / memory forms)"
          % os.path.relpath(src, ROOT))
    print("%-64s %5s %5s %5s %8s %5s %8s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "w/SIMD", "LDS(st.)", "instr"))
    for b, n, d in zip(blocks, names, dem):
        def g(k):
            mm = re.search(k + r": (\d+)", b)
            return int(mm.group(1)) if mm else -1
        d = re.sub(r"^void ", "", d).replace("(anonymous namespace)::", "")
        d = re.sub(r"\(.*", "", d)
        print("%-64s %5d %5d %5d %8d %5d %8d %7d" % (d[:64], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                                                    g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), sizes.get(n, -1)))


if __name__ == "__main__":
    main()
