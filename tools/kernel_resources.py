#!/usr/bin/env python
"""Registers / scratch / LDS of the kernels in libpasst_amd.so whose name contains one of the given substrings
(from the code objects' metadata notes; the same extraction as tests/test_abi_cpu.py's no-scratch check).

    python tools/kernel_resources.py [--lib other.so] attn_ gemm_tn"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    argv = sys.argv[1:]
    so = os.path.join(ROOT, "passt_amd", "libpasst_amd.so")
    if argv and argv[0] == "--lib":
        so, argv = argv[1], argv[2:]
    pats = argv or [""]
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([LLVM + "llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so, os.path.join(d, "copy.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
        for k, st in enumerate(starts):
            part, co = os.path.join(d, f"b{k}.bin"), os.path.join(d, f"b{k}.co")
            open(part, "wb").write(blob[st:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            subprocess.run([LLVM + "clang-offload-bundler", "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={part}", f"--output={co}"], check=True)
            notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count")[1:]:
                blk = ".agpr_count" + blk
                n = re.search(r"\.name:\s+(\S+)", blk)
                if not n or not any(p in n.group(1) for p in pats):
                    continue

                def g(key):
                    m = re.search(key + r":\s+(\d+)", blk)
                    return m.group(1) if m else "?"
                print(f"{n.group(1)[:78]:78s} vgpr {g(r'.vgpr_count'):>3s} agpr {g(r'.agpr_count'):>3s} sgpr {g(r'.sgpr_count'):>3s} "
                      f"scratch {g(r'.private_segment_fixed_size'):>4s} lds {g(r'.group_segment_fixed_size')}")


if __name__ == "__main__":
    main()
