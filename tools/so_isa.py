#!/usr/bin/env python3
"""Disassemble the gfx950 code objects embedded in a built shared library (the .hip_fatbin section: one clang offload bundle per
translation unit) without recompiling anything.

    python tools/so_isa.py passt_amd/libpasst_amd.so            # per code object: first kernel, instruction count, cache-policy bits
    code_objects(path) -> [(first_symbol, disassembly_text)]     # for tests (tests/test_abi_cpu.py)
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so_path, arch="gfx950"):
    with tempfile.TemporaryDirectory() as d:
        fb = os.path.join(d, "fatbin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, so_path, os.path.join(d, "copy.so")],
                       check=True, capture_output=True)
        blob = open(fb, "rb").read()
        out = []
        for m in re.finditer(re.escape(MAGIC), blob):
            p = m.start()
            q = p + len(MAGIC)
            n, = struct.unpack_from("<Q", blob, q)
            q += 8
            for _ in range(n):
                off, size, tl = struct.unpack_from("<QQQ", blob, q)
                q += 24
                triple = blob[q:q + tl].decode()
                q += tl
                if arch in triple and size:
                    co = os.path.join(d, f"co{len(out)}.co")
                    open(co, "wb").write(blob[p + off:p + off + size])
                    text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
                    syms = re.findall(r"^[0-9a-f]+ <(\S+)>:", text, flags=re.M)
                    out.append((syms[0] if syms else "", text))
        return out


def policy_counts(text):
    """instructions carrying the non-temporal bit, by class"""
    return {"buffer_store nt": len(re.findall(r"buffer_store_dwordx4 .* nt", text)),
            "buffer_load nt": len(re.findall(r"buffer_load_dwordx4 .* nt", text)),
            "lds_dma nt": len(re.findall(r"global_load_lds_dwordx4 .* nt", text)),
            "global nt": len(re.findall(r"global_(?:load|store)_dword\S* .* nt", text))}


def kernel_bodies(text):
    """{kernel symbol: its disassembly}"""
    parts = re.split(r"^[0-9a-f]+ <(\S+)>:\n", text, flags=re.M)
    return {parts[i]: parts[i + 1] for i in range(1, len(parts) - 1, 2)}


if __name__ == "__main__":
    for first, text in code_objects(sys.argv[1]):
        print(f"{first[:70]:70s} {len(text.splitlines()):8d} lines  {policy_counts(text)}")
