#!/usr/bin/env python3
"""Register / scratch budget of every kernel in one translation unit (device-only compile, no GPU needed).

    tools/kernel_meta.py k_g1fft.hip [extra hipcc flags]

Prints name, VGPRs, AGPRs, SGPRs, spilled VGPRs, private (scratch) bytes per lane, LDS bytes.
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM_MAC = {"k_msm.hip", "k_g1fft.hip", "k_g1slp.hip", "k_g1circ.hip", "k_g1misc.hip", "k_verify.hip", "k_verify_many.hip", "k_table.hip",
           "k_msm_glv16.hip", "k_msm_glv15.hip", "k_msm_glv14.hip", "k_msm_glv12.hip", "k_msm_glv8.hip"}  # = ASM_MAC_TUS of csrc/Makefile
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    src = os.path.basename(sys.argv[1])
    out = os.environ.get("KM_OUT", "/tmp/km")
    os.makedirs(out, exist_ok=True)
    co = os.path.join(out, src + ".co")
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-unused-value"]
    if src in ASM_MAC:
        flags.append("-DFQ_ASM_MAC")
    flags += sys.argv[2:]
    subprocess.check_call(["hipcc", *flags, "--cuda-device-only", "-x", "hip", "-c",
                           os.path.join(REPO, "rust-eth-kzg_amd", "csrc", src), "-o", co])
    elf = co
    with open(co, "rb") as f:
        head = f.read(4)
    if head != b"\x7fELF":  # offload bundle: pull the gfx950 code object out
        elf = co + ".elf"
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={elf}"])
    txt = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", elf], text=True)
    for blk in txt.split("- .agpr_count:")[1:]:
        def g(k):
            m = re.search(r"\." + k + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        print(f"{g('name')[:64]:64s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>4s} sgpr {g('sgpr_count'):>4s} "
              f"spill {g('vgpr_spill_count'):>4s} scratch {g('private_segment_fixed_size'):>6s} lds {g('group_segment_fixed_size'):>6s}")
    print("code object:", elf)


if __name__ == "__main__":
    main()
