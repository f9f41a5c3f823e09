#!/usr/bin/env python3
"""Build container only (needs /root/reference): parse the two machine-generated mirrors of the reference's C ABI that are
checked into its tree -- the Nim header (nbindgen) and the C# P/Invoke table (csbindgen), both produced from
bindings/c/src/lib.rs:79-566 -- into tests/golden/abi_signatures.json: symbol -> return type + ordered argument types.
tests/test_abi_exports.py parses include/c_eth_kzg.h the same way and requires equality for all 16 symbols, so a changed
argument order or width in the drop-in header fails on the CPU, not in a consumer's process.
The JSON is data (names and type shapes), not reference source text."""
import hashlib
import json
import os
import re

REF = "/root/reference/bindings"
NIM = os.path.join(REF, "nim/nim_code/nim_eth_kzg/header.nim")
CS = os.path.join(REF, "csharp/csharp_code/EthKZG.bindings/native_methods.g.cs")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "abi_signatures.json")

NIM_T = {"ptr DASContext": "DASContext*", "pointer": "ptr", "ptr pointer": "ptr*", "uint64": "uint64", "bool": "bool",
         "CResult": "CResult", "void": "void"}
CS_T = {"DASContext*": "DASContext*", "byte*": "uint8*", "byte**": "uint8**", "ulong": "uint64", "ulong*": "uint64*",
        "bool*": "bool*", "bool": "bool", "CResult": "CResult", "void": "void"}


def parse_nim(text):
    out = {}
    for m in re.finditer(r"proc (eth_kzg_\w+)\*\((.*?)\):\s*([\w ]+?)\s*\{\.importc", text, re.S):
        args = [a.split(":", 1)[1].strip() for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = {"ret": NIM_T[m.group(3).strip()], "args": [NIM_T[a] for a in args]}
    return out


def parse_cs(text):
    out = {}
    for m in re.finditer(r"internal static extern ([\w*]+) (eth_kzg_\w+)\((.*?)\);", text):
        args = []
        for a in [x.strip() for x in m.group(3).split(",") if x.strip()]:
            a = re.sub(r"\[.*?\]\s*", "", a)          # [MarshalAs(UnmanagedType.U1)] bool: one byte, as C's bool
            args.append(CS_T[a.rsplit(" ", 1)[0].strip()])
        out[m.group(2)] = {"ret": CS_T[m.group(1)], "args": args, "arg_names": [a.rsplit(" ", 1)[1].lstrip("@") for a in
                           [re.sub(r"\[.*?\]\s*", "", x.strip()) for x in m.group(3).split(",") if x.strip()]]}
    return out


def main():
    nim_text, cs_text = open(NIM).read(), open(CS).read()
    nim, cs = parse_nim(nim_text), parse_cs(cs_text)
    assert sorted(nim) == sorted(cs) and len(nim) == 16, (sorted(nim), sorted(cs))
    doc = {"note": "parsed by tools/gen_abi_signatures.py from the reference's generated mirrors of bindings/c/src/lib.rs; "
                   "nim: 'ptr' = an untyped pointer, 'ptr*' = pointer to pointers; csharp: typed (byte = uint8, ulong = uint64)",
           "sources": {"bindings/nim/nim_code/nim_eth_kzg/header.nim": hashlib.sha256(nim_text.encode()).hexdigest(),
                       "bindings/csharp/csharp_code/EthKZG.bindings/native_methods.g.cs": hashlib.sha256(cs_text.encode()).hexdigest()},
           "nim": nim, "csharp": cs}
    json.dump(doc, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, len(nim), "symbols")


if __name__ == "__main__":
    main()
