#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel in csrc/libgpcc_hip.so (the gfx950 code objects inside the fat binary: roc-obj-ls /
roc-obj-extract + llvm-readelf --notes).  Prints the kernels that spill or use scratch, and the totals.
  python tools/kernel_resources.py [library] [--all]"""
import os
import re
import subprocess
import sys
import tempfile

lib = next((a for a in sys.argv[1:] if not a.startswith("--")), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpcc.jl_amd", "csrc", "libgpcc_hip.so"))
show_all = "--all" in sys.argv
readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
import struct
blob = open(lib, "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
objs = []
pos = blob.find(MAGIC)
while pos >= 0:     # one bundle per translation unit: header = magic, u64 entries, then (offset, size, triple length, triple) each
    n = struct.unpack_from("<Q", blob, pos + 24)[0]
    q = pos + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", blob, q)
        triple = blob[q + 24:q + 24 + tl].decode()
        q += 24 + tl
        if "gfx950" in triple and size:
            objs.append(blob[pos + off:pos + off + size])
    pos = blob.find(MAGIC, pos + 24)
uris = objs
rows = []
with tempfile.TemporaryDirectory() as td:
    for i, data in enumerate(objs):
        co = os.path.join(td, "co%d.o" % i)
        with open(co, "wb") as f:
            f.write(data)
        notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True).stdout
        cur = {}
        for ln in notes.splitlines():
            ln = ln.strip().lstrip("- ").strip()
            mm = re.match(r"\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)", ln)
            if not mm:
                continue
            k, v = mm.group(1), mm.group(2)
            if k == "name":
                if v.startswith("_Z") or v.startswith("gpcc"):
                    if "vgpr_count" in cur and "name" in cur:
                        rows.append(cur)
                        cur = {}
                    if "name" not in cur or "vgpr_count" in cur:
                        cur = dict((kk, vv) for kk, vv in cur.items() if kk != "name") if "vgpr_count" not in cur else {}
                    cur["name"] = v
            else:
                if k == "group_segment_fixed_size" and "name" in cur and "vgpr_count" in cur:
                    rows.append(cur)   # (keys come in alphabetical order: this one is the FIRST of the next kernel's record, its name follows)
                    cur = {}
                cur[k] = int(v)
        if "vgpr_count" in cur and "name" in cur:
            rows.append(cur)
demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, demangle):
    r["pretty"] = d.split("(")[0]
bad = [r for r in rows if r.get("vgpr_spill_count", 0) or r.get("private_segment_fixed_size", 0)]
print("%d kernels in %d gfx950 code objects of %s (%.1f MB)" % (len(rows), len(uris), os.path.basename(lib), os.path.getsize(lib) / 1e6))
for r in (rows if show_all else bad):
    print("  %-70s vgpr %3d  spilled %3d  scratch %4d B  lds %6d B" % (r["pretty"][:70], r.get("vgpr_count", 0), r.get("vgpr_spill_count", 0),
                                                                       r.get("private_segment_fixed_size", 0), r.get("group_segment_fixed_size", 0)))
print("kernels with spills or scratch: %d" % len(bad))
