#!/usr/bin/env python3
"""Register / scratch metadata of every gfx950 kernel embedded in libyolov6_hip.so (llvm-readelf --notes of each code object):
name, vgprs, agprs, sgprs, spilled vgprs / sgprs, scratch bytes, LDS bytes.  `python tools/kernel_meta.py [substring]`"""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
blob = open(os.path.join(ROOT, "yolov6_amd", "lib", "libyolov6_hip.so"), "rb").read()
pat = sys.argv[1] if len(sys.argv) > 1 else ""
pos = blob.find(b"\x7fELF", 1)
rows = []
while pos >= 0:
    e_shoff, = struct.unpack_from("<Q", blob, pos + 0x28)
    es, en = struct.unpack_from("<HH", blob, pos + 0x3A)
    em, = struct.unpack_from("<H", blob, pos + 0x12)
    size = e_shoff + es * en
    if em == 224 and size > 0:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob[pos:pos + size]); f.flush()
            txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        cur = {}
        for l in txt.split("\n"):
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", l)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == "agpr_count" and cur.get("name"):
                rows.append(cur); cur = {}
            if k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "symbol"):
                cur[k] = v
        if cur.get("symbol") or cur.get("name"):
            rows.append(cur)
    pos = blob.find(b"\x7fELF", pos + 4)
seen = set()
for r in rows:
    n = r.get("symbol") or r.get("name") or "?"
    if pat not in n or n in seen or "vgpr_count" not in r:
        continue
    seen.add(n)
    dem = subprocess.run(["c++filt", n.replace(".kd", "")], capture_output=True, text=True).stdout.strip()
    print(f"{dem[:110]:110s} v{r.get('vgpr_count')} a{r.get('agpr_count')} s{r.get('sgpr_count')} vspill {r.get('vgpr_spill_count')} sspill {r.get('sgpr_spill_count')} scratch {r.get('private_segment_fixed_size')} lds {r.get('group_segment_fixed_size')}")
