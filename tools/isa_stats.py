#!/usr/bin/env python3
"""Per-kernel static statistics from a `hipcc -S --cuda-device-only` listing: code bytes, VGPRs, LDS, VALU / DPP /
ds_bpermute / s_waitcnt counts.  usage: isa_stats.py file.s [name-filter ...]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
filters = sys.argv[2:]
blocks = re.findall(r'^(_Z\w+):[^\n]*\n(.*?); codeLenInByte = (\d+).*?; NumVgprs: (\d+).*?; LDSByteSize: (\d+)', txt, re.S | re.M)
names = subprocess.run(['c++filt'], input="\n".join(b[0] for b in blocks), capture_output=True, text=True).stdout.split("\n")
for b, n in zip(blocks, names):
    n = re.sub(r"\(.*", "", n.replace("fq3::", "").replace("unsigned short", "bf16").replace("void ", ""))
    if filters and not any(f in n for f in filters):
        continue
    body = b[1]
    cnt = lambda pat: len(re.findall(pat, body, re.M))
    valu = cnt(r'^\s+v_')
    print(f"{n[:60]:60s} code={b[2]:>6s} vgpr={b[3]:>4s} lds={b[4]:>6s} valu={valu:5d} "
          f"bperm={cnt('ds_bpermute'):3d} dpp={cnt('_dpp'):3d} swap={cnt('permlane'):3d} waits={cnt('s_waitcnt'):3d}")
