#!/usr/bin/env python3
"""Per-kernel scratch (private segment) and register use of the BUILT library: unbundles every gfx950 code object embedded in
libfq3hip.so (clang offload bundles in .hip_fatbin), reads the AMDGPU metadata notes with llvm-readelf and lists every kernel whose
private segment is not empty (spills / stack arrays).  usage: check_scratch.py [path/to/libfq3hip.so] [--all] [--top N]
--top N: the N kernels with the most registers (arch + accumulation VGPRs; above 256 a wave64 kernel runs one wave per SIMD).
Exit code 1 if any kernel uses scratch.  tests/test_abi.py::test_no_kernel_uses_scratch runs it."""
import os, re, struct, subprocess, sys, tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size > 0:
                yield data[i + off:i + off + size]
        pos = i + len(MAGIC)


def kernels(path):
    out = []
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n  - (?=\.agpr_count:)", txt)[1:]:
            blk = blk.split("\namdhsa.target")[0]
            g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "0"])[1]
            out.append(dict(name=g("name").strip("'"), scratch=int(g("private_segment_fixed_size")), vgpr=int(g("vgpr_count")),
                            agpr=int(g("agpr_count")), lds=int(g("group_segment_fixed_size"))))
    return out


def main():
    args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--top"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = args[0] if args else os.path.join(root, "faster-qwen3-tts_amd", "lib", "libfq3hip.so")
    ks = kernels(path)
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in ks), capture_output=True, text=True).stdout.split("\n")
    bad = 0
    for k, n in zip(ks, names):
        n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("fq3::", "").replace("unsigned short", "bf16").replace("void ", ""))
        if k["scratch"] or "--all" in sys.argv:
            print(f"{n[:90]:90s} scratch={k['scratch']:5d} vgpr={k['vgpr']:3d} agpr={k['agpr']:3d} lds={k['lds']}")
        bad += k["scratch"] > 0
    if "--top" in sys.argv:
        n_top = int(sys.argv[sys.argv.index("--top") + 1])
        ranked = sorted(zip(ks, names), key=lambda kn: -(kn[0]["vgpr"] + kn[0]["agpr"]))[:n_top]
        print(f"top {n_top} register users (arch + acc VGPRs):")
        for k, n in ranked:
            n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("fq3::", "").replace("unsigned short", "bf16").replace("void ", ""))
            print(f"  {k['vgpr'] + k['agpr']:4d} = {k['vgpr']:3d} + {k['agpr']:3d}  lds={k['lds']:6d}  {n[:100]}")
    print(f"{len(ks)} kernels, {bad} with a non-empty private segment")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
