"""Instruction-class trace of one kernel from a `hipcc -S --cuda-device-only` listing: shows how VALU / LDS work is woven
around the MFMAs.   python tools/isa_mix.py listing.s <mangled-name-substring> [--runs]
M = bf16 MFMA, F = f32 MFMA, v = VALU, d = LDS, g = global/buffer, s = SALU, w = s_waitcnt, n = s_nop, b = branch/barrier"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
def cls(op):
    if op.startswith("v_mfma"):
        return "M" if "bf16" in op else "F"
    if op.startswith("v_accvgpr"): return "a"
    if op.startswith("v_"): return "v"
    if op.startswith("ds_"): return "d"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "g"
    if op.startswith("s_waitcnt"): return "w"
    if op.startswith("s_nop"): return "n"
    if op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm")): return "b"
    if op.startswith("s_"): return "s"
    return "?"
tr = []
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith((";", ".")): continue
    if re.match(r"^[.\w$]+:", t):
        tr.append("\n" + t.split(":")[0] + ": ")
        continue
    tr.append(cls(t.split()[0]))
s = "".join(tr)
from collections import Counter
print(Counter(c for c in s if c.isalpha() and c in "MFvadgswnb"))
if "--runs" in sys.argv:
    # lengths of VALU runs between consecutive MFMAs inside the block with the most MFMAs
    blk = max(s.split("\n"), key=lambda b: b.count("M"))
    body = blk.split(": ", 1)[-1]
    gaps = [len(re.sub(r"[^va]", "", g)) for g in re.split(r"[MF]", body)]
    print("block with %d bf16 MFMAs, %d f32 MFMAs, %d instructions; VALU between consecutive MFMAs: histogram" % (body.count("M"), body.count("F"), len(body)))
    print(sorted(Counter(gaps).items()))
    i0 = body.find("M")
    print(body[max(0, i0 - 200):i0 + 3500])
else:
    print(s[:4000])
