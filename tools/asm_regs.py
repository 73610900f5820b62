"""Per-kernel resources from a hipcc -S device listing:  python tools/asm_regs.py file.s [filter]
(name, VGPRs (arch + acc), SGPR spills, LDS bytes, waves per SIMD allowed by the registers, code bytes if the listing carries them)"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]   # noqa: E731
    name = g("name")
    if flt and flt not in name:
        continue
    v = int(g("vgpr_count"))
    alloc = (v + 7) // 8 * 8
    print("%-70s vgpr %3d (acc %s) waves/simd %d  sgpr_spill %3s vgpr_spill %s lds %6s" % (
        name.replace("_ZN12_GLOBAL__N_1", ""), v, blk.split("\n")[0].strip(), min(8, 512 // alloc), g("sgpr_spill_count"), g("vgpr_spill_count"), g("group_segment_fixed_size")))
