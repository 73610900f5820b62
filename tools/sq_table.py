"""Per kernel-instance issue accounting from the SQ counters of tools/pmc_sq.sh (sq_counters.json) -- the evidence behind "what bounds the
convolution family" in DESIGN.md.  python tools/sq_table.py <sq_counters.json> [<mfma_util.json>] > table.md

Units (MI355X_MICROARCH.md, PMC section): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (32 per v_mfma_f32_16x16x4_f32, 64 per v_mfma_f32_32x32x2_f32); SQ_INSTS_* count
wave instructions.  All values are averages per launch over one eager `bench.py --steps 2 --warmup 1 --no_graph` run.

Columns
  waves         wavefronts per launch
  life          mean wave lifetime in cycles (4 * SQ_WAVE_CYCLES / SQ_WAVES)
  mfma/wave     MFMA instructions per wave (busy cycles / 32, / 64 for the 32x32x2 kernels, / waves)
  cyc/mfma      wave lifetime per MFMA it issues; the pipe's floor is 32 (64) with ONE wave per SIMD and 32 (64) * k with k resident waves
                sharing the SIMD, so cyc/mfma / (32 * k) is the inverse of the pipe utilisation while the wave is alive
  wait          SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share of the wave lifetime spent waiting on an outstanding instruction (s_waitcnt)
  lds wait      SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  bank          SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: share of LDS pipe cycles lost to bank conflicts
  valu/mfma     non-MFMA VALU instructions per MFMA (address arithmetic, normalise-on-load, epilogue)
  salu/mfma     SALU instructions per MFMA
  lds/mfma      LDS instructions per MFMA (fragment reads + staging writes)
  vmem/mfma     global load instructions per MFMA
  util          MFMA pipe utilisation over the launch (mfma_util.json: busy / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)), when given
"""
import json
import sys


def main():
    sq = json.load(open(sys.argv[1]))
    util = {}
    if len(sys.argv) > 2:
        util = json.load(open(sys.argv[2])).get("kernels", {})
    rows = []
    for name, c in sq.items():
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if busy <= 0 or c.get("SQ_WAVES", 0) <= 0:
            continue
        per = 64.0 if ("wide" in name or "rowrun" in name or "flat" in name) else 32.0
        nm = busy / per
        waves = c["SQ_WAVES"]
        life = 4.0 * c["SQ_WAVE_CYCLES"] / waves
        wc = c["SQ_WAVE_CYCLES"]
        u = util.get(name, {})
        rows.append((busy, name, waves, life, nm / waves, life / (nm / waves), c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_WAIT_INST_LDS", 0) / wc,
                     c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0)), (c.get("SQ_INSTS_VALU", 0) - nm) / nm,
                     c.get("SQ_INSTS_SALU", 0) / nm, c.get("SQ_INSTS_LDS", 0) / nm, c.get("SQ_INSTS_VMEM_RD", 0) / nm, u.get("mfma_util"), u.get("launches")))
    rows.sort(reverse=True)
    print("| kernel instance | waves | life (cyc) | mfma/wave | cyc/mfma | wait | lds wait | bank | valu/mfma | salu/mfma | lds/mfma | vmem/mfma | util |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| `%s` | %.0f | %.0f | %.0f | %.0f | %.2f | %.3f | %.3f | %.1f | %.1f | %.2f | %.2f | %s |" % (
            r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11], r[12], "%.2f" % r[13] if r[13] is not None else "-"))


if __name__ == "__main__":
    main()
