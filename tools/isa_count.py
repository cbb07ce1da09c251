"""Instruction mix of the steady-state path of a kernel's inner loop, from `hipcc -S` output, priced with the per-instruction
issue costs of the round-4 pricing micro-benchmark (tools/ubench4.hip, profiles/r04_ubench4_pricing.txt) -- an
instruction-count ceiling that does not come from timing the same loop.

usage: python tools/isa_count.py <file.s> <kernel-substring> <units-per-iteration> [block labels of the hot path ...]
Without block labels it lists the kernel's basic blocks (size, multiplier instructions) so that the steady-state
path can be picked by reading the listing."""
import json, re, sys
from collections import Counter

# Price list (round 4): cycles per wave-instruction per SIMD at saturation, measured with s_memtime per physical SIMD and an
# occupancy sweep (tools/ubench4.hip -> profiles/r04_ubench4_pricing.txt, the W = 8 column).  Three classes fall out:
#   full rate  (VOP2 add / sub / and / xor / shifts by register, mov)            2.20 cycles  (the guide's 2-cycle wave64 issue)
#   half rate  (v_mul_lo/hi_u32, 64-bit shifts and adds, v_add3, v_alignbit, v_mad_u32_u24, v_or3, v_fma_f64)   4.12 cycles
#   v_mad_u64_u32 and the carry ops that follow it (v_addc_co_u32 ...)            4.27 cycles
# A ceiling in units/s is these cycles at the chip's MAXIMUM clock, 2400 MHz (/opt/skills/guides/MI355X_MICROARCH.md,
# chip-level parameters) -- not at the clock some run happened to hold: the multiplier stream is power-limited to
# 2.06-2.2 GHz (ubench4's g1_madd row: 2061 MHz; SQ counters of the real kernel: 2.17-2.2 GHz), and that shortfall is
# part of the distance to the ceiling, reported next to it.
CLOCK_HZ = 2.4e9
FULL, HALF, MAD = 2.20, 4.12, 4.27
CYCLES = {"v_mad_u64_u32": MAD, "v_addc_co_u32": MAD, "v_subb_co_u32": MAD, "v_subbrev_co_u32": MAD, "v_add_co_u32": MAD,
          "v_sub_co_u32": MAD, "v_subrev_co_u32": MAD,
          "v_mul_lo_u32": HALF, "v_mul_hi_u32": HALF, "v_lshrrev_b64": HALF, "v_lshlrev_b64": HALF, "v_lshl_add_u64": HALF,
          "v_add3_u32": HALF, "v_alignbit_b32": HALF, "v_mad_u32_u24": HALF, "v_or3_b32": HALF, "v_fma_f64": HALF,
          "v_lshlrev_b32": HALF, "v_lshl_add_u32": HALF, "v_lshl_or_b32": HALF, "v_and_or_b32": HALF, "v_bfe_u32": HALF,
          "v_add_lshl_u32": HALF, "v_xad_u32": HALF, "v_mad_u64_u32_e64": MAD}
DEFAULT_CYCLES = FULL   # the remaining VOP1 / VOP2 forms (v_add_u32, v_sub_u32, v_and_b32, v_mov_b32, v_cndmask_b32, shifts ...)
UBENCH_NS = {k: v / CLOCK_HZ * 1e9 for k, v in CYCLES.items()}
DEFAULT_NS = DEFAULT_CYCLES / CLOCK_HZ * 1e9


def main():
    path, kern, units = sys.argv[1], sys.argv[2], float(sys.argv[3])
    hot = set(sys.argv[4:])
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(kern) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], ["entry", Counter()]
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), Counter()]
            continue
        t = l.strip().split()
        if t and not t[0].startswith((";", ".")):
            cur[1][t[0]] += 1
    blocks.append(cur)
    if not hot:
        for name, c in blocks:
            print(f"{name:12s} {sum(c.values()):6d} instr  v_mad_u64_u32 {c['v_mad_u64_u32']:5d}")
        return
    mix = Counter()
    for name, c in blocks:
        if name in hot:
            mix.update(c)
    valu = Counter()
    for k, v in mix.items():
        if k.startswith("v_"):
            valu[re.sub(r"_e(32|64)$", "", k)] += v
    t_all = sum(v * UBENCH_NS.get(k, DEFAULT_NS) for k, v in valu.items())
    t_mad = mix["v_mad_u64_u32"] * UBENCH_NS["v_mad_u64_u32"]
    simds, lanes = 1024, 64
    out = {"kernel": kern, "hot_blocks": sorted(hot), "units_per_iteration": units,
           "instructions": sum(mix.values()), "valu_instructions": sum(valu.values()),
           "v_mad_u64_u32": mix["v_mad_u64_u32"],
           "top": dict(valu.most_common(12)),
           "ns_per_wave_iteration_multiplier_only": t_mad, "ns_per_wave_iteration_all_valu": t_all,
           "ceiling_units_per_s_multiplier_only": simds * lanes * units / (t_mad * 1e-9),
           "ceiling_units_per_s_all_valu": simds * lanes * units / (t_all * 1e-9),
           "cycles_per_wave_iteration_multiplier_only": mix["v_mad_u64_u32"] * MAD,
           "cycles_per_wave_iteration_all_valu": sum(v * CYCLES.get(k, DEFAULT_CYCLES) for k, v in valu.items()),
           "clock_hz": CLOCK_HZ, "cycles_full_half_mad": [FULL, HALF, MAD],
           "pricing": "cycles per wave-instruction per SIMD at saturation, s_memtime per physical SIMD, occupancy sweep "
                      "(tools/ubench4.hip, profiles/r04_ubench4_pricing.txt), at the 2400 MHz maximum clock; 1024 SIMDs x 64 lanes"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
