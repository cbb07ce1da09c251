"""Instruction mix of the steady-state path of a kernel's inner loop, from `hipcc -S` output, priced with the per-instruction
issue times of the integer-pipe micro-benchmark (tools/ubench.hip, profiles/r01_ubench_v2_madadd.txt) -- an
instruction-count ceiling that does not come from timing the same loop.

usage: python tools/isa_count.py <file.s> <kernel-substring> <units-per-iteration> [block labels of the hot path ...]
Without block labels it lists the kernel's basic blocks (size, multiplier instructions) so that the steady-state
path can be picked by reading the listing."""
import json, re, sys
from collections import Counter

# ns per wave-instruction on one SIMD = measured "cyc/wave-instr/SIMD" at the nominal 2400 MHz / 2.4
UBENCH_NS = {"v_mad_u64_u32": 5.26 / 2.4, "v_mul_lo_u32": 4.79 / 2.4, "v_mul_hi_u32": 4.59 / 2.4, "v_add_u32": 3.42 / 2.4,
             "v_lshrrev_b64": 4.53 / 2.4, "v_lshl_add_u64": 4.93 / 2.4, "v_alignbit_b32": 4.42 / 2.4,
             "v_and_b32": 2.61 / 2.4, "v_mov_b32": 3.01 / 2.4}
DEFAULT_NS = 3.42 / 2.4   # other VALU instructions: priced like v_add_u32


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
           "pricing": "ns per wave-instruction per SIMD from tools/ubench (profiles/r01_ubench_v2_madadd.txt); 1024 SIMDs x 64 lanes"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
