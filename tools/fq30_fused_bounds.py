"""Exact column bounds of the fused Fq30 multiplication / squaring (typlonk_amd/csrc/fq30.hpp,
fq30_mulsqr_fused): which mads of which columns can carry out of the 64-bit accumulator.

Program order inside a column: reduction terms m_i * p_(k-i), then the product terms, then (first half)
m_k * p_0.  Inputs are normalised 30-bit digits, m_i < 2^30.  Prints the capture schedule the header
hard-codes (columns 11..14: product terms from index 10 / 5 on; m_k * p_0 of columns 10..12) and fails
if the header's rule would miss a term."""
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
M = (1 << 30) - 1
PL = [(P >> (30 * i)) & M for i in range(13)]
LIM = (1 << 64) - 1


def schedule(kind):
    carry, out = 0, []
    for k in range(26):
        red = M * (sum(PL[1:k + 1]) if k < 13 else sum(PL[k - 12:13]))
        last = M * PL[0] if k < 13 else 0
        terms = []
        if k <= 24:
            lo, hi = max(0, k - 12), min(k, 12)
            if kind == "mul":
                terms = [M * M] * (hi - lo + 1)
            else:
                if k % 2 == 0:
                    terms.append(M * M)
                terms += [(2 * M) * M for i in range(lo, hi + 1) if 2 * i < k]
        run, first = carry + red, None
        assert run <= LIM
        for idx, t in enumerate(terms):
            if run + t > LIM and first is None:
                first = idx
            run += t
        cap_last = last > 0 and run + last > LIM
        run += last
        assert run < (1 << 66)  # third word holds it (and the shifted carry stays < 2^36)
        out.append((k, len(terms), first, cap_last))
        carry = run >> 30
    return out


def header_rule(kind, k, idx):
    first = 10 if kind == "mul" else 5
    return 11 <= k <= 14 and idx >= first


if __name__ == "__main__":
    for kind in ("mul", "sqr"):
        n = 0
        for k, nterms, first, cap_last in schedule(kind):
            need = set(range(first, nterms)) if first is not None else set()
            have = {i for i in range(nterms) if header_rule(kind, k, i)}
            assert need <= have, (kind, k, need, have)
            assert (not cap_last) or 10 <= k <= 12, (kind, k)
            n += len(have) + (1 if 10 <= k <= 12 else 0)
            if need or cap_last:
                print(f"{kind}: column {k:2d}: {nterms:2d} product terms, capture from index {first}, m_k*p_0 captured: {cap_last}")
        print(f"{kind}: {n} carry-capturing mads per operation")
