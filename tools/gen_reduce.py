#!/usr/bin/env python3
"""Generator + checker of the joint DPP row reduction of csrc/blend_bwd.hip (run on the CPU; prints the inline-asm body).

The backward blend has, per surviving PAIR of list entries and per lane (= pixel), NS values of entry 0 in registers A[0..NS)
and NS values of entry 1 in B[0..NS) (NS = 9: colour-only, 12: all upstream gradients).  They have to be summed over the 64
pixels of the wave.  This network does the part inside every 16-lane row with DPP adds, for BOTH entries at once:

  step 1 (partner lane ^ 8): lanes 0-7 of a row keep entry 0, lanes 8-15 entry 1     -> NS registers, 2 NS instructions
  step 2 (lane ^ 4), 3 (lane ^ 2), 4 (lane ^ 1): the usual halving butterfly on the NS values of the half-row

and leaves TWO registers per lane: lane h = (lane & 7) of a half-row holds slot `slot_r0[h]` in A[0] and `slot_r1[h]` in A[1]
(None = the register holds a duplicate / partial that must not be used).  The sums over the four rows of the wave are taken by
the LDS (ds_add_f32 of all four rows into one accumulator word per slot), see blend_bwd.hip.

The script simulates every instruction on symbolic values (one row of 16 lanes: exact bookkeeping of which (lane, value) terms
a register holds), checks that the advertised registers hold every term exactly once, checks the DPP read-after-VALU-write
distance (2 wait states), and prints the asm text.  `python tools/gen_reduce.py 9` / `12`.
"""
import sys
from collections import Counter


def xor_src(kind):
    """source lane (within the row of 16) that lane c reads under the DPP control `kind`"""
    return {
        "row_ror:8": lambda c: (c + 8) % 16,
        "row_shl:4": lambda c: c + 4,          # used with bank_mask 0x5 only (lanes 0-3, 8-11)
        "row_shr:4": lambda c: c - 4,          # used with bank_mask 0xa only (lanes 4-7, 12-15)
        "quad_perm:[2,3,0,1]": lambda c: c ^ 2,
        "quad_perm:[1,0,3,2]": lambda c: c ^ 1,
    }[kind]


class Net:
    def __init__(self, ns):
        self.ns = ns
        self.ins = []          # (op, dst, src, ctrl, bank_mask) | ("cnd", dst, a, b, maskname) | ("nop", n)
        # symbolic state: reg name -> list of 16 Counters of (src_lane, value_name)
        self.reg = {}
        for j in range(ns):
            self.reg["A%d" % j] = [Counter({(c, "a%d" % j): 1}) for c in range(16)]
            self.reg["B%d" % j] = [Counter({(c, "b%d" % j): 1}) for c in range(16)]
        self.last_write = {}   # reg -> instruction index of its last VALU write

    def _hazard(self, src):
        """a DPP instruction reads `src` through the lane crossbar: 2 wait states after the VALU write of src"""
        lw = self.last_write.get(src)
        idx = len(self.ins)
        if lw is not None:
            gap = idx - lw - 1      # instructions in between
            if gap < 2:
                self.ins.append(("nop", 2 - gap))

    def dpp_add(self, dst, src, ctrl, bank):
        """dst[lanes in bank] = src[partner lane] + src[own lane]"""
        self._hazard(src)
        f = xor_src(ctrl)
        new = [Counter(x) for x in self.reg[dst]]
        for c in range(16):
            if not (bank >> (c // 4)) & 1:
                continue
            p = f(c)
            assert 0 <= p < 16, (ctrl, c)
            new[c] = self.reg[src][p] + self.reg[src][c]
        self.reg[dst] = new
        self.ins.append(("dpp", dst, src, ctrl, bank))
        self.last_write[dst] = len(self.ins) - 1

    def cnd(self, dst, a, b, bit):
        """dst = lane bit `bit` set ? b : a"""
        new = []
        for c in range(16):
            new.append(Counter(self.reg[b][c] if (c >> bit) & 1 else self.reg[a][c]))
        self.reg[dst] = new
        self.ins.append(("cnd", dst, a, b, bit))
        self.last_write[dst] = len(self.ins) - 1


def build(ns):
    n = Net(ns)
    A = lambda j: "A%d" % j
    B = lambda j: "B%d" % j
    # step 1: lanes 0-7 (banks 0,1) keep entry 0, lanes 8-15 (banks 2,3) entry 1
    for j in range(ns):
        n.dpp_add(A(j), A(j), "row_ror:8", 0x3)
    for j in range(ns):
        n.dpp_add(A(j), B(j), "row_ror:8", 0xc)
    # step 2: lanes with bit 2 clear keep the first ceil(ns/2) slots, the others the rest (moved down)
    k = (ns + 1) // 2
    lo = list(range(k))               # slots kept by bit2 = 0 lanes
    hi = list(range(k, ns))           # slots kept by bit2 = 1 lanes, stored in A[0..len(hi))
    for j in range(len(hi)):
        n.dpp_add(A(j), A(j), "row_shl:4", 0x5)
        n.dpp_add(A(j), A(hi[j]), "row_shr:4", 0xa)
    for j in range(len(hi), k):       # slots only the bit2 = 0 lanes keep
        n.dpp_add(A(j), A(j), "row_shl:4", 0x5)
    slots = {0: lo, 1: hi}            # by bit2
    # step 3 (lane ^ 2): k values -> ceil(k/2)
    for j in range(k):
        n.dpp_add(A(j), A(j), "quad_perm:[2,3,0,1]", 0xf)
    k3 = (k + 1) // 2
    for j in range(k - k3):
        n.cnd(A(j), A(j), A(k3 + j), 1)
    # step 4 (lane ^ 1): k3 values -> ceil(k3/2); do the registers not touched by the selects first (hazard distance)
    order = list(range(k - k3, k3)) + list(range(k - k3))
    for j in order:
        n.dpp_add(A(j), A(j), "quad_perm:[1,0,3,2]", 0xf)
    k4 = (k3 + 1) // 2
    for j in range(k3 - k4):
        n.cnd(A(j), A(j), A(k4 + j), 0)
    assert k4 <= 2, "more than two registers left per lane"
    # which slot does lane h of a half-row hold in A0 / A1?
    def held(h, r):
        b2, b1, b0 = (h >> 2) & 1, (h >> 1) & 1, h & 1
        s2 = slots[b2]
        s3 = (s2[:k3] if not b1 else s2[k3:]) + [None] * k3
        s3 = s3[:k3]
        s4 = (s3[:k4] if not b0 else s3[k4:]) + [None] * k4
        return s4[r] if r < k4 else None
    slot_r0 = [held(h, 0) for h in range(8)]
    slot_r1 = [held(h, 1) for h in range(8)]
    # verify: the advertised registers hold the full row sum of their slot of their entry, every term exactly once
    for c in range(16):
        e, h = c >> 3, c & 7
        for r, tab in ((0, slot_r0), (1, slot_r1)):
            s = tab[h]
            if s is None:
                continue
            want = Counter({(l, "%s%d" % ("ab"[e], s)): 1 for l in range(16)})
            got = n.reg["A%d" % r][c]
            assert got == want, "lane %d reg %d slot %s: %r" % (c, r, s, got)
    seen = sorted(s for t in (slot_r0, slot_r1) for s in t if s is not None)
    assert seen == list(range(ns)), seen
    return n, slot_r0, slot_r1


def emit(n, ns):
    """inline-asm text; operands: %0..%(ns-1) = A (in/out), %ns..%(2ns-1) = B (in), then the two lane-bit masks
    (bit 1 set: 0xCCCC..., bit 0 set: 0xAAAA...)"""
    op = {}
    for j in range(ns):
        op["A%d" % j] = "%%%d" % j
        op["B%d" % j] = "%%%d" % (ns + j)
    m = {1: "%%%d" % (2 * ns), 0: "%%%d" % (2 * ns + 1)}
    lines = ['"s_nop 1\\n\\t"']
    count = 0
    for ins in n.ins:
        if ins[0] == "nop":
            lines.append('"s_nop %d\\n\\t"' % (ins[1] - 1))
        elif ins[0] == "dpp":
            _, d, s, ctrl, bank = ins
            lines.append('"v_add_f32_dpp %s, %s, %s %s row_mask:0xf bank_mask:0x%x\\n\\t"' % (op[d], op[s], op[s], ctrl, bank))
            count += 1
        else:
            _, d, a, b, bit = ins
            lines.append('"v_cndmask_b32_e64 %s, %s, %s, %s\\n\\t"' % (op[d], op[a], op[b], m[bit]))
            count += 1
    return lines, count


if __name__ == "__main__":
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    net, r0, r1 = build(ns)
    lines, count = emit(net, ns)
    print("// NS = %d: %d VALU instructions for two entries; lane h = lane & 7 of a half-row holds" % (ns, count))
    print("//   A0: slots %r" % (r0,))
    print("//   A1: slots %r" % (r1,))
    print("\n".join("\t\t\t" + l for l in lines))
