#!/usr/bin/env python3
"""Generates the hand-scheduled instruction blocks of the compress kernel (gpu-wah_amd/csrc/wah_compress.hip):

  classify_pass1.inc         eight steps (0-7 or 8-15, by the operands bound to it) of PASS 1: which of the 64 groups of
                             a step end a run.  The 64-lane mask of a step lives in VCC just long enough to be counted
                             on the scalar unit (the sum = the words the segment compresses to) and to be shifted, lane
                             by lane, into a per-lane flag word: v_addc_co f = f + f + vcc.  7 vector + 2 scalar
                             instructions per step, software-pipelined by one step so that every hazard slot (a DPP
                             source must be two instructions old) is filled with the NEXT step's independent work.
  classify_pass2_{a,b}.inc   steps 0-7 / 8-15 of PASS 2: the step's mask comes back out of the flag word (v_add_co
                             f = f + f: the carry IS the mask), then the rank of every run end (v_mbcnt over the mask,
                             seeded with the running count) and the compaction stores: the group's value at its rank,
                             its position beside it.  8.5 vector + 2 LDS instructions per step.
  classify_pass2_skip_*.inc  the same for segments that compressed to few words: steps without a run end are branched over.

Pass 1 needs nothing but the groups and leaves ONE register per segment behind; the kernel publishes the word counts of
all its segments right after it and runs pass 2 while the other workgroups' counts are on their way.
(Two halves each: an asm statement takes at most 30 operands.)

Operands: x0..x8 the groups of the half's steps and of the step after it (the one after step 15 = "never equal"),
f the flag word (pass 1: bit 15 - s = step s; pass 2 expects it shifted to the top: bit 31 - s), na/ta and nb/tb two
sets of temporaries (alternating between steps), cnt the scalar count, st a scalar temporary; pass 2: ps the packed
position words, cn the running count in a VECTOR register (so that the ranking needs no scalar work), ln2 the lane id in
both halves of a register, vb/pb LDS byte bases of the value / position arrays, dm the dump slot index (lanes that end
no run store there: cheaper than masking EXEC)."""
import os

HALF = 8


def temps(s):
    return ("na", "ta") if s % 2 == 0 else ("nb", "tb")


def pass1():
    out = []
    emit = out.append
    n, t = temps(0)  # prologue: carry and fill test of the half's first step
    emit(f"v_mov_b32_dpp %[{n}], %[x1] wave_rol:1 row_mask:0xf bank_mask:0xf")
    emit(f"v_add_u32 %[{t}], 1, %[x0]")
    emit(f"v_and_b32 %[{t}], 0x7ffffffe, %[{t}]")
    for s in range(HALF):
        n, t = temps(s)
        n2, t2 = temps(s + 1)
        last = s == HALF - 1
        emit(f"v_mov_b32_dpp %[{n}], %[x{s}] wave_shl:1 row_mask:0xf bank_mask:0xf")  # next group (lane 63 keeps the carry)
        emit(f"v_bitop3_b32 %[{n}], %[x{s}], %[{n}], %[{t}] bitop3:0xbe")               # z = (x ^ next) | t
        emit(f"v_cmp_ne_u32 vcc, 0, %[{n}]")                                             # run ends
        if not last:  # the next step's independent work fills the hazard slots
            emit(f"v_mov_b32_dpp %[{n2}], %[x{s + 2}] wave_rol:1 row_mask:0xf bank_mask:0xf")
            emit(f"v_add_u32 %[{t2}], 1, %[x{s + 1}]")
            emit(f"v_and_b32 %[{t2}], 0x7ffffffe, %[{t2}]")
        else:
            emit("s_nop 1")
        emit("s_bcnt1_i32_b64 %[st], vcc")
        emit("v_addc_co_u32 %[f], vcc, %[f], %[f], vcc")  # f = 2 f + (this lane ends a run); clobbers vcc
        emit("s_add_u32 %[cnt], %[cnt], %[st]")
    return out


def pass2(half):
    out = []
    emit = out.append

    def positions(g):  # positions 64 g + lane and 64 (g + 1) + lane, packed
        emit(f"v_add_u32 %[ps], 0x{((64 * (g + 1)) << 16) | (64 * g):08x}, %[ln2]")

    # the mask of a step must be two instructions old before it is read as data: the address arithmetic and the stores
    # of the step before fill those slots (in front of the first step: wait states)
    emit("v_add_co_u32 %[f], vcc, %[f], %[f]")  # vcc = the step's run ends (top bit of every lane's flag word)
    emit("s_nop 1")
    for s in range(HALF):
        g = half * HALF + s  # step of the segment
        n, t = temps(s)
        last = s == HALF - 1
        emit(f"v_mbcnt_lo_u32_b32 %[{n}], vcc_lo, %[cn]")
        emit(f"v_mbcnt_hi_u32_b32 %[{n}], vcc_hi, %[{n}]")
        emit("v_bcnt_u32_b32 %[cn], vcc_lo, %[cn]")
        emit("v_bcnt_u32_b32 %[cn], vcc_hi, %[cn]")
        if g % 2 == 0:
            positions(g)
        emit(f"v_cndmask_b32 %[{n}], %[dm], %[{n}], vcc")  # non-end lanes: dump slot
        if not last:
            emit("v_add_co_u32 %[f], vcc, %[f], %[f]")  # the next step's mask
        emit(f"v_lshl_add_u32 %[{t}], %[{n}], 2, %[vb]")
        emit(f"v_lshl_add_u32 %[{n}], %[{n}], 1, %[pb]")
        emit(f"ds_write_b32 %[{t}], %[x{s}]")
        emit(f"ds_write_b16{'_d16_hi' if g % 2 else ''} %[{n}], %[ps]")
    return out


def pass2_skip(half):
    """PASS 2 for a segment that compressed to few words (known from pass 1): a step without a single run end -- the
    inside of a long fill -- costs one vector and two scalar instructions instead of eleven."""
    out = []
    emit = out.append
    for s in range(HALF):
        g = half * HALF + s
        n, t = temps(s)
        emit("v_add_co_u32 %[f], vcc, %[f], %[f]")  # vcc = the step's run ends
        emit("s_cmp_eq_u64 vcc, 0")
        emit(f"s_cbranch_scc1 .Lwah_p2skip_%=_{g}")
        emit(f"v_mbcnt_lo_u32_b32 %[{n}], vcc_lo, %[cn]")
        emit(f"v_mbcnt_hi_u32_b32 %[{n}], vcc_hi, %[{n}]")
        emit("v_bcnt_u32_b32 %[cn], vcc_lo, %[cn]")
        emit("v_bcnt_u32_b32 %[cn], vcc_hi, %[cn]")
        emit(f"v_add_u32 %[ps], 0x{64 * g:x}, %[ln2]")  # position 64 g + lane (low half)
        emit(f"v_cndmask_b32 %[{n}], %[dm], %[{n}], vcc")
        emit(f"v_lshl_add_u32 %[{t}], %[{n}], 2, %[vb]")
        emit(f"v_lshl_add_u32 %[{n}], %[{n}], 1, %[pb]")
        emit(f"ds_write_b32 %[{t}], %[x{s}]")
        emit(f"ds_write_b16 %[{n}], %[ps]")
        emit(f".Lwah_p2skip_%=_{g}:")
    return out


here = os.path.dirname(os.path.abspath(__file__))
for name, lines in (("classify_pass1.inc", pass1()), ("classify_pass2_a.inc", pass2(0)), ("classify_pass2_b.inc", pass2(1)),
                    ("classify_pass2_skip_a.inc", pass2_skip(0)), ("classify_pass2_skip_b.inc", pass2_skip(1))):
    with open(os.path.join(here, "..", "gpu-wah_amd", "csrc", name), "w") as f:
        f.write("// GENERATED by tools/gen_classify_block.py -- do not edit (see there for the schedule)\n")
        for line in lines:
            f.write('"' + line + '\\n\\t"\n')
    print(f"{name}: {len(lines)} instructions, {sum(1 for l in lines if l.startswith('v_'))} vector")
