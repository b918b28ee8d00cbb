#!/usr/bin/env python3
"""Generates diner_amd/csrc/f16_core.inc: the assembly-level GEMM core of points_mlp_f16.hip.

    python3 diner_amd/csrc/gen_f16_core.py [RING] > diner_amd/csrc/f16_core.inc      (the Makefile does this)

Why a generator: the fused point/MLP kernel keeps its whole hidden state in registers, and hipcc's
allocation of that 256-register kernel decided its speed (round 1: 147 spilled VGPRs, every
restructuring attempt ended in 400+).  The core therefore owns its registers.  The kernel is built with
__attribute__((amdgpu_num_vgpr(CAP))), CAP = 256 - 128 - 16*RING: hipcc allocates v0..v[CAP-1] only and treats every
higher VGPR as reserved, and exactly those are the core's (named literally; the clobber lists make the kernel
descriptor allocate them):

  v[CAP   : CAP+63]    x    accumulator grid of this wave (64 features x 64 points, 4 tiles of 32x32)
  v[CAP+64: CAP+127]   net  second accumulator grid (the residual block's inner activation)
  v[CAP+128: 255]      the weight ring: RING k-blocks x (2 feature tiles x hi/lo) fragments, written by
                       global_load and read by the MFMAs as their A operand -- it stays in flight across
                       barriers, layers and the compiler's glue code
  v0..v[CAP-1]         the activation fragments (two buffers) and addresses, as ordinary asm temporaries
(A first version kept these in AGPRs; hipcc then used the same AGPRs as spill space and even as load destinations
between the asm statements.  tests/test_isa_guard.py checks on the generated ISA that the compiler stays below CAP,
uses no AGPR and no scratch.)

One statement = one layer for one wave:   s_waitcnt lgkmcnt(0); s_barrier; G1; s_barrier; G2; s_barrier
  G1 = k-blocks of LDS region 0 (k <  256: the operand rows written by waves 0-3)
  G2 = k-blocks of LDS region 1 (k >= 256: written by waves 4-7)
The three barriers per layer are what the half-layer stagger of the two wave groups is built on
(points_mlp_f16.hip, "schedule").

Per k-block (16 k): 4 fragment loads of 16 B per lane from L2 (RING-1 k-blocks ahead; at the end of a
layer they run on into the NEXT layer's first k-blocks, so the ring never drains), 4 ds_read_b128 (the next
k-block's activation fragments, one ahead), counted waits, 12 MFMAs (v_mfma_f32_32x32x16_f16; per product
W_lo*a_hi, W_hi*a_lo, W_hi*a_hi: small terms first).

Weight stream of one (wave, feature tile): contiguous over k-blocks, 2 KiB per k-block: [hi 64 lanes x 16 B | lo].
LDS operand image (bytes): unit-row u = k/8 holds rows (points) 0..63 x 16 B: hi at u*2048, lo at
u*2048 + 1024.  Lane (r = lane&31, hh = lane>>5) reads for k-block kb: u = 2*kb + hh, point r (+32).
"""
import sys

CAP = 96                                        # set by main(): 256 - 128 - 16 * RING
PRIO = 0                                        # --prio=N: s_setprio N for the S phases, 0 inside the layer blocks
SPIN_LIMIT = 1 << 18                            # polls before a flow-mode wait gives up (~25 ms; a real wait lasts microseconds)
PRODS = "small"                                 # --prods=share: W_hi*a_lo, W_hi*a_hi, W_lo*a_hi (neighbours share an operand; probe)
ORDER = "prod"                                  # --order=acc|prod: MFMA order inside a point tile of a half-step (shape 16)
SHAPE = 32                                      # --shape=16: v_mfma_f32_16x16x32_f16 (Block16), default v_mfma_f32_32x32x16_f16
PRIO_B = 0                                      # --priob=N: GEMM priority of waves 4-7 (waves 0-3: 0)
WBITS = ""                                      # --wbits=nt|sc0|...: cache-policy bits of the weight-stream loads
STAMPS = False                                  # --stamps: diagnostic layer blocks only (namespace of --ns), 6 s_memtime stamps each (tools/trace_f16.py)
FLOW = False                                    # --flow: arrival counters in LDS instead of the three workgroup barriers per layer
SLEEP = 1                                       # --sleep=N: s_sleep between two polls of an arrival counter (N x 64 clocks)
X_OFF, NET_OFF, RING_OFF = 0, 64, 128           # relative to CAP
POISON_OFF = 24                                 # byte offset (from the arrival counters) of the workgroup's poison word, see Block.wait


def acc(base, tn, tp):
    lo = CAP + base + 16 * (2 * tn + tp)
    return f"v[{lo}:{lo + 15}]"


def ring(slot, tn, part):  # part 0 = hi, 1 = lo
    lo = CAP + RING_OFF + 16 * slot + 8 * tn + 4 * part
    return f"v[{lo}:{lo + 3}]"


class Block:
    noload = False   # probe-only ablation: the MFMAs read whatever the ring holds, no weight stream

    def __init__(self, name, acc_base, nkb1, nkb2, D, region1_off=65536):
        assert D in (2, 4) and nkb1 % D == 0 and nkb2 % D == 0 and nkb1 > 0
        self.name, self.base, self.nkb1, self.nkb2, self.D = name, acc_base, nkb1, nkb2, D
        self.region1_off = region1_off     # byte offset of the operand rows written by waves 4-7 (k >= 16 * nkb1)
        self.lines = []
        self.nlabel = 0

    def e(self, s):
        self.lines.append(s)

    def loads(self, slot):
        """the 4 fragment loads of one k-block into ring slot `slot`; the lane offset then moves on one k-block"""
        for tn, w in ((0, "%[w0]"), (1, "%[w1]")):
            if self.noload:
                continue
            self.e(f"global_load_dwordx4 {ring(slot, tn, 0)}, %[voff], {w}{WBITS}")
            self.e(f"global_load_dwordx4 {ring(slot, tn, 1)}, %[voff], {w} offset:1024{WBITS}")
        self.e("v_add_u32 %[voff], 2048, %[voff]")

    def switch_to_next(self):
        self.e("s_mov_b64 %[w0], %[nw0]")
        self.e("s_mov_b64 %[w1], %[nw1]")
        self.e("v_mov_b32 %[voff], %[loff]")

    def frag_reads(self, buf, imm):
        # order: hi tp0, hi tp1, lo tp0, lo tp1
        for i in range(4):
            self.e(f"ds_read_b128 %[f{buf}{i}], %[ab] offset:{imm + 512 * i}")

    def mfmas(self, slot, buf):
        b = self.base
        hi = [f"%[f{buf}0]", f"%[f{buf}1]"]
        lo = [f"%[f{buf}2]", f"%[f{buf}3]"]
        prods = ((1, hi), (0, lo), (0, hi))                # W_lo*a_hi, W_hi*a_lo, W_hi*a_hi
        order = [(p, tn, tp) for tn in range(2) for tp in range(2) for p in prods] if ORDER == "acc" else \
                [(p, tn, tp) for p in prods for tn in range(2) for tp in range(2)]
        for (wpart, frags), tn, tp in order:               # --order=acc: the 3 products of an accumulator back to back (see Block16)
            self.e(f"v_mfma_f32_32x32x16_f16 {acc(b, tn, tp)}, {ring(slot, tn, wpart)}, {frags[tp]}, {acc(b, tn, tp)}")

    # ---- synchronisation -------------------------------------------------------------------------------------------------
    # barrier mode: s_barrier.  flow mode: four monotonic counters in LDS at %[ctr] (byte offsets 0 SA, 4 SB, 8 G1, 12 G2):
    #   SA / SB  waves 0-3 / 4-7 that have finished their S phase (operand rows written): +4 per layer each
    #   G1 / G2  waves that have finished reading region 0 / region 1 of the layer: +8 per layer each
    # %[lay] = number of the layer being executed (1, 2, ...: every wave counts the same sequence); %[half] = 0 / 1.
    def signal(self, off_expr):
        """one lane adds 1 to the counter at %[ctr] + off (off_expr: an immediate or 'half' for 4 * %[half])"""
        self.e("s_waitcnt lgkmcnt(0)")
        self.e("s_mov_b64 %[ex], exec")
        self.e("s_mov_b64 exec, 1")
        if off_expr == "half":
            self.e("ds_add_u32 %[ctrh], %[one]")
        elif off_expr == "half+8":
            self.e("ds_add_u32 %[ctrh], %[one] offset:8")
        else:
            self.e(f"ds_add_u32 %[ctr], %[one] offset:{off_expr}")
        self.e("s_mov_b64 exec, %[ex]")

    def stamp(self, i):
        """diagnostic build: clock into st.t[i]; SMEM returns out of order with LDS, so drain before any counted lgkmcnt wait"""
        if STAMPS:
            self.e(f"s_memtime %[tk{i}]")
            self.e("s_waitcnt lgkmcnt(0)")

    def wait(self, addr, off, mult):
        """spin (with s_sleep) until counter >= mult * %[lay]"""
        lbl, done = f"W{self.nlabel}_%=", f"D{self.nlabel}_%="
        self.nlabel += 1
        self.e(f"s_mul_i32 %[tgt], %[lay], {mult}")
        self.e(f"s_mov_b32 %[spin], {SPIN_LIMIT}")                   # bounded: a protocol error must end as a wrong result, never as a hung GPU
        self.e(f"{lbl}:")
        self.e(f"ds_read_b32 %[pv], %[{addr}] offset:{off}")
        self.e("s_waitcnt lgkmcnt(0)")
        self.e("v_readfirstlane_b32 %[cnt], %[pv]")
        self.e("s_cmp_ge_u32 %[cnt], %[tgt]")
        self.e(f"s_cbranch_scc1 {done}")
        self.e("s_sub_u32 %[spin], %[spin], 1")
        self.e("s_cmp_eq_u32 %[spin], 0")
        self.e(f"s_cbranch_scc1 T{lbl}")
        self.e(f"s_sleep {SLEEP}")
        self.e(f"s_branch {lbl}")
        # timed out (a protocol error; unreachable in a correct build): the wave goes on with operand rows nobody vouches for, so the
        # WHOLE tile must come out non-finite, not just this wave's share: set the workgroup's poison word (LDS, %[ctr] + POISON_OFF;
        # never cleared).  The kernels read it where their results leave: the point kernel's head writes NaN for every sample of the
        # tile (-> DINER_STATUS_NONFINITE), the training core's epilogue writes NaN into C and inf into amax_out.  Loud, no hang.
        self.e(f"T{lbl}:")
        self.e(f"v_mov_b32 v{CAP + X_OFF}, 0x7fc00000")
        self.e("v_mov_b32 %[pv], 1")
        self.e(f"ds_write_b32 %[ctr], %[pv] offset:{POISON_OFF}")
        self.e(f"{done}:")

    def body(self, tail, switch):
        """D consecutive k-blocks: k-block j of the iteration consumes ring slot j and fragment buffer j & 1 (D is even).
        tail = the half's last iteration (its last k-block prefetches no fragments); switch = the layer's last
        iteration (after its first load group the weight stream moves on to the next layer)."""
        D = self.D
        for j in range(D):
            self.loads((j + D - 1) % D)                              # k-block kb + D - 1
            if switch and j == 0:
                self.switch_to_next()
            if j < D - 1:
                self.frag_reads((j + 1) & 1, (j + 1) * 4096)
                pending = 4
            elif not tail:
                self.e(f"v_add_u32 %[ab], {D * 4096}, %[ab]")       # the next iteration's first k-block
                self.frag_reads((j + 1) & 1, 0)
                pending = 4
            else:
                pending = 0
            self.e(f"s_waitcnt vmcnt({4 * (D - 1)})")                # this k-block's weights: all but the D-1 younger groups
            self.e(f"s_waitcnt lgkmcnt({pending})")                  # this k-block's fragments
            self.mfmas(j, j & 1)

    def half(self, nkb, region, last_half):
        iters = nkb // self.D
        self.e(f"v_add_u32 %[ab], {self.region1_off * region}, %[ab0]")
        self.frag_reads(0, 0)                                        # the half's first k-block, behind the barrier
        if iters > 1:
            lbl = f"L{self.nlabel}_%="
            self.nlabel += 1
            self.e(f"s_mov_b32 %[cnt], {iters - 1}")
            self.e(f"{lbl}:")
            self.body(tail=False, switch=False)
            self.e("s_sub_u32 %[cnt], %[cnt], 1")
            self.e("s_cmp_lg_u32 %[cnt], 0")
            self.e(f"s_cbranch_scc1 {lbl}")
        self.body(tail=True, switch=last_half)

    def both_first(self):
        """lin_in (both its halves in physical region 0, region1_off != 65536): the kernel is free to deal its 8 unit-rows to the 8 waves in
        any order (points_mlp_f16.hip: the rows that need the depth fetch go to waves 0-3, which have slack), so the block waits for BOTH
        arrival counters before its first half instead of one per half.  The block is two k-steps long: nothing is lost."""
        return FLOW and self.region1_off != 65536 and self.nkb2 > 0

    def emit(self):
        D = self.D
        step = 4096 if isinstance(self, Block16) else 2048
        self.e(f"v_add_u32 %[voff], {(D - 1) * step}, %[loff]")      # k-blocks 0..D-2 of this layer are already in the ring
        if PRIO:
            self.e("s_setprio 0")                                     # GEMM at low priority: the partner's S phase (VALU, LDS, gather) goes first
            if PRIO_B:                                                # ... and waves 4-7 (the critical path: they never wait) ahead of waves 0-3
                self.e("s_cmp_lg_u32 %[half], 0")
                self.e(f"s_cbranch_scc0 PB{self.nlabel}_%=")
                self.e(f"s_setprio {PRIO_B}")
                self.e(f"PB{self.nlabel}_%=:")
                self.nlabel += 1
        if FLOW:
            self.stamp(0)
            self.signal("half")                                       # my operand rows are written: SA or SB += 1
            self.wait("ctr", 0, 4)                                    # G1 reads region 0: all of waves 0-3 have written theirs (SA >= 4 lay)
            if self.both_first():
                self.wait("ctr", 4, 4)                                # lin_in: ANY wave may have written ANY of its 8 unit-rows: both groups first
            self.stamp(1)
        else:
            self.e("s_waitcnt lgkmcnt(0)")                            # this wave's operand stores (its S phase) have landed
            self.e("s_barrier")
        self.half(self.nkb1, 0, last_half=(self.nkb2 == 0))
        if FLOW:
            self.stamp(2)
            self.signal(8)                                            # G1 += 1: my reads of region 0 are over
            if self.nkb2 and not self.both_first():
                self.wait("ctr", 4, 4)                                # G2 reads region 1: SB >= 4 lay
            self.stamp(3)
        else:
            self.e("s_barrier")
        if self.nkb2:
            self.half(self.nkb2, 1, last_half=True)
        self.e("s_nop 15")                                            # MFMA results -> VALU reads in the glue (XDL write -> VALU read wait states)
        self.e("s_nop 7")
        if FLOW:
            self.stamp(4)
            self.signal(12)                                           # G2 += 1 (a layer without a second half signals right away: the counts stay in step)
            # before this wave may overwrite its operand rows (its next S phase): everybody has finished reading them.  Waves 0-3 own
            # region 0 (readers: G1), waves 4-7 region 1 (readers: G2).  lin_in keeps BOTH its halves in physical region 0 (unit-rows 0-7),
            # so behind it waves 0-3 wait for G2 as well; waves 4-7 always see G1 complete once G2 is (each wave runs G1 before G2).
            self.wait("ctrh", 8, 8)
            if self.region1_off != 65536:
                self.wait("ctr", 12, 8)
            self.stamp(5)
        else:
            self.e("s_barrier")
        if PRIO:
            self.e(f"s_setprio {PRIO}")                               # the glue code that follows (this wave's S phase) outranks the partner's MFMAs
        return self.lines


class Block16(Block):
    """The same layer block on v_mfma_f32_16x16x32_f16 (--shape=16).  Why: at the chip's power limit the 16x16x32 shape sustains a
    higher clock than 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md: 1.12-1.15x the FLOP/s), and this kernel is energy-bound.

    Accumulators: 16 tiles of 16 features x 16 points, 4 registers each: register i of tile (tf, tp) = feature 64w + 16tf + 4q + i of point
    16tp + c16, lane = 16q + c16 (v[CAP + base + 4 (4 tf + tp) + i]).
    A k-step is 32 k = two half-steps: half j works on feature tiles 2j, 2j + 1 with ring slot j (16 registers: [tf2][hi/lo] x 4) for all 4
    point tiles -- so the ring keeps its size and its one-half-step prefetch distance -- and the 8 activation fragments of the k-step
    (point tile x hi/lo, 32 registers, ONE set) are reloaded on the fly: in half 1, as soon as a point tile's 6 MFMAs have been issued,
    its two fragments are re-read for the next k-step (18 MFMAs = 288+ cycles ahead of their first use).
    Weight stream of a wave: ONE sequence of half-steps, 4 KiB each: [tf2][hi/lo][lane][8 halfs], lane = 16 kq + r holds
    W[feature 16 (2j + tf2) + r][k = 32 kstep + 8 kq ..+7].  nkb1 / nkb2 count k-steps."""

    def __init__(self, name, acc_base, nks1, nks2, D, region1_off=65536, zero_init=False, convert_tail=False, bias_init=False):
        assert D == 2 and nks1 > 0
        # bias_init: acc := bias + W a without an initialisation pass -- the first product of every accumulator takes the bias as its C
        # operand: 4 register quads %[bq0..3] (features 64w + 16tf + 4q ..+3: the same for the 4 point tiles of a lane), read from LDS
        # by the caller.  The first k-step is peeled off the loop for it.
        self.bias_init = bias_init
        assert not (bias_init and zero_init)
        # convert_tail: the layer's result is needed ONLY as the next layer's operand (net of a residual block: relu -> fp16 hi/lo
        # split -> LDS).  The relu + split then happens IN PLACE in the layer's last k-step, tile by tile behind the tile's last product
        # (two tiles later: far beyond the MFMA write -> VALU read wait states), in the shadow of the remaining MFMAs, instead of in the
        # wave's next S phase, which is on the layer schedule's critical path.  Afterwards registers 0..3 of a tile hold
        # lo(v0,v1) | hi(v0,v1) | lo(v2,v3) | hi(v2,v3) as packed fp16 pairs; the glue only stores them (store_converted).
        self.convert_tail = convert_tail
        self.converting = False
        self.name, self.base, self.nkb1, self.nkb2, self.D = name, acc_base, nks1, nks2, D
        self.region1_off = region1_off
        # zero_init: the accumulators need no initialisation by the caller -- the first product of every accumulator takes the inline
        # constant 0 as its C operand (D = A*B + 0).  Only for blocks whose first half is ONE k-step (lin_in), where "first" is static.
        self.zero_init = zero_init
        assert not zero_init or nks1 == 1
        self.first_step = False
        self.lines = []
        self.nlabel = 0

    def acc16(self, tf, tp):
        lo = CAP + self.base + 4 * (4 * tf + tp)
        return f"v[{lo}:{lo + 3}]"

    @staticmethod
    def fr(tp, part):
        return f"%[f{tp // 2}{(tp % 2) * 2 + part}]"

    def loads(self, slot):
        for i in range(4):                                            # (tf2, part) = (i >> 1, i & 1)
            if not self.noload:
                self.e(f"global_load_dwordx4 {ring(slot, i >> 1, i & 1)}, %[voff], %[w0] offset:{1024 * i}{WBITS}")
        self.e("v_add_u32 %[voff], 4096, %[voff]")

    def switch_to_next(self):
        self.e("s_mov_b64 %[w0], %[nw0]")
        self.e("v_mov_b32 %[voff], %[loff]")

    def frag_read(self, tp, imm=0):
        for part in range(2):
            self.e(f"ds_read_b128 {self.fr(tp, part)}, %[ab] offset:{imm + 256 * tp + 1024 * part}")

    def mfmas16(self, j, tp):
        prods = ((0, 1), (0, 0), (1, 0)) if PRODS == "share" else ((1, 0), (0, 1), (0, 0))   # default: W_lo*a_hi, W_hi*a_lo, W_hi*a_hi (small terms first)
        pairs = [(p, tf2) for tf2 in range(2) for p in prods] if ORDER == "acc" else [(p, tf2) for p in prods for tf2 in range(2)]
        for (wpart, fpart), tf2 in pairs:                             # --order=acc: the 3 products of an accumulator back to back
            a = self.acc16(2 * j + tf2, tp)
            c = a
            if self.first_step and (wpart, fpart) == prods[0]:        # the accumulator's first product of the layer
                c = "0" if self.zero_init else f"%[bq{2 * j + tf2}]"  # zero_init / bias_init
            self.e(f"v_mfma_f32_16x16x32_f16 {a}, {ring(j, tf2, wpart)}, {self.fr(tp, fpart)}, {c}")

    def convert(self, tf, tp):
        """relu (keeps every NaN, like torch.relu and relu_split4<SAFE>) + hi/lo split of accumulator tile (tf, tp), in place"""
        r = CAP + self.base + 4 * (4 * tf + tp)
        for i in range(4):
            self.e(f"v_cmp_ngt_f32_e64 %[m{i}], 0, v{r + i}")
        for i in range(4):
            self.e(f"v_cndmask_b32_e64 v{r + i}, 0, v{r + i}, %[m{i}]")
        for a, b in ((r, r + 1), (r + 2, r + 3)):
            self.e(f"v_cvt_pk_f16_f32 %[cva], v{a}, v{b}")
            self.e(f"v_fma_mixlo_f16 v{a}, %[cva], -1.0, v{a} op_sel_hi:[1,0,0]")                       # lo16 of v_a := f16(v_a - hi_a); the rest of v_a stays
            self.e(f"v_fma_mixhi_f16 v{a}, %[cva], -1.0, v{b} op_sel:[1,0,0] op_sel_hi:[1,0,0]")       # hi16 of v_a := f16(v_b - hi_b)
            self.e(f"v_mov_b32 v{b}, %[cva]")

    def body(self, tail, switch):
        done = []                                                     # (tf, tp) of the tiles whose last product has been issued
        for j in range(2):
            self.loads((j + 1) % 2)                                   # the next half-step's weights into the slot just used up
            if switch and j == 0:
                self.switch_to_next()
            if j == 1 and not tail:
                self.e("v_add_u32 %[ab], 8192, %[ab]")               # the next k-step's fragments (this one's reads are all issued)
            self.e("s_waitcnt vmcnt(4)")                              # this half-step's weights: all but the 4 loads just issued
            for tp in range(4):
                if j == 0:
                    self.e(f"s_waitcnt lgkmcnt({6 - 2 * tp})")       # fragments arrive in issue order, two per point tile
                self.mfmas16(j, tp)
                if j == 1 and not tail:
                    self.frag_read(tp)                                # rolling reload: needed again 18 MFMAs from here
                if self.converting:                                   # the layer's last k-step: convert the tile that finished two tiles ago
                    for tf2 in range(2):
                        done.append((2 * j + tf2, tp))
                    while len(done) > 2:
                        self.convert(*done.pop(0))
        if self.converting:
            self.e("s_nop 15")                                        # the last two tiles: MFMA write -> VALU read wait states
            self.e("s_nop 7")
            for t in done:
                self.convert(*t)

    def half(self, nks, region, last_half):
        self.e(f"v_add_u32 %[ab], {self.region1_off * region}, %[ab0]")
        for tp in range(4):
            self.frag_read(tp)                                        # the half's first k-step, behind the barrier / arrival wait
        self.first_step = self.zero_init and region == 0              # (nks == 1 there: the tail body below is the layer's first k-step)
        peel = self.bias_init and region == 0
        if peel:                                                      # the layer's first k-step, peeled: C = bias
            assert nks > 2
            self.first_step = True
            self.body(tail=False, switch=False)
            self.first_step = False
            nks -= 1
        if nks > 1:
            lbl = f"L{self.nlabel}_%="
            self.nlabel += 1
            self.e(f"s_mov_b32 %[cnt], {nks - 1}")
            self.e(f"{lbl}:")
            self.body(tail=False, switch=False)
            self.e("s_sub_u32 %[cnt], %[cnt], 1")
            self.e("s_cmp_lg_u32 %[cnt], 0")
            self.e(f"s_cbranch_scc1 {lbl}")
        self.converting = self.convert_tail and last_half
        self.body(tail=True, switch=last_half)
        self.converting = False
        self.first_step = False


def clobbers(D):
    return ", ".join(f'"v{i}"' for i in range(CAP, 256))


def asm_body(lines):
    return "\n".join(f'        "{l}\\n"' for l in lines)


def cxx(block):
    lines = block.emit()
    frag_ops = ", ".join(f'[f{b}{i}] "=&v"(f{b}{i})' for b in range(2) for i in range(4))
    cv = getattr(block, "convert_tail", False)
    cv_decl = "    unsigned cva; unsigned long long m0, m1, m2, m3;\n" if cv else ""
    cv_outs = ', [cva] "=&v"(cva), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)' if cv else ""
    bi = getattr(block, "bias_init", False)
    bi_args = ", const f32x4 &bq0, const f32x4 &bq1, const f32x4 &bq2, const f32x4 &bq3" if bi else ""
    bi_ins = ', [bq0] "v"(bq0), [bq1] "v"(bq1), [bq2] "v"(bq2), [bq3] "v"(bq3)' if bi else ""
    return f"""
// {block.name}: accumulators v[{CAP + block.base}:{CAP + block.base + 63}], {block.nkb1} + {block.nkb2} k-blocks, ring depth {block.D}
// w0/w1: this wave's weight streams (feature tile 0/1) of THIS layer; nw0/nw1: of the layer executed next
__device__ __forceinline__ void {block.name}(uint64_t w0, uint64_t w1, uint64_t nw0, uint64_t nw1, unsigned loff, unsigned ab0, const Sync &sy{bi_args}{", Stamps &st" if STAMPS else ""})
{{
    h8 f00, f01, f02, f03, f10, f11, f12, f13;
    unsigned ab, voff, cnt;
{cv_decl}{flow_decl()}
    asm volatile(
{asm_body(lines)}
        : {frag_ops}, [ab] "=&v"(ab), [voff] "=&v"(voff), [cnt] "=&s"(cnt), [w0] "+s"(w0), [w1] "+s"(w1){flow_outs()}{cv_outs}{"".join(f', [tk{i}] "=&s"(st.t[{i}])' for i in range(6)) if STAMPS else ""}
        : [nw0] "s"(nw0), [nw1] "s"(nw1), [loff] "v"(loff), [ab0] "v"(ab0){flow_ins()}{bi_ins}
        : "memory", "scc", {clobbers(block.D)});
}}
"""


def gather_cxx(D, GQ=3):
    """x += bilerp(G)(uv) for this wave's 64 features x 64 points (resnetfc.py:152-153 with lin_z hoisted to feature maps).

    The accumulator layout has the POINT on the lane, so a gather straight into it makes every 4-lane step of a load touch 4
    different cache lines: 64 tag look-ups per instruction, and the CU's texture addresser, not latency, bounded the phase
    (2 and 4 quads in flight took the same time).  So the loads are COALESCED instead -- lane = (point of a group of 4, float4 of the
    wave's 256-byte feature slice): 16 lanes read one texel slice contiguously -- the 4 taps are blended in that layout, and the
    result goes through LDS to the accumulator layout: the staging area is this wave's own 8 unit-rows of the operand image
    (16 KiB = 64 points x 256 B), which are dead between the barrier that ends the previous layer and this wave's operand store;
    float4 q of point p sits at p*256 + ((q ^ (p & 15)) * 16): conflict-free for the coalesced writes and the transposed reads.

    Loads land in the `net` grid's registers (dead between a block's fc_1 operand store and the next fc_0's bias init): GQ = 3 groups
    (a group = 4 points x 4 taps = 4 float4 per lane) in flight, the grid's last 16 registers are the routine's float4 temporaries.
    gather_issue() starts the first GQ groups one layer EARLY (they land while the preceding GEMM runs); gather_finish() consumes
    group after group with counted waits, re-issuing into the freed slots.
    Footprint table in LDS (written once per view): per point 4 x u32 byte offsets of the tap texels, 4 x f32 weights."""
    assert GQ == 3
    NG = 16
    NETB = CAP + NET_OFF
    TO, TW, TT, RD = NETB + 48, NETB + 52, NETB + 56, NETB + 60      # float4 temporaries

    def q4(r):
        return f"v[{r}:{r + 3}]"

    def slot(i, tap):
        return NETB + 16 * (i % GQ) + 4 * tap

    def taps_read(i, which):          # offsets (which = 0) or weights (1) of this lane's point of group i
        return f"ds_read_b128 {q4(TO if which == 0 else TW)}, %[tbase] offset:{i * 128 + 16 * which}"

    def issue(i):
        """the byte offsets of group i's taps are in TO"""
        out = [f"v_add_u32 %[a{t}], %[qb], v{TO + t}" for t in range(4)]
        out += [f"global_load_dwordx4 {q4(slot(i, t))}, %[a{t}], %[G]" for t in range(4)]
        return out

    # ---- early issue: groups 0..GQ-1
    first = []
    for i in range(GQ):
        first.append(taps_read(i, 0))
        first.append("s_waitcnt lgkmcnt(0)")
        first += issue(i)
    # ---- finish (early = the first GQ groups were issued before the preceding layer block: the ring's prefetch sits behind them)
    def finish(early):
      fin = [] if early else list(first)
      for i in range(NG):
        fin.append(taps_read(i, 1))                                  # this group's weights
        nxt = i + GQ < NG
        if nxt:
            fin.append(taps_read(i + GQ, 0))                          # offsets of the group that goes into the slot freed below
        gap = 4 * (D - 1) if early else 0                             # the ring's prefetch, issued between the early groups and the rest
        younger = 4 * (GQ - 1) + gap if i < GQ else 4 * min(GQ - 1, NG - 1 - i)
        fin.append(f"s_waitcnt vmcnt({younger})")                     # exact, or stricter when glue loads are outstanding; never weaker
        fin.append(f"s_waitcnt lgkmcnt({1 if nxt else 0})")           # the weights have landed (LDS returns in order)
        for j in range(4):
            fin.append(f"v_mul_f32 v{TT + j}, v{slot(i, 0) + j}, v{TW}")
        for t in range(1, 4):
            for j in range(4):
                fin.append(f"v_fmac_f32 v{TT + j}, v{slot(i, t) + j}, v{TW + t}")
        if nxt:
            fin.append("s_waitcnt lgkmcnt(0)")                        # the next group's offsets
            fin += issue(i + GQ)
        fin.append(f"ds_write_b128 %[sw{i % 4}], {q4(TT)} offset:{i * 1024}")
        fin.append("s_nop 1")                                         # TT is rewritten by the next group's first v_mul: the store must have read it
      # ---- read back in the accumulator layout and add
      fin.append("s_waitcnt lgkmcnt(0)")
      if SHAPE == 16:
        # lane = 16q + c16: float4 4tf + q of points 16tp + c16 (rows 4096 bytes apart); %[hp] = q ^ c16, %[rb] = staging row of point c16
        for tf in range(4):
            fin.append(f"v_xor_b32 %[a0], {4 * tf}, %[hp]")
            fin.append("v_lshl_add_u32 %[a0], %[a0], 4, %[rb]")
            tmp = (TO, TW, TT, RD)
            for tp in range(4):
                fin.append(f"ds_read_b128 {q4(tmp[tp])}, %[a0] offset:{4096 * tp}")
            fin.append("s_waitcnt lgkmcnt(0)")
            for tp in range(4):
                xr = CAP + X_OFF + 4 * (4 * tf + tp)
                for j in range(4):
                    fin.append(f"v_add_f32 v{xr + j}, v{xr + j}, v{tmp[tp] + j}")
      else:
       for tn in range(2):
        for g in range(4):
            K = tn * 8 + 2 * g
            fin.append(f"v_xor_b32 %[a0], {K}, %[hp]")
            fin.append("v_lshl_add_u32 %[a0], %[a0], 4, %[rb]")
            fin.append(f"ds_read_b128 {q4(TT)}, %[a0]")
            fin.append(f"ds_read_b128 {q4(RD)}, %[a0] offset:8192")
            fin.append("s_waitcnt lgkmcnt(0)")
            for tp, src in ((0, TT), (1, RD)):
                xr = CAP + X_OFF + 16 * (2 * tn + tp) + 4 * g
                for j in range(4):
                    fin.append(f"v_add_f32 v{xr + j}, v{xr + j}, v{src + j}")
      return fin
    return f"""
// x += bilerp(G)(uv), coalesced.  tbase = LDS address of the footprint table + 32 * (lane >> 4); qb = wave * 256 + (lane & 15) * 16;
// sw[k] = staging write address of this lane for the groups i with i % 4 == k (the group's i * 1024 goes into the instruction);
// rb = staging row of point c (point 32 + c: 8192 further); hp = h ^ (c & 15)
struct GatherRegs {{ unsigned tbase, qb, sw[4], rb, hp; }};
__device__ __forceinline__ void gather_issue(uint64_t G, const GatherRegs &r)
{{
    unsigned a0, a1, a2, a3;
    asm volatile(
{asm_body(first)}
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3)
        : [G] "s"(G), [tbase] "v"(r.tbase), [qb] "v"(r.qb)
        : "memory", {clobbers(0)});
}}
// EARLY: gather_issue() ran before the preceding layer block; otherwise the whole gather happens here
template <bool EARLY> __device__ __forceinline__ void gather_finish(uint64_t G, const GatherRegs &r);
template <> __device__ __forceinline__ void gather_finish<false>(uint64_t G, const GatherRegs &r)
{{
    unsigned a0, a1, a2, a3;
    asm volatile(
{asm_body(finish(False))}
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3)
        : [G] "s"(G), [tbase] "v"(r.tbase), [qb] "v"(r.qb), [sw0] "v"(r.sw[0]), [sw1] "v"(r.sw[1]), [sw2] "v"(r.sw[2]), [sw3] "v"(r.sw[3]),
          [rb] "v"(r.rb), [hp] "v"(r.hp)
        : "memory", {clobbers(0)});
}}
template <> __device__ __forceinline__ void gather_finish<true>(uint64_t G, const GatherRegs &r)
{{
    unsigned a0, a1, a2, a3;
    asm volatile(
{asm_body(finish(True))}
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3)
        : [G] "s"(G), [tbase] "v"(r.tbase), [qb] "v"(r.qb), [sw0] "v"(r.sw[0]), [sw1] "v"(r.sw[1]), [sw2] "v"(r.sw[2]), [sw3] "v"(r.sw[3]),
          [rb] "v"(r.rb), [hp] "v"(r.hp)
        : "memory", {clobbers(0)});
}}
"""


def flow_decl():
    return "    unsigned pv, tgt, spin; unsigned long long ex;" if FLOW else ""


def flow_outs():
    return ', [pv] "=&v"(pv), [tgt] "=&s"(tgt), [spin] "=&s"(spin), [ex] "=&s"(ex)' if FLOW else ""


def flow_ins():
    return ', [ctr] "v"(sy.ctr), [ctrh] "v"(sy.ctrh), [one] "v"(sy.one), [lay] "s"(sy.lay), [half] "s"(sy.half)' if FLOW else ""


def sync_struct():
    return f"""
// synchronisation state a layer block takes: barrier mode needs none of it; flow mode ({'ON' if FLOW else 'off'} in this build):
// ctr = LDS address of the four arrival counters (SA, SB, G1, G2), ctrh = ctr + 4 * (wave >= 4), one = 1, lay = number of the layer, half = (wave >= 4)
constexpr bool F16_FLOW = {'true' if FLOW else 'false'};
constexpr int F16_POISON_OFF = {POISON_OFF};   // byte offset of the poison word behind the counters: set by a wait that gave up (never in a correct build)
struct Sync {{ unsigned ctr, ctrh, one, lay, half; }};
// one lane adds 1 to the LDS counter at `addr` (after this wave's LDS operations have completed)
__device__ __forceinline__ void flow_signal(unsigned addr, unsigned one)
{{
    unsigned long long ex;
    asm volatile(
{asm_body(["s_waitcnt lgkmcnt(0)", "s_mov_b64 %[ex], exec", "s_mov_b64 exec, 1", "ds_add_u32 %[a], %[one]", "s_mov_b64 exec, %[ex]"])}
        : [ex] "=&s"(ex) : [a] "v"(addr), [one] "v"(one) : "memory");
}}
// spin (with s_sleep) until the LDS counter at `addr` has reached `target`; false if the bounded spin gave up
__device__ __forceinline__ bool flow_wait(unsigned addr, unsigned target)
{{
    unsigned pv, cnt, spin;
    asm volatile(
{asm_body([f"s_mov_b32 %[spin], {SPIN_LIMIT}", "W_%=:", "ds_read_b32 %[pv], %[a]", "s_waitcnt lgkmcnt(0)", "v_readfirstlane_b32 %[cnt], %[pv]", "s_cmp_ge_u32 %[cnt], %[tgt]", "s_cbranch_scc1 D_%=", "s_sub_u32 %[spin], %[spin], 1", "s_cmp_eq_u32 %[spin], 0", "s_cbranch_scc1 D_%=", f"s_sleep {SLEEP}", "s_branch W_%=", "D_%=:"])}
        : [pv] "=&v"(pv), [cnt] "=&s"(cnt), [spin] "=&s"(spin) : [a] "v"(addr), [tgt] "s"(target) : "memory", "scc");
    return spin != 0;
}}
"""


def prologue(D):
    """Fill the ring with k-blocks 0..D-2 of the first layer (once per kernel; afterwards every layer block
    prefetches its successor's first k-blocks)."""
    b = (Block16 if SHAPE == 16 else Block)("prologue", 0, D, 0, D)
    b.e("v_mov_b32 %[voff], %[loff]")
    keep, b.noload = b.noload, False
    for j in range(D if keep else D - 1):                # the no-load ablation still starts from real weights in every slot
        b.loads(j)
    return f"""
__device__ __forceinline__ void ring_prologue(uint64_t w0, uint64_t w1, unsigned loff)
{{
    unsigned voff;
    asm volatile(
{asm_body(b.lines)}
        : [voff] "=&v"(voff)
        : [w0] "s"(w0), [w1] "s"(w1), [loff] "v"(loff)
        : "memory", {clobbers(D)});
}}
"""


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    D = int(args[0]) if args else 4
    ns = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--ns=")), None)
    Block.noload = "--noload" in sys.argv
    global CAP, PRIO
    CAP = 256 - 128 - 16 * D
    PRIO = next((int(a.split('=', 1)[1]) for a in sys.argv[1:] if a.startswith('--prio=')), 0)
    global FLOW, STAMPS, WBITS, PRIO_B, SHAPE, ORDER, PRODS, SLEEP
    SLEEP = next((int(a.split('=', 1)[1]) for a in sys.argv[1:] if a.startswith('--sleep=')), 1)
    PRODS = next((a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--prods=')), "small")
    ORDER = next((a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--order=')), "prod")
    SHAPE = next((int(a.split('=', 1)[1]) for a in sys.argv[1:] if a.startswith('--shape=')), 32)
    assert SHAPE in (16, 32)
    B_ = Block16 if SHAPE == 16 else Block
    NK = 8 if SHAPE == 16 else 16                  # k-steps (32 k) / k-blocks (16 k) per half of a 512-wide layer
    NKI = 1 if SHAPE == 16 else 2                  # ... of lin_in (64 inputs)
    PRIO_B = next((int(a.split('=', 1)[1]) for a in sys.argv[1:] if a.startswith('--priob=')), 0)
    WBITS = next((" " + a.split("=", 1)[1].replace(",", " ") for a in sys.argv[1:] if a.startswith("--wbits=")), "")
    FLOW = '--flow' in sys.argv
    STAMPS = '--stamps' in sys.argv
    if STAMPS:
        assert ns and FLOW and D == 2
        out = [f"// GENERATED by gen_f16_core.py {' '.join(sys.argv[1:])} -- do not edit.  Diagnostic twins of the layer blocks: the same code plus 6 clock\n"
               f"// stamps (entry | region 0 ready | G1 done | region 1 ready | G2 done | operand rows free), used by the TRACE kernel only.\n"
               f"namespace {ns} {{\nstruct Stamps {{ unsigned long long t[6]; }};\n"]
        out.append(cxx(B_("layer_x_full", X_OFF, NK, NK, D)))
        out.append(cxx(B_("layer_net_full", NET_OFF, NK, NK, D)))
        if SHAPE == 16:
            out.append(cxx(B_("layer_net_full_cv", NET_OFF, NK, NK, D, convert_tail=True)))
            out.append(cxx(B_("layer_net_full_b", NET_OFF, NK, NK, D, bias_init=True)))
        out.append(cxx(B_("layer_x_in", X_OFF, NKI, NKI, D, region1_off=8192)))
        if SHAPE == 16:
            out.append(cxx(B_("layer_x_in0", X_OFF, NKI, NKI, D, region1_off=8192, zero_init=True)))
        out.append("}  // namespace " + ns + "\n")
        sys.stdout.write("\n".join(out))
        return
    out = [f"""// GENERATED by gen_f16_core.py (ring depth {D}{", no weight loads: ablation" if Block.noload else ""}) -- do not edit; see the generator for the design.
{"namespace " + ns + " {" if ns else "#pragma once"}
// generator arguments: {" ".join(sys.argv[1:])}
constexpr int F16_RING = {D};
constexpr int F16_SHAPE = {SHAPE};    // MFMA shape of the core: 32 = 32x32x16, 16 = 16x16x32 (accumulator layout, weight stream and fragment addressing differ)
constexpr int F16_VGPR_CAP = {CAP};   // hipcc's share (amdgpu_num_vgpr); the core owns v[{CAP}:255]
constexpr int F16_X = {CAP + X_OFF}, F16_NET = {CAP + NET_OFF};   // first register of the two accumulator grids
"""]
    out.append(sync_struct())
    out.append(prologue(D))
    out.append(cxx(B_("layer_x_full", X_OFF, NK, NK, D)))
    out.append(cxx(B_("layer_net_full", NET_OFF, NK, NK, D)))
    if SHAPE == 16:   # net = W0 relu(x) + b0 whose only reader is fc_1's operand: relu + split in the block's last k-step (convert_tail)
        out.append(cxx(B_("layer_net_full_cv", NET_OFF, NK, NK, D, convert_tail=True)))
        out.append(cxx(B_("layer_net_full_b", NET_OFF, NK, NK, D, bias_init=True)))     # net := b0 + W0 relu(x), the bias as the first C operand
    # lin_in: 64 inputs = 4 k-blocks; waves 0-3 write unit-rows 0-3 (k < 32), waves 4-7 unit-rows 4-7 (8 KiB further)
    if D == 2:   # (the ring-4 build is a probe-only variant: tools/chain_probe.hip)
        out.append(cxx(B_("layer_x_in", X_OFF, NKI, NKI, D, region1_off=8192)))
        if SHAPE == 16:   # x := W_in * in (no bias init by the caller: lin_in's bias travels in the lin_z[0] map)
            out.append(cxx(B_("layer_x_in0", X_OFF, NKI, NKI, D, region1_off=8192, zero_init=True)))
    out.append(gather_cxx(D))
    if ns:
        out.append("}  // namespace " + ns + "\n")
    sys.stdout.write("\n".join(out))


if __name__ == "__main__":
    main()
